/*
 * fluid_amd.h -- C ABI of the MI355X Stable-Fluids step (libfluid_amd.so).
 *
 * Drop-in boundary for the vel_step + dens_step hot path of the reference's
 * project/sequential/FluidSequential.c.  The reference has no library or FFI
 * interface of its own: its boundary is the two calls at FluidSequential.c:305-306
 * with N, DT, VIS, DIFF as compile-time macros (:6-9).  Each entry point below
 * cites the reference lines whose behaviour it reproduces.  Plain C types only.
 *
 * Host arrays ("fields") are what the reference's main() allocates (:277-282):
 * (N+2)*(N+2) floats, row-major, cell (column j, row i) at j + i*(N+2), ghost
 * ring at index 0 and N+1.  The caller owns them; the library owns all device
 * memory and scratch and never allocates per step.
 *
 * Every function returns FLUID_OK (0) or a FLUID_E_* code and never exits or
 * aborts (the reference's CUDA variants exit(EXIT_FAILURE) in their CHECK
 * macro, naivePar/FluidParallelBlockPerElement-Naive.cu:26-35).
 * fluid_last_error() returns a description of the calling thread's last failure.
 *
 * Threading: one context is used by one host thread at a time; different
 * contexts are independent.
 */
#ifndef FLUID_AMD_H
#define FLUID_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLUID_OK        0
#define FLUID_E_INVALID 1 /* bad N, odd or negative sweep count, null pointer, bad field id */
#define FLUID_E_NOMEM   2 /* device or host allocation failed */
#define FLUID_E_HIP     3 /* a HIP runtime call failed */
#define FLUID_E_COMM    4 /* the multi-GPU exchange callback failed or is missing */

/* Field ids of a context = the six arrays of the reference's main()
 * (FluidSequential.c:277-282) plus six library-owned scratch fields: TMP0-2 are the other half of a solve's
 * ping-pong (the reference malloc()s one per diffuse call, :88), TMP3-5 receive `x + dt*s` when add_source runs
 * inside the first launch of the solve that consumes it (FLUID_PARAM_FUSE_ADD_SOURCE). */
enum {
    FLUID_U = 0, FLUID_V = 1, FLUID_DENS = 2,
    FLUID_U_PREV = 3, FLUID_V_PREV = 4, FLUID_DENS_PREV = 5,
    FLUID_TMP0 = 6, FLUID_TMP1 = 7, FLUID_TMP2 = 8,
    FLUID_TMP3 = 9, FLUID_TMP4 = 10, FLUID_TMP5 = 11,
    FLUID_NFIELDS = 12
};

/* Jacobi kernels (identical results, different data paths).  TB = temporally
 * blocked: up to 16 sweeps per launch, same bits as that many single-sweep launches. */
enum { FLUID_JACOBI_STREAM = 0, FLUID_JACOBI_LDS = 1, FLUID_JACOBI_NAIVE = 2, FLUID_JACOBI_TB = 3 };

/* Tuning knobs for fluid_set_param(); none of them changes results with FLUID_STORAGE_F32.  (With FLUID_STORAGE_F16
 * a fused launch rounds once when it stores, so the launch schedule -- TB_MAX_SWEEPS, TB_MIN_CELLS -- is part of the
 * result there; all other knobs are speed only in both storage types.) */
enum {
    FLUID_PARAM_TB_MAX_SWEEPS = 0, /* most sweeps fused per launch by FLUID_JACOBI_TB: 16 (default), 8, 4 or 2 */
    FLUID_PARAM_TB_ROWS = 1,       /* output rows per wave strip of FLUID_JACOBI_TB; 0 = auto          */
    FLUID_PARAM_HALO = 2,          /* multi-GPU ghost-zone depth (clamped to slab height - 1)          */
    FLUID_PARAM_TB_FAST_DIVISION = 3 /* how FLUID_JACOBI_TB may replace x/beta by an exactly equivalent reciprocal form,
                                      each after proving the equivalence for that beta on all 2^32 float inputs on
                                      the device.  2 (default): one float multiply when beta is a power of two (and
                                      alpha 1), else Markstein's residual correction with the residual scaled by 2^24
                                      (two float multiplies and two fused multiply-adds, exact for every |x| < 2^104; a
                                      wave that stores inf or NaN -- the only thing larger dividends can turn into --
                                      repeats its strip with the double-precision form).  3: the double-precision
                                      multiply for every such beta (the round-2 default).  1: the two-term float
                                      reciprocal fma(x, hi, x*lo) in waves whose right-hand side is nowhere smaller
                                      than beta * 2^-72 (which bounds every dividend away from the range where that
                                      form is one ulp off), the double-precision form elsewhere.  0: always divide */
    ,FLUID_PARAM_TB_EDGE_ROWS_PCT = 5 /* strip height of the two windows that carry the ghost columns, in % of
                                      the interior windows' (default 40; 0 = same): load balance only  */
    ,FLUID_PARAM_TB_LANE_COLUMNS = 6 /* columns per lane of FLUID_JACOBI_TB: 2 (default; thin waves, 4 per SIMD)
                                      or 4 (2 per SIMD): speed only                                      */
    ,FLUID_PARAM_TB_T16_MIN_CELLS = 7 /* 16-sweep launches (fp32 storage, 2-column lanes) on slabs of at least this many
                                      cells, for either form of the solve; -1 (default): the measured rule -- from
                                      8 M cells, the general form only once a field outgrows 96 MiB     */
    ,FLUID_PARAM_TB_AUTOTUNE = 8   /* 1 (default): with TB_ROWS = 0 the strip height of each launch shape is measured
                                      at run time -- the first ~20 launches of a shape try a handful of heights, the
                                      fastest is kept for the process; 0: the closed-form choice.  Speed only.  */
    ,FLUID_PARAM_SLAB_OVERLAP = 10 /* 1 (default): on row slabs fluid_step runs the density diffusion on a second stream
                                      beside the velocity path (a slab's launches leave most of the chip idle), when the
                                      ghost zones cover a whole solve; 0: one stream.  Speed only.                */
    ,FLUID_PARAM_EARLY_ADVECT = 11 /* 1 (default): on row slabs fluid_step starts each advection on the PREVIOUS step's velocity
                                      bound (+25 %) while the new bound is still on its way to the host, and repeats it in
                                      the rare case the bound grew past that (its inputs are still intact then), so the
                                      GPU does not idle while the host reads the bound; 0: wait first.  Speed only. */
    ,FLUID_PARAM_FUSE_DIVERGENCE = 9 /* 1 (default): inside fluid_step / fluid_vel_step on one GPU the divergence of a
                                      projection is computed by the first launch of the pressure solve that consumes it
                                      (no separate pass over u, v); 0: its own kernel first.  Speed only.        */
    ,FLUID_PARAM_FUSE_ADD_SOURCE = 12 /* 1 (default): inside fluid_step / fluid_vel_step / fluid_dens_step a non-zero source is
                                      added by the first launch of the diffusion that consumes the sum (which reads both
                                      operands anyway: the source is its first guess) and stored out of place, instead of
                                      by a pass of its own over the field; 0: add_source as its own kernel.  Speed only. */
    ,FLUID_PARAM_XCHG_OVERLAP = 13 /* 1 (default): on row slabs every exchange is enqueued on a stream of its own, ordered against
                                      the compute stream by events, and the halo exchange that feeds a Jacobi solve is not
                                      waited for at once: the solve's first launch runs the strips that need this slab's own
                                      rows only while the rows travel, and the strips next to the slab's edges behind the
                                      exchange's event.  0: exchanges in line on the context's stream.  Speed only.   */
    ,FLUID_PARAM_F16_PRESSURE_SCALE = 14 /* fp16 storage only.  1 (default): inside a step the divergence and the pressure of a
                                      projection are stored multiplied by 2^(floor(log2 N) - 2) -- plain, they are of the order
                                      h * |velocity| and fall into fp16's subnormal range from a few thousand cells per side
                                      on -- and divided back exactly (in the gradient subtraction; on the host when u_prev /
                                      v_prev are downloaded).  0: plain values.  Changes fp16 results (not fp32 ones).      */
    ,FLUID_PARAM_TB_MIN_CELLS = 4  /* FLUID_JACOBI_TB fuses sweeps only on slabs of at least this many cells
                                      (default 0: always); smaller ones run one-thread-per-cell sweeps   */
};

typedef struct fluid_ctx fluid_ctx;

const char *fluid_last_error(void);

/* ---- the reference's loop body on host arrays ----------------------------
 * step():     FluidSequential.c:298-306 for z > 0 -- the three *_prev source
 *             arrays are zero, then vel_step(u,v,u_prev,v_prev,visc) and
 *             dens_step(dens,dens_prev,u,v,diff), 40 Jacobi sweeps per solve
 *             (:91).  u, v, dens are updated in place.
 * step_src(): FluidSequential.c:305-306 verbatim, sources supplied by the
 *             caller (the z == 0 step).  On return u_prev holds the last
 *             pressure, v_prev the last divergence and dens_prev the diffused
 *             density, exactly where the reference leaves them (SWAPs at
 *             :201,209,228-229,181,184).  `iters` must be even and >= 0: with an
 *             odd count the reference free()s the caller's array (:100,103).
 * Both keep a per-thread cached context for the last N, upload, run one step
 * on the current HIP device, and download. */
int step(int N, float dt, float diff, float visc, float *u, float *v, float *dens);
int step_src(int N, float dt, float diff, float visc, int iters,
             float *u, float *v, float *dens,
             float *u_prev, float *v_prev, float *dens_prev);
/* Drops the cached context of step()/step_src() (frees its device memory). */
int fluid_release_cached(void);

/* alpha = ((dt*coef)*N)*N, beta = 1 + 4*alpha in float (FluidSequential.c:179-180,199-200). */
int fluid_coefficients(int N, float dt, float coef, float *alpha, float *beta);

/* ---- device-resident context --------------------------------------------- */
typedef struct fluid_config {
    int n;              /* interior size N (grid is (N+2)^2), N >= 1                    */
    int rank, nranks;   /* row-slab decomposition: this context owns slab `rank` of `nranks` */
    int halo;           /* Jacobi ghost-zone depth between exchanges (0 = default)      */
    int jacobi_variant; /* FLUID_JACOBI_*                                               */
    void *stream;       /* hipStream_t to run on, or NULL for a library-owned stream    */
    void *arena;        /* device memory of fluid_arena_bytes_ex(n, storage) bytes, or NULL to hipMalloc */
    size_t arena_bytes;
    int storage;        /* FLUID_STORAGE_F32 (default; bit parity with the reference) or FLUID_STORAGE_F16 */
} fluid_config;

/* Field storage on the device.  F16 (BASELINE config "fp16 fields with fp32 Jacobi accumulate"):
 * fields are IEEE half, every operator widens its inputs to float, computes exactly as the fp32
 * path does, and rounds to nearest once when it stores; the fused Jacobi kernel keeps the
 * intermediate sweeps of a launch in fp32 registers.  Host arrays at this ABI stay float. */
enum { FLUID_STORAGE_F32 = 0, FLUID_STORAGE_F16 = 1 };

size_t fluid_arena_bytes(int N);                       /* fp32 storage */
size_t fluid_arena_bytes_ex(int N, int storage);
/* Device layout of one field: W = N+2 rows of `pitch` elements, column c at
 * element index c + xoff; field f starts f*field_floats elements into the arena. */
int fluid_layout(int N, int *pitch, int *xoff, size_t *field_floats);

int fluid_create(int N, fluid_ctx **out);                      /* 1 GPU, defaults */
int fluid_create_ex(const fluid_config *cfg, fluid_ctx **out);
int fluid_destroy(fluid_ctx *ctx);
int fluid_synchronize(fluid_ctx *ctx);

/* Interior rows [*row_lo, *row_hi) owned by this context's slab (1..N+1 for one GPU). */
int fluid_owned_rows(fluid_ctx *ctx, int *row_lo, int *row_hi);
/* Device address of row 0 of a field, valid until the next solver call: a
 * field keeps its id but may trade buffers with TMP0 inside a solve, so the
 * exchange callback must ask every time. */
int fluid_field_ptr(fluid_ctx *ctx, int field, void **dev_ptr);
/* Device address of the 4-byte reduction scalar (a non-negative float) that
 * FLUID_XCHG_MAX_BEGIN reduces in place; it lives in the last 256 bytes of the arena. */
int fluid_scalar_ptr(fluid_ctx *ctx, void **dev_ptr);

/* Host <-> device copies of a whole field, or of rows [row_lo,row_hi) of it
 * (host pointer is always to the full (N+2)^2 array). Synchronous. */
int fluid_upload(fluid_ctx *ctx, int field, const float *host);
int fluid_download(fluid_ctx *ctx, int field, float *host);
int fluid_upload_rows(fluid_ctx *ctx, int field, const float *host, int row_lo, int row_hi);
int fluid_download_rows(fluid_ctx *ctx, int field, float *host, int row_lo, int row_hi);
int fluid_fill(fluid_ctx *ctx, int field, float value);

/* nsteps loop bodies of FluidSequential.c:289-312 on the resident fields.  With
 * use_sources != 0 the first step consumes the resident *_prev fields as
 * sources (z == 0); every other step zeroes them first (:298-302). */
int fluid_step(fluid_ctx *ctx, float dt, float diff, float visc, int iters,
               int nsteps, int use_sources);
int fluid_vel_step(fluid_ctx *ctx, float dt, float visc, int iters);   /* FluidSequential.c:189-241 */
int fluid_dens_step(fluid_ctx *ctx, float dt, float diff, int iters);  /* FluidSequential.c:176-186 */

/* ---- single operators on resident fields (field ids; all in place) -------- */
int fluid_op_set_bnd(fluid_ctx *ctx, int b, int x);                               /* :62-75   */
int fluid_op_add_source(fluid_ctx *ctx, int x, int s, float dt);                  /* :78-82   */
int fluid_op_jacobi_sweep(fluid_ctx *ctx, int b, int x, int x0, int out,
                          float alpha, float beta);                               /* :92-101, one k */
int fluid_op_diffuse(fluid_ctx *ctx, int b, int x, int x0, float alpha, float beta,
                     int iters);                                                  /* :85-104; uses TMP0 */
int fluid_op_advect(fluid_ctx *ctx, int b, int d, int d0, int u, int v, float dt); /* :107-141 */
int fluid_op_divergence(fluid_ctx *ctx, int u, int v, int p, int div);            /* :143-158 */
int fluid_op_subtract_gradient(fluid_ctx *ctx, int u, int v, int p);              /* :161-173 */

/* ---- diagnostics (wavefront reductions; never alter the fields) ----------- */
/* max over owned interior cells of |beta*x - alpha*(L+R+U+D) - x0| */
int fluid_residual(fluid_ctx *ctx, int x, int x0, float alpha, float beta, float *out);
/* max over owned interior cells of max(|u|,|v|) */
int fluid_absmax_velocity(fluid_ctx *ctx, int u, int v, float *out);

int fluid_set_jacobi_variant(fluid_ctx *ctx, int variant);
/* How FLUID_JACOBI_TB divides by `beta` in a solve with these coefficients (diagnostic; runs the on-device proof
 * if this beta has not been seen): 0 true division, 2 double-precision reciprocal, 3 two-term float reciprocal
 * where the right-hand side allows it (else as 2), 4 exact float reciprocal (beta a power of two, alpha 1), 5 float
 * reciprocal with scaled residual correction. */
int fluid_division_mode(fluid_ctx *ctx, float alpha, float beta, int *mode);
/* The launch depths FLUID_JACOBI_TB uses for one solve of `iters` sweeps on a grid (or slab) of `rows` x N cells: host
 * logic only (no device needed) -- `pressure_form`: alpha 1 / beta 4; `max_sweeps`, `t16_min_cells` as the parameters of the
 * same names (-1: default rule).  Writes up to `capacity` depths and the number of launches. */
int fluid_plan_sweeps(int N, int rows, int storage, int pressure_form, int iters, int max_sweeps, int t16_min_cells,
                      int *depths, int capacity, int *count);
/* Launch shapes whose strip height is still being measured (FLUID_PARAM_TB_AUTOTUNE), process-wide: a benchmark runs
 * untimed steps until this reaches 0. */
int fluid_autotune_pending(fluid_ctx *ctx, int *shapes_open);
int fluid_set_param(fluid_ctx *ctx, int key, int value);

/* ---- timing: HIP events on the context's stream around every operator -------
 * Categories are the reference's per-kernel timers (timeSource, timeDiffusion,
 * timeDivergence, timeProjection, timeAdvection: FluidSequential.c:16,192-234). */
enum {
    FLUID_TIME_SOURCE = 0, FLUID_TIME_DIFFUSION = 1, FLUID_TIME_DIVERGENCE = 2,
    FLUID_TIME_PROJECTION = 3, FLUID_TIME_ADVECTION = 4, FLUID_TIMING_CATEGORIES = 5
};
typedef struct fluid_timing {
    double jacobi_ms;      /* device time inside Jacobi solves since the last reset (= category DIFFUSION) */
    long long sweeps;      /* Jacobi sweeps executed in those solves                */
    long long solves;
    double category_ms[FLUID_TIMING_CATEGORIES];
    long long category_calls[FLUID_TIMING_CATEGORIES];
    long long jacobi_launches;        /* Jacobi kernel launches in those solves                              */
    long long jacobi_field_launches;  /* the same, counting a launch once per field it sweeps (a batched launch
                                         sweeps up to three): x12 B x cells = the launches' compulsory bytes */
    double pressure_ms;               /* the part of jacobi_ms spent in the pressure solves of fluid_step /
                                         fluid_vel_step (alpha 1, beta 4, b 0: FluidSequential.c:222,240)     */
    long long pressure_sweeps;
} fluid_timing;
int fluid_timing_enable(fluid_ctx *ctx, int on);
int fluid_timing_read(fluid_ctx *ctx, fluid_timing *out, int reset);

/* ---- opt-in extension (changes results: NOT the reference's fixed 40 sweeps) -
 * Jacobi in blocks of `check_every` (even) sweeps until the max-norm residual
 * max|beta*x - alpha*(L+R+U+D) - x0| <= tol, or max_iters.  The report proposes
 * a convergence-aware solve as future work (document/main.tex:356). */
int fluid_op_diffuse_tol(fluid_ctx *ctx, int b, int x, int x0, float alpha, float beta,
                         float tol, int max_iters, int check_every,
                         int *iters_done, float *residual);

/* ---- multi-GPU: row slabs, one context (and one process) per GPU ----------
 * The solver calls back whenever rows must move between slabs; the host layer
 * implements it with RCCL (torch.distributed) on the context's stream.
 *   FLUID_XCHG_HALO  : for each listed field, send `depth` owned rows at each
 *                      slab edge to that neighbour and receive its rows into
 *                      the rows just outside the owned range.
 *   FLUID_XCHG_GATHER: all-gather the owned rows (end slabs: plus the ghost
 *                      row) of each listed field, so every rank holds the full field.
 *   FLUID_XCHG_MAX   : *scalar = max over ranks of *scalar (host value, synchronous).
 *   FLUID_XCHG_MAX_BEGIN / _END: the same reduction split so the solver can keep the GPU busy
 *                      while it is in flight.  BEGIN (scalar == NULL): enqueue, on the context's
 *                      stream, an in-place MAX over ranks of the device scalar (fluid_scalar_ptr);
 *                      the solver then copies it to the host asynchronously.  END: *scalar holds
 *                      that host copy; a transport that cannot reduce on the device leaves BEGIN
 *                      empty and replaces *scalar by the max over ranks here.
 * Return 0 on success. */
enum { FLUID_XCHG_HALO = 0, FLUID_XCHG_GATHER = 1, FLUID_XCHG_MAX = 2, FLUID_XCHG_MAX_BEGIN = 3,
       FLUID_XCHG_MAX_END = 4 };
typedef int (*fluid_exchange_fn)(void *user, int kind, const int *fields, int nfields,
                                 int depth, float *scalar);
int fluid_set_exchange(fluid_ctx *ctx, fluid_exchange_fn fn, void *user);
/* The stream (hipStream_t) an exchange callback should enqueue on, asked from INSIDE the callback: with
 * FLUID_PARAM_XCHG_OVERLAP the library runs its exchanges on a stream of their own (already ordered behind the kernels that
 * produced the rows; the library orders the consumers behind it).  A callback that keeps using the stream given to
 * fluid_create_ex stays correct -- it is merely not overlapped with compute. */
int fluid_exchange_stream(fluid_ctx *ctx, void **stream);
/* Jacobi launches so far that ran as interior strips + edge strips around an exchange in flight (diagnostic). */
int fluid_split_launches(fluid_ctx *ctx, long long *count);
/* Runs the installed exchange now for the listed fields (HALO with `depth` rows, or GATHER): how a caller collects a
 * whole field on every rank, and how the transport can be exercised on its own. */
int fluid_exchange_now(fluid_ctx *ctx, int kind, const int *fields, int nfields, int depth);

/* ---- the library's own exchange: RCCL over xGMI, no host in the path ------------------------------------
 * north_star: "one-row ghost cells exchanged via RCCL Sendrecv over xGMI" (the reference is single-device:
 * naivePar/FluidParallelBlockPerElement-Naive.cu:351-355 only ever selects device 0).  Halo rows travel as grouped
 * ncclSend/ncclRecv between neighbouring slabs, the advect fall-back as grouped ncclBroadcast, the velocity bound
 * as an in-place ncclAllReduce(max) on the device scalar; everything is enqueued on the context's stream.
 * librccl is bound at run time ($FLUID_RCCL_LIB if set; else an RCCL the process already holds; else the
 * system's), so single-GPU users need none.
 *   fluid_rccl_available():       FLUID_OK if librccl could be loaded and every entry point bound in THIS process (no
 *                                 communication): all ranks agree on it (e.g. a MIN all-reduce over the transport that
 *                                 carries the id) BEFORE anyone enters the collective attach -- a rank that cannot load
 *                                 the library would otherwise leave its peers waiting inside ncclCommInitRank
 *   fluid_rccl_unique_id():       one rank calls it and hands the FLUID_RCCL_ID_BYTES bytes to all others (any way)
 *   fluid_exchange_rccl_attach(): every rank, with its context (rank / nranks from fluid_create_ex) and that id, on
 *                                 the thread whose current HIP device is the context's: creates the communicator,
 *                                 runs one all-reduce and one send/receive probe, installs the exchange
 *   ..._attach_comm():            the same on a communicator (ncclComm_t) the caller owns
 *   ..._detach():                 removes it (and destroys a communicator the library created)
 *   ..._calls():                  exchanges issued so far: halo, gather, max                                  */
#define FLUID_RCCL_ID_BYTES 128
int fluid_rccl_available(void);
int fluid_rccl_unique_id(void *id, size_t bytes);
int fluid_exchange_rccl_attach(fluid_ctx *ctx, const void *id, size_t bytes);
int fluid_exchange_rccl_attach_comm(fluid_ctx *ctx, void *nccl_comm);
int fluid_exchange_rccl_detach(fluid_ctx *ctx);
int fluid_exchange_rccl_calls(fluid_ctx *ctx, long long *halo, long long *gather, long long *max);

#ifdef __cplusplus
}
#endif
#endif /* FLUID_AMD_H */
