/*
 * fluid_oracle.c -- CPU restatement of the Stable-Fluids time step.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this.  The shipped path
 * (fluidsimulationcuda_amd/) never links, imports or calls anything in oracle/.
 *
 * What it restates: the vel_step + dens_step hot path of the reference's
 * project/sequential/FluidSequential.c, with N, dt, diff, visc and the Jacobi
 * iteration count as run-time arguments (the reference bakes them in as
 * macros, FluidSequential.c:6-9, and hard-codes 40 sweeps at :91).
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks every function
 * here bit-for-bit against the reference itself, compiled from where it lies
 * under /root/reference by oracle/build_ref.sh (outputs in oracle/_ref/), and
 * tests/test_oracle_golden.py checks it against the committed vectors in
 * tests/golden/ that were captured from that same build
 * (tests/golden/make_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off (no -march / -mfma: contraction changes
 * the bits, SURVEY.md section 7 "Bit-level parity").
 *
 * Layout: one field = (n+2)*(n+2) floats, row-major, cell (col j, row i) at
 * j + i*(n+2); ghost ring at index 0 and n+1 (FluidSequential.c:95,250-258).
 */
#include <pthread.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define AT(j, i) ((size_t)(j) + (size_t)(i) * w)

/* ---- a2: boundary (FluidSequential.c:62-75) ------------------------------
 * Edges copy (or negate, when b selects that axis) the adjacent interior
 * cell; each corner is the mean of its two neighbouring edge ghosts, the
 * horizontal neighbour first (:71-74). */
void fo_set_bnd(int n, int b, float *x)
{
    const size_t w = (size_t)n + 2;
    for (int k = 1; k <= n; ++k) {
        x[AT(0, k)]     = (b == 1) ? -x[AT(1, k)] : x[AT(1, k)];
        x[AT(n + 1, k)] = (b == 1) ? -x[AT(n, k)] : x[AT(n, k)];
        x[AT(k, 0)]     = (b == 2) ? -x[AT(k, 1)] : x[AT(k, 1)];
        x[AT(k, n + 1)] = (b == 2) ? -x[AT(k, n)] : x[AT(k, n)];
    }
    x[AT(0, 0)]         = 0.5f * (x[AT(1, 0)]     + x[AT(0, 1)]);
    x[AT(0, n + 1)]     = 0.5f * (x[AT(1, n + 1)] + x[AT(0, n)]);
    x[AT(n + 1, 0)]     = 0.5f * (x[AT(n, 0)]     + x[AT(n + 1, 1)]);
    x[AT(n + 1, n + 1)] = 0.5f * (x[AT(n, n + 1)] + x[AT(n + 1, n)]);
}

/* ---- a3: sources (FluidSequential.c:78-82): every cell, ghosts included. */
void fo_add_source(int n, float dt, float *x, const float *s)
{
    const size_t cells = ((size_t)n + 2) * ((size_t)n + 2);
    for (size_t c = 0; c < cells; ++c) {
        const float inc = dt * s[c];
        x[c] = x[c] + inc;
    }
}

/* ---- a4: one Jacobi sweep + its boundary (FluidSequential.c:92-101) ------
 * out = (x0 + alpha*(((L + R) + U) + D)) / beta, true division (:95-96). */
void fo_jacobi_sweep(int n, int b, const float *x, const float *x0, float *out,
                     float alpha, float beta)
{
    const size_t w = (size_t)n + 2;
    for (int i = 1; i <= n; ++i) {
        const float *up = x + (size_t)(i - 1) * w;
        const float *me = x + (size_t)i * w;
        const float *dn = x + (size_t)(i + 1) * w;
        const float *rhs = x0 + (size_t)i * w;
        float *o = out + (size_t)i * w;
        for (int j = 1; j <= n; ++j) {
            float nb = me[j - 1] + me[j + 1];
            nb = nb + up[j];
            nb = nb + dn[j];
            const float num = rhs[j] + alpha * nb;
            o[j] = num / beta;
        }
    }
    fo_set_bnd(n, b, out);
}

/* Rows [row_lo,row_hi) of one sweep, no boundary: lets bench.py's all-cores
 * baseline split a sweep over host threads (set_bnd follows once all bands are
 * done).  Same arithmetic as fo_jacobi_sweep. */
void fo_jacobi_rows(int n, const float *x, const float *x0, float *out,
                    float alpha, float beta, int row_lo, int row_hi)
{
    const size_t w = (size_t)n + 2;
    for (int i = row_lo; i < row_hi; ++i) {
        const float *up = x + (size_t)(i - 1) * w;
        const float *me = x + (size_t)i * w;
        const float *dn = x + (size_t)(i + 1) * w;
        const float *rhs = x0 + (size_t)i * w;
        float *o = out + (size_t)i * w;
        for (int j = 1; j <= n; ++j) {
            float nb = me[j - 1] + me[j + 1];
            nb = nb + up[j];
            nb = nb + dn[j];
            const float num = rhs[j] + alpha * nb;
            o[j] = num / beta;
        }
    }
}

/* Whole solve (FluidSequential.c:85-104).  The initial guess is whatever x
 * holds on entry, ghosts included.  The reference ping-pongs pointers and is
 * only safe for an even count (:100,103); here an odd count is handled by a
 * final copy so x always receives the result.  Returns 0, or -1 on OOM. */
int fo_diffuse(int n, int b, float *x, const float *x0, float alpha, float beta,
               int iters)
{
    const size_t cells = ((size_t)n + 2) * ((size_t)n + 2);
    float *scratch = (float *)malloc(cells * sizeof(float));
    if (!scratch) return -1;
    float *cur = x, *nxt = scratch;
    for (int k = 0; k < iters; ++k) {
        fo_jacobi_sweep(n, b, cur, x0, nxt, alpha, beta);
        float *t = cur; cur = nxt; nxt = t;
    }
    if (cur != x) memcpy(x, cur, cells * sizeof(float));
    free(scratch);
    return 0;
}

/* The same solve with every sweep split into row bands over `threads` POSIX threads (bench.py's
 * all-cores CPU baseline; the reference itself is single-threaded).  A sweep reads only the previous
 * sweep's field, so bands are independent; a barrier separates the sweeps, thread 0 applies the
 * boundary between two barriers.  Bit-identical to fo_diffuse.  Returns 0, -1 on OOM / thread failure. */
struct fo_mt_job {
    int n, b, iters, threads, id;
    float *x, *scratch;
    const float *x0;
    float alpha, beta;
    pthread_barrier_t *bar;
};

static void *fo_mt_worker(void *arg)
{
    const struct fo_mt_job *j = (const struct fo_mt_job *)arg;
    const int lo = 1 + (int)(((long long)j->n * j->id) / j->threads);
    const int hi = 1 + (int)(((long long)j->n * (j->id + 1)) / j->threads);
    float *cur = j->x, *nxt = j->scratch;
    for (int k = 0; k < j->iters; ++k) {
        fo_jacobi_rows(j->n, cur, j->x0, nxt, j->alpha, j->beta, lo, hi);
        pthread_barrier_wait(j->bar);
        if (j->id == 0) fo_set_bnd(j->n, j->b, nxt);
        pthread_barrier_wait(j->bar);
        float *t = cur; cur = nxt; nxt = t;
    }
    return NULL;
}

int fo_diffuse_mt(int n, int b, float *x, const float *x0, float alpha, float beta,
                  int iters, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    if (threads > 256) threads = 256;
    const size_t cells = ((size_t)n + 2) * ((size_t)n + 2);
    float *scratch = (float *)malloc(cells * sizeof(float));
    if (!scratch) return -1;
    pthread_barrier_t bar;
    if (pthread_barrier_init(&bar, NULL, (unsigned)threads) != 0) { free(scratch); return -1; }
    struct fo_mt_job jobs[256];
    pthread_t tid[256];
    int started = 0, rc = 0;
    for (int t = 0; t < threads; ++t) {
        jobs[t] = (struct fo_mt_job){n, b, iters, threads, t, x, scratch, x0, alpha, beta, &bar};
        if (t > 0 && pthread_create(&tid[t], NULL, fo_mt_worker, &jobs[t]) != 0) break;
        ++started;
    }
    if (started != threads) {
        /* cannot run short-handed (the barrier counts `threads`): the started workers are parked at the
         * first barrier and nothing has been written yet beyond scratch; fail the process loudly */
        abort();
    }
    fo_mt_worker(&jobs[0]);
    for (int t = 1; t < threads; ++t) pthread_join(tid[t], NULL);
    pthread_barrier_destroy(&bar);
    if (iters & 1) memcpy(x, scratch, cells * sizeof(float));
    free(scratch);
    return rc;
}

/* ---- a5: semi-Lagrangian advection (FluidSequential.c:107-141) ---------- */
void fo_advect(int n, int b, float dt, float *d, const float *d0,
               const float *u, const float *v)
{
    const size_t w = (size_t)n + 2;
    const float dt0 = dt * (float)n;                 /* :111 */
    const float lo = 0.5f;
    const float hi = (float)n + 0.5f;                /* exact for n < 2^23 */
    for (int i = 1; i <= n; ++i) {
        for (int j = 1; j <= n; ++j) {
            float px = (float)j - dt0 * u[AT(j, i)]; /* :114 */
            float py = (float)i - dt0 * v[AT(j, i)]; /* :115 */
            if (px < lo) px = lo;
            if (px > hi) px = hi;
            const int j0 = (int)px, j1 = j0 + 1;
            if (py < lo) py = lo;
            if (py > hi) py = hi;
            const int i0 = (int)py, i1 = i0 + 1;
            const float s1 = px - (float)j0, s0 = 1.0f - s1;
            const float t1 = py - (float)i0, t0 = 1.0f - t1;
            const float a = t0 * d0[AT(j0, i0)] + t1 * d0[AT(j0, i1)];
            const float c = t0 * d0[AT(j1, i0)] + t1 * d0[AT(j1, i1)];
            d[AT(j, i)] = s0 * a + s1 * c;           /* :136-137 */
        }
    }
    fo_set_bnd(n, b, d);
}

/* ---- a6: divergence + pressure clear (FluidSequential.c:143-158) -------- */
void fo_divergence(int n, const float *u, const float *v, float *p, float *div)
{
    const size_t w = (size_t)n + 2;
    const float h = 1.0f / (float)n;
    const float scale = -0.5f * h;                   /* (-0.5f*h) first, :151 */
    for (int i = 1; i <= n; ++i) {
        for (int j = 1; j <= n; ++j) {
            float g = u[AT(j + 1, i)] - u[AT(j - 1, i)];
            g = g + v[AT(j, i + 1)];
            g = g - v[AT(j, i - 1)];
            div[AT(j, i)] = scale * g;
            p[AT(j, i)] = 0.0f;
        }
    }
    fo_set_bnd(n, 0, div);
    fo_set_bnd(n, 0, p);
}

/* ---- a7: pressure-gradient subtraction (FluidSequential.c:161-173) ------ */
void fo_subtract_gradient(int n, float *u, float *v, const float *p)
{
    const size_t w = (size_t)n + 2;
    const float h = 1.0f / (float)n;
    for (int i = 1; i <= n; ++i) {
        for (int j = 1; j <= n; ++j) {
            const float gx = 0.5f * (p[AT(j + 1, i)] - p[AT(j - 1, i)]);
            const float gy = 0.5f * (p[AT(j, i + 1)] - p[AT(j, i - 1)]);
            u[AT(j, i)] = u[AT(j, i)] - gx / h;      /* true /h, :167-168 */
            v[AT(j, i)] = v[AT(j, i)] - gy / h;
        }
    }
    fo_set_bnd(n, 1, u);
    fo_set_bnd(n, 2, v);
}

/* alpha = ((dt*coef)*n)*n in float, beta = 1 + 4*alpha
 * (FluidSequential.c:179-180,199-200). */
void fo_coefficients(int n, float dt, float coef, float *alpha, float *beta)
{
    float a = dt * coef;
    a = a * (float)n;
    a = a * (float)n;
    *alpha = a;
    *beta = 1.0f + 4.0f * a;
}

static int project(int n, int iters, float *u, float *v, float *p, float *div)
{
    fo_divergence(n, u, v, p, div);
    if (fo_diffuse(n, 0, p, div, 1.0f, 4.0f, iters)) return -1;
    fo_subtract_gradient(n, u, v, p);
    return 0;
}

/* ---- a9: velocity step (FluidSequential.c:189-241) -----------------------
 * On return u,v hold the new velocity, u0 the last pressure, v0 the last
 * divergence (SURVEY.md 3.1 "where results land"). */
int fo_vel_step(int n, float dt, float visc, int iters,
                float *u, float *v, float *u0, float *v0)
{
    float alpha, beta;
    fo_add_source(n, dt, u, u0);
    fo_add_source(n, dt, v, v0);
    fo_coefficients(n, dt, visc, &alpha, &beta);
    /* after the reference's SWAPs the diffused field lives in the *_prev
     * buffer and the source-added field is the right-hand side */
    if (fo_diffuse(n, 1, u0, u, alpha, beta, iters)) return -1;
    if (fo_diffuse(n, 2, v0, v, alpha, beta, iters)) return -1;
    if (project(n, iters, u0, v0, /*p=*/u, /*div=*/v)) return -1;
    fo_advect(n, 1, dt, u, u0, u0, v0);
    fo_advect(n, 2, dt, v, v0, u0, v0);
    return project(n, iters, u, v, /*p=*/u0, /*div=*/v0);
}

/* ---- a8: density step (FluidSequential.c:176-186) ------------------------
 * On return x holds the new density, x0 the diffused-not-advected one. */
int fo_dens_step(int n, float dt, float diff, int iters,
                 float *x, float *x0, const float *u, const float *v)
{
    float alpha, beta;
    fo_add_source(n, dt, x, x0);
    fo_coefficients(n, dt, diff, &alpha, &beta);
    if (fo_diffuse(n, 0, x0, x, alpha, beta, iters)) return -1;
    fo_advect(n, 0, dt, x, x0, u, v);
    return 0;
}

/* One loop body of the reference's main (FluidSequential.c:305-306). */
int fo_step_src(int n, float dt, float diff, float visc, int iters,
                float *u, float *v, float *dens,
                float *u_prev, float *v_prev, float *dens_prev)
{
    if (fo_vel_step(n, dt, visc, iters, u, v, u_prev, v_prev)) return -1;
    return fo_dens_step(n, dt, diff, iters, dens, dens_prev, u, v);
}

/* Loop body for z > 0: sources zeroed first (FluidSequential.c:298-302). */
int fo_step(int n, float dt, float diff, float visc, int iters,
            float *u, float *v, float *dens,
            float *u_prev, float *v_prev, float *dens_prev)
{
    const size_t cells = ((size_t)n + 2) * ((size_t)n + 2);
    memset(u_prev, 0, cells * sizeof(float));
    memset(v_prev, 0, cells * sizeof(float));
    memset(dens_prev, 0, cells * sizeof(float));
    return fo_step_src(n, dt, diff, visc, iters, u, v, dens,
                       u_prev, v_prev, dens_prev);
}

/* ---- a10: the reference's synthetic input (FluidSequential.c:244-271) ----
 * Draws come from the caller's generator so the same routine serves glibc
 * rand() (to mirror the reference here) and the build's own portable PRNG
 * (bench.py on the GPU box).  draw() must return a value in [0, 100). */
void fo_initialize(int n, int (*draw)(void *), void *state,
                   float *dens, float *dens_prev, float *u, float *u_prev,
                   float *v, float *v_prev)
{
    const int w = n + 2;
    const int c = w / 2, r = w / 8;
    for (int i = 0; i < w; ++i)
        for (int j = 0; j < w; ++j) {
            const int in = (j < c + r) && (j >= c - r) && (i < c + r) && (i >= c - r);
            dens_prev[(size_t)j + (size_t)i * w] = in ? (float)draw(state) / 1000.0f : 0.0f;
            dens[(size_t)j + (size_t)i * w] = 0.0f;
        }
    for (int i = 0; i < w; ++i)
        for (int j = 0; j < w; ++j) {
            const size_t k = (size_t)j + (size_t)i * w;
            u_prev[k] = (float)draw(state) / 100.0f;
            v_prev[k] = (float)draw(state) / 100.0f;
            u[k] = 0.0f;
            v[k] = 0.0f;
        }
}

static int draw_glibc(void *unused) { (void)unused; return rand() % 100; }

/* glibc rand() with an explicit seed (the reference never seeds => seed 1). */
void fo_initialize_glibc(int n, unsigned seed, float *dens, float *dens_prev,
                         float *u, float *u_prev, float *v, float *v_prev)
{
    srand(seed);
    fo_initialize(n, draw_glibc, NULL, dens, dens_prev, u, u_prev, v, v_prev);
}

/* Portable generator used for bench/test inputs: SplitMix64 -> [0,100). */
static int draw_splitmix(void *state)
{
    unsigned long long *s = (unsigned long long *)state;
    unsigned long long z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (int)((z >> 33) % 100u);
}

void fo_initialize_portable(int n, unsigned long long seed, float *dens,
                            float *dens_prev, float *u, float *u_prev,
                            float *v, float *v_prev)
{
    unsigned long long s = seed;
    fo_initialize(n, draw_splitmix, &s, dens, dens_prev, u, u_prev, v, v_prev);
}
