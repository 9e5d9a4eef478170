/*
 * drop_in.c -- a C caller of libfluid_amd.so shaped like the reference's own main()
 * (project/sequential/FluidSequential.c:273-334): six (N+2)^2 float arrays owned by the caller,
 * Z time steps, the mean time per step printed at the end.  The only change against the reference's
 * loop body is the one INTEGRATION.md section 1 describes: the zeroing of the *_prev arrays and the
 * two calls vel_step(...) / dens_step(...) at FluidSequential.c:298-306 become
 *     z == 0:  step_src(N, DT, DIFF, VIS, 40, u, v, dens, u_prev, v_prev, dens_prev);
 *     z  > 0:  step(N, DT, DIFF, VIS, u, v, dens);
 * Plain C99 against include/fluid_amd.h; no HIP, no C++, no Python in this file.
 *
 *   gcc -std=c99 -pedantic -Wall -Iinclude examples/drop_in.c -o drop_in \
 *       -Lfluidsimulationcuda_amd -lfluid_amd -Wl,-rpath,$PWD/fluidsimulationcuda_amd -Wl,-rpath-link,/opt/rocm/lib
 *   ./drop_in N Z [in_prefix out_prefix]
 *
 * With in_prefix the three source fields are read from <in_prefix>_{u,v,dens}_prev.f32 (raw little-endian
 * float32, (N+2)^2 values, row-major) -- that is how tests/test_gpu_dropin.py feeds it the reference's own
 * initial state and checks the result against the reference's golden snapshot; without it the sources are
 * drawn from rand() by the recipe of the reference's initializeParameters (:244-271).  With out_prefix the
 * final u, v, dens are written the same way.
 */
#define _POSIX_C_SOURCE 199309L      /* clock_gettime under -std=c99 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "fluid_amd.h"

#define DT 0.016f
#define VIS 0.0025f
#define DIFF 0.1f

static double seconds(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + (double)ts.tv_nsec * 1.e-9;
}

static int read_field(const char *prefix, const char *name, float *x, size_t count)
{
    char path[1024];
    FILE *f;
    size_t got;
    snprintf(path, sizeof path, "%s_%s.f32", prefix, name);
    f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    got = fread(x, sizeof(float), count, f);
    fclose(f);
    if (got != count) { fprintf(stderr, "%s: short read\n", path); return 1; }
    return 0;
}

static int write_field(const char *prefix, const char *name, const float *x, size_t count)
{
    char path[1024];
    FILE *f;
    size_t put;
    snprintf(path, sizeof path, "%s_%s.f32", prefix, name);
    f = fopen(path, "wb");
    if (!f) { perror(path); return 1; }
    put = fwrite(x, sizeof(float), count, f);
    fclose(f);
    return put != count;
}

/* density source in the centred square of half-width (N+2)/8, velocity sources everywhere */
static void synthetic_sources(int N, float *dens_prev, float *u_prev, float *v_prev)
{
    const int w = N + 2, c = w / 2, r = w / 8;
    int i, j;
    for (i = 0; i < w; i++)
        for (j = 0; j < w; j++)
            dens_prev[j + i * w] = (i >= c - r && i < c + r && j >= c - r && j < c + r) ? (float)(rand() % 100) / 1000.0f : 0.0f;
    for (i = 0; i < w * w; i++) {
        u_prev[i] = (float)(rand() % 100) / 100.0f;
        v_prev[i] = (float)(rand() % 100) / 100.0f;
    }
}

int main(int argc, char **argv)
{
    int N, Z, z, rc = 0;
    size_t cells;
    float *u, *v, *dens, *u_prev, *v_prev, *dens_prev;
    double total = 0.0;
    const char *in_prefix = argc > 3 ? argv[3] : NULL, *out_prefix = argc > 4 ? argv[4] : NULL;

    if (argc < 3) {
        fprintf(stderr, "usage: %s N Z [in_prefix out_prefix]\n", argv[0]);
        return 2;
    }
    N = atoi(argv[1]);
    Z = atoi(argv[2]);
    cells = (size_t)(N + 2) * (size_t)(N + 2);
    u = calloc(cells, sizeof(float));
    v = calloc(cells, sizeof(float));
    dens = calloc(cells, sizeof(float));
    u_prev = calloc(cells, sizeof(float));
    v_prev = calloc(cells, sizeof(float));
    dens_prev = calloc(cells, sizeof(float));
    if (!u || !v || !dens || !u_prev || !v_prev || !dens_prev) { fprintf(stderr, "out of memory\n"); return 1; }
    if (in_prefix) {
        if (read_field(in_prefix, "u_prev", u_prev, cells) || read_field(in_prefix, "v_prev", v_prev, cells) ||
            read_field(in_prefix, "dens_prev", dens_prev, cells))
            return 1;
    } else {
        synthetic_sources(N, dens_prev, u_prev, v_prev);
    }

    for (z = 0; z < Z && rc == FLUID_OK; z++) {
        const double t0 = seconds();
        if (z == 0)
            rc = step_src(N, DT, DIFF, VIS, 40, u, v, dens, u_prev, v_prev, dens_prev);
        else
            rc = step(N, DT, DIFF, VIS, u, v, dens);
        total += seconds() - t0;
    }
    if (rc != FLUID_OK) {
        fprintf(stderr, "libfluid_amd error %d: %s\n", rc, fluid_last_error());
        return 1;
    }
    printf("Tot: %f (mean seconds per step over %d steps, host arrays in and out every step)\n", total / (Z > 0 ? Z : 1), Z);

    if (out_prefix && (write_field(out_prefix, "u", u, cells) || write_field(out_prefix, "v", v, cells) ||
                       write_field(out_prefix, "dens", dens, cells)))
        return 1;
    fluid_release_cached();
    free(u); free(v); free(dens); free(u_prev); free(v_prev); free(dens_prev);
    return 0;
}
