/* cma_pool_client.c -- ocd_cma_run_many (include/ocd_cma.h, ABI 8) with host threads, from plain C and without a GPU:
 * the episode launch is a function pointer (here a host function of the candidate's weights), exactly as the Python
 * tests of tests/test_cma_lockstep.py do it, but with nothing of the interpreter in the process -- so the whole program,
 * csrc/ocd_cma.c included, can be compiled with -fsanitize=thread (tests/test_cma_pool_tsan.py builds and runs it).
 *
 *   cma_pool_client R GENS THREADS GROUPS [WAIT_US]  ->  prints "checksum <hex>" over every run's costs, means and step sizes
 *   (WAIT_US: the stand-in stream wait sleeps that long -- beyond a few milliseconds the idle workers go to sleep on their
 *    condition variable and are woken for the next tells)
 *
 * The checksum must not depend on THREADS / GROUPS; a data race makes ThreadSanitizer report and exit non-zero. */
#define _POSIX_C_SOURCE 200809L
#include "../../include/ocd_cma.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define D 7
#define MAXR 64

static int64_t g_S = 1;

/* ocd_rollout_indexed's signature: fp32 "returns" from the fp32 weight rows and the init rows the index names */
static int32_t rollout_idx(const void *scn, const float *init, int64_t n_rows, const float *w, int64_t p_rows, const int32_t *idx,
                           int64_t E, float *ret, float *traj, float *ctrl, void *stream)
{
    (void)scn; (void)n_rows; (void)p_rows; (void)traj; (void)ctrl; (void)stream;
    for (int64_t e = 0; e < E; ++e) {
        const float *wr = w + (size_t)idx[3 * e] * D, *in = init + (size_t)idx[3 * e + 1] * 4;
        float acc = 0.0f;
        for (int i = 0; i < D; ++i) { const float dlt = wr[i] - 0.2f * (float)(i + 1) / D; acc = acc + dlt * dlt * (1.0f + (float)i); }
        ret[e] = -(acc * (1.0f + in[0])) + 0.01f * (float)(idx[3 * e + 2] % 2);
    }
    return 0;
}

static long g_wait_us = 0;
static int32_t sync_noop(void *stream)
{
    (void)stream;
    if (g_wait_us > 0) {
        struct timespec t = {g_wait_us / 1000000, (g_wait_us % 1000000) * 1000};
        nanosleep(&t, NULL);
    }
    return 0;
}

static uint64_t mix(uint64_t h, const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv)
{
    const int R = argc > 1 ? atoi(argv[1]) : 28, gens = argc > 2 ? atoi(argv[2]) : 50;
    const int threads = argc > 3 ? atoi(argv[3]) : 4, groups = argc > 4 ? atoi(argv[4]) : 0;
    g_wait_us = argc > 5 ? atol(argv[5]) : 0;
    if (R < 1 || R > MAXR || gens < 1) return 2;
    if (ocd_cma_abi_version() != OCD_CMA_ABI_VERSION) { fprintf(stderr, "abi %d\n", ocd_cma_abi_version()); return 3; }
    ocd_cma *es[MAXR];
    double *X[MAXR], *cost[MAXR];
    int64_t run_n0[MAXR], run_N[MAXR], run_p0[MAXR];
    int64_t P_rows = 0, N_rows = 0, E_max = 0;
    for (int r = 0; r < R; ++r) {
        double x0[D];
        for (int i = 0; i < D; ++i) x0[i] = 0.1 * (double)((r * 7 + i * 3) % 11) - 0.5;
        if (ocd_cma_create(D, x0, 0.1 + 0.05 * (r % 4), 0, (uint32_t)(100 + r), &es[r]) != 0) return 4;
        const int lam = ocd_cma_popsize(es[r]);
        X[r] = (double *)calloc((size_t)lam * D, sizeof(double));
        cost[r] = (double *)calloc((size_t)lam, sizeof(double));
        run_N[r] = 1 + r % 3; run_n0[r] = N_rows; run_p0[r] = P_rows;
        N_rows += run_N[r]; P_rows += lam; E_max += (int64_t)lam * run_N[r] * g_S;
    }
    float *inits = (float *)calloc((size_t)N_rows * 4, sizeof(float));
    for (int64_t i = 0; i < N_rows * 4; ++i) inits[i] = 0.01f * (float)(i % 13) - 0.05f;
    float *w = (float *)calloc((size_t)P_rows * D, sizeof(float)), *ret = (float *)calloc((size_t)E_max, sizeof(float));
    int32_t *index = (int32_t *)calloc((size_t)E_max * 3, sizeof(int32_t));
    double *hist_w = (double *)calloc((size_t)gens * P_rows * D, sizeof(double)), *hist_c = (double *)calloc((size_t)gens * P_rows, sizeof(double));
    uint8_t *evaluated = (uint8_t *)calloc((size_t)gens * R, 1), *active = (uint8_t *)malloc((size_t)R), *pending = (uint8_t *)calloc((size_t)R, 1);
    int32_t *nonf = (int32_t *)calloc((size_t)gens * R, sizeof(int32_t)), *flags = (int32_t *)calloc((size_t)R * OCD_CMA_N_STOP, sizeof(int32_t));
    int64_t *launched = (int64_t *)calloc((size_t)gens, sizeof(int64_t));
    double *seconds = (double *)calloc((size_t)gens * 8, sizeof(double));
    memset(active, 1, (size_t)R);
    double stop_opts[OCD_CMA_N_STOP];
    /* pycma's defaults (reward_design/cmaes.py) in the header's order, with maxiter inside the call (the drop-out path runs) */
    const double defaults[OCD_CMA_N_STOP] = {0.0, INFINITY, 1e-11, 1e-12, 1e-11, 1e3, 1e14, 1e20, 248.0, 1.0, 0.0, 0.0};
    memcpy(stop_opts, defaults, sizeof stop_opts);
    stop_opts[0] = (double)(gens * 3 / 4);
    void *streams[8];
    for (int k = 0; k < 8; ++k) streams[k] = (void *)(uintptr_t)(0x1000 + 16 * k);
    ocd_cma_many_args a;
    memset(&a, 0, sizeof a);
    a.init_dev = inits; a.N_rows = N_rows; a.P_rows = P_rows; a.S = g_S; a.R = R; a.normalise_variant = 0;
    a.run_n0 = run_n0; a.run_N = run_N; a.run_p0 = run_p0;
    a.w_pinned = w; a.index_pinned = index; a.ret_pinned = ret;
    a.rollout = (ocd_cma_rollout_indexed_fn)rollout_idx; a.sync = sync_noop;
    a.max_generations = gens; a.stop_opts = stop_opts; a.active = active; a.X = X; a.cost = cost;
    a.hist_w = hist_w; a.hist_cost = hist_c; a.evaluated = evaluated; a.seconds = seconds; a.nonfinite = nonf;
    a.episodes_launched = launched; a.stop_flags = flags; a.pending_nan = pending;
    a.n_groups = groups; a.streams = groups > 1 ? streams : NULL; a.host_threads = threads;
    int64_t done = 0;
    const int32_t st = ocd_cma_run_many(es, &a, &done);
    if (st != 0) { fprintf(stderr, "ocd_cma_run_many -> %d\n", st); return 5; }
    uint64_t h = 1469598103934665603ull;
    h = mix(h, &done, sizeof done);
    h = mix(h, hist_c, sizeof(double) * (size_t)done * P_rows);
    h = mix(h, hist_w, sizeof(double) * (size_t)done * P_rows * D);
    h = mix(h, evaluated, (size_t)done * R);
    h = mix(h, flags, sizeof(int32_t) * (size_t)R * OCD_CMA_N_STOP);
    for (int r = 0; r < R; ++r) {
        double mean[D], sigma, C[D * D], bx[D], bf, md;
        int64_t gen, ce;
        ocd_cma_state(es[r], mean, &sigma, C, bx, &bf, &gen, &ce, &md);
        h = mix(h, mean, sizeof mean); h = mix(h, &sigma, sizeof sigma); h = mix(h, C, sizeof C); h = mix(h, &gen, sizeof gen);
        ocd_cma_destroy(es[r]);
        free(X[r]); free(cost[r]);
    }
    printf("done %lld\nchecksum %016llx\n", (long long)done, (unsigned long long)h);
    free(inits); free(w); free(ret); free(index); free(hist_w); free(hist_c); free(evaluated); free(active); free(pending);
    free(nonf); free(flags); free(launched); free(seconds);
    return 0;
}
