/* A C program on the other side of the boundary: include/ocd.h and include/ocd_cma.h consumed as a reference
 * maintainer's native code would consume them -- no Python, no torch.  TEST INFRASTRUCTURE (tests/test_abi_c_client.py
 * compiles and runs it).
 *
 *   abi_client <libocd_hip.so> <libocd_cma.so> [gpu|desc]
 *
 * Without "gpu": loads both libraries, resolves every entry point it needs by name, validates a scenario descriptor on
 * the host and checks that a compute call WITHOUT a device fails loudly (OCD_ERR_NO_DEVICE, message set) -- there is
 * no CPU fallback behind the ABI.  With "gpu": runs the finite_horizon H = 5 scenario's episodes for two candidates
 * x three inits through ocd_rollout_episodes with plain hipMalloc'ed buffers (HIP runtime resolved by dlsym as well)
 * and one native CMA-ES generation through ocd_cma_run with the launch / wait passed as function pointers; prints the
 * returns and costs (the Python test compares them with the oracle). */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ocd.h"
#include "../../include/ocd_cma.h"

#define SYM(lib, type, name) type name = (type)dlsym(lib, #name); if (!name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }

typedef int32_t (*fn_abi_version)(void);
typedef int32_t (*fn_device_count)(void);
typedef const char *(*fn_last_error)(void);
typedef int32_t (*fn_scenario_create)(const ocd_scenario_desc *, ocd_scenario **);
typedef void (*fn_scenario_destroy)(ocd_scenario *);
typedef int32_t (*fn_rollout)(const ocd_scenario *, const float *, const float *, int64_t, int64_t, int64_t, int64_t, float *, float *, float *, void *);
typedef int32_t (*fn_rollout_indexed)(const ocd_scenario *, const float *, int64_t, const float *, int64_t, const int32_t *, int64_t,
                                      float *, float *, float *, void *);
typedef int32_t (*fn_sync)(void *);
typedef int32_t (*fn_index_error)(const ocd_scenario *, int64_t *);
typedef int32_t (*fn_cma_create)(int32_t, const double *, double, int32_t, uint32_t, ocd_cma **);
typedef void (*fn_cma_destroy)(ocd_cma *);
typedef int32_t (*fn_cma_run)(ocd_cma *, const ocd_cma_run_args *, int64_t *, int32_t *, int32_t *);
typedef int32_t (*fn_normalise)(const double *, int64_t, int64_t, int32_t, float *);
typedef int (*fn_hipMalloc)(void **, size_t);
typedef int (*fn_hipHostMalloc)(void **, size_t, unsigned);
typedef int (*fn_hipMemcpy)(void *, const void *, size_t, int);
typedef int (*fn_hipFree)(void *);

/* finite_horizon_env(horizon=5) as the reference hard-codes it (mpc_ord.py:162-207; world.py:143-152; car.py:33;
 * merging.py:23,72-81; naive_planner.py:20) */
static void finite_horizon_h5(ocd_scenario_desc *d)
{
    memset(d, 0, sizeof *d);
    d->abi_version = OCD_ABI_VERSION;
    d->reward_kind = OCD_REWARD_LANE_FEATURES;
    d->n_cars = 2; d->n_lanes = 3; d->horizon = 5; d->n_iter = 100; d->extra_inits = 0; d->check_plans = 0;
    d->episode_len = 15; d->n_samples = 1; d->teleport_step = 0; d->teleport_period = 0;
    for (int s = 0; s < OCD_MAX_SAMPLES; ++s) d->teleport_car[s] = -1;
    d->teleport_state[0] = 10.0f;
    d->dt = 0.1f; d->dt_sq = (float)(0.1 * 0.1); d->learning_rate = 0.1f; d->ego_friction = 0.2f; d->target_speed = 1.0f;
    d->lane_center[0] = (float)(0.0 + -1.0 * 0.1 * 1); d->lane_center[1] = 0.0f; d->lane_center[2] = (float)(0.0 + -1.0 * 0.1 * -1);
    d->lane_origin_y = -5.0f; d->lane_normal_y = 0.0f;                       /* StraightLane((0, -5), (0, 10), 0.1): world.py:150 */
    d->fence_lo = (float)(0.05 * 3 - 0.05); d->fence_width = 0.05f; d->fence_shape = (float)(5.0 / 0.05);
    d->bump_half_x = 0.08f; d->bump_half_y = 0.15f;
    d->other_init[0][0] = 0.0f; d->other_init[0][1] = -0.6f; d->other_init[0][2] = 0.5f; d->other_init[0][3] = (float)(M_PI / 2);
    d->other_friction[0] = 0.0f;
    /* MPC_ORD.designer_weights: the car's float32 weights normalised once more (mpc_ord.py:24) */
    const double raw[7] = {-5, 0, 0, 0, -6, -50, -50};
    double n2 = 0; for (int i = 0; i < 7; ++i) n2 += raw[i] * raw[i];
    float w32[7]; for (int i = 0; i < 7; ++i) w32[i] = (float)(raw[i] / sqrt(n2));
    float m2 = 0; for (int i = 0; i < 7; ++i) m2 += w32[i] * w32[i];
    for (int i = 0; i < 7; ++i) d->designer_weights[i] = w32[i] / sqrtf(m2);
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: abi_client libocd_hip.so libocd_cma.so [gpu]\n"); return 2; }
    const int gpu = argc > 3 && !strcmp(argv[3], "gpu");
    void *hip = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!hip) { fprintf(stderr, "dlopen %s: %s\n", argv[1], dlerror()); return 2; }
    void *cma = dlopen(argv[2], RTLD_NOW);
    if (!cma) { fprintf(stderr, "dlopen %s: %s\n", argv[2], dlerror()); return 2; }
    SYM(hip, fn_abi_version, ocd_abi_version)
    SYM(hip, fn_device_count, ocd_device_count)
    SYM(hip, fn_last_error, ocd_last_error)
    SYM(hip, fn_scenario_create, ocd_scenario_create)
    SYM(hip, fn_scenario_destroy, ocd_scenario_destroy)
    SYM(hip, fn_rollout, ocd_rollout_episodes)
    SYM(hip, fn_rollout_indexed, ocd_rollout_indexed)
    SYM(hip, fn_index_error, ocd_scenario_index_error)
    SYM(hip, fn_sync, ocd_stream_synchronize)
    SYM(cma, fn_cma_create, ocd_cma_create)
    SYM(cma, fn_cma_destroy, ocd_cma_destroy)
    SYM(cma, fn_cma_run, ocd_cma_run)
    SYM(cma, fn_normalise, ocd_normalise_weights)
    if (ocd_abi_version() != OCD_ABI_VERSION) { fprintf(stderr, "ABI version %d != header %d\n", ocd_abi_version(), OCD_ABI_VERSION); return 1; }
    ocd_scenario_desc d;
    finite_horizon_h5(&d);
    ocd_scenario *scn = NULL;
    if (ocd_scenario_create(&d, &scn) != OCD_OK || !scn) { fprintf(stderr, "scenario_create: %s\n", ocd_last_error()); return 1; }
    ocd_scenario_desc bad = d; bad.horizon = OCD_MAX_HORIZON + 1;
    ocd_scenario *none = NULL;
    if (ocd_scenario_create(&bad, &none) != OCD_ERR_INVALID_ARG) { fprintf(stderr, "an invalid descriptor was accepted\n"); return 1; }
    const int32_t n_dev = ocd_device_count(); /* < 0 (OCD_ERR_NO_DEVICE) when the runtime finds no GPU */
    printf("abi %d devices %d\n", ocd_abi_version(), n_dev);
    const float inits[3][4] = {{0.01f, -0.9f, 0.8f, (float)(M_PI / 2)}, {-0.03f, -0.88f, 0.82f, (float)(M_PI / 2)}, {0.05f, -0.92f, 0.78f, (float)(M_PI / 2)}};
    const double cand[2][7] = {{-5, 0, 0, 0, -6, -50, -50}, {-0.21963165, -0.01184596, 0.34379187, -0.04687411, -0.06364365, -0.54138792, -0.7308079}};
    float w32[2][7];
    if (ocd_normalise_weights(&cand[0][0], 2, 7, 1, &w32[0][0]) != 0) return 1;
    if (argc > 3 && !strcmp(argv[3], "desc")) { /* the descriptor as bytes: the test compares it with the Python host mirror's */
        const unsigned char *b = (const unsigned char *)&d;
        printf("desc ");
        for (size_t i = 0; i < sizeof d; ++i) printf("%02x", b[i]);
        printf("\n");
    }
    if (!gpu) {
        /* without a GPU a compute call must say so, not fall back to anything; WITH one this mode launches nothing (the
         * buffers below are host memory: a kernel reading them would fault) */
        if (n_dev < 1) {
            float ret[6];
            const int32_t st = ocd_rollout_episodes(scn, &inits[0][0], &w32[0][0], 2, 3, 0, 6, ret, NULL, NULL, NULL);
            if (st != OCD_ERR_NO_DEVICE || !strstr(ocd_last_error(), "no CPU fallback")) {
                fprintf(stderr, "expected OCD_ERR_NO_DEVICE, got %d (%s)\n", st, ocd_last_error());
                return 1;
            }
            printf("no-device status %d: %s\n", st, ocd_last_error());
        }
        ocd_scenario_destroy(scn);
        return 0;
    }
    void *rt = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);
    if (!rt) rt = dlopen("libamdhip64.so.7", RTLD_NOW);
    if (!rt) { fprintf(stderr, "HIP runtime: %s\n", dlerror()); return 2; }
    SYM(rt, fn_hipMalloc, hipMalloc)
    SYM(rt, fn_hipHostMalloc, hipHostMalloc)
    SYM(rt, fn_hipMemcpy, hipMemcpy)
    SYM(rt, fn_hipFree, hipFree)
    float *init_dev, *w_dev, *ret_dev, ret[6];
    if (hipMalloc((void **)&init_dev, sizeof inits) || hipMalloc((void **)&w_dev, sizeof w32) || hipMalloc((void **)&ret_dev, sizeof ret)) return 1;
    hipMemcpy(init_dev, inits, sizeof inits, 1);
    hipMemcpy(w_dev, w32, sizeof w32, 1);
    if (ocd_rollout_episodes(scn, init_dev, w_dev, 2, 3, 0, 6, ret_dev, NULL, NULL, NULL) != OCD_OK) { fprintf(stderr, "rollout: %s\n", ocd_last_error()); return 1; }
    if (ocd_stream_synchronize(NULL) != OCD_OK) return 1;
    hipMemcpy(ret, ret_dev, sizeof ret, 2);
    printf("returns");
    for (int i = 0; i < 6; ++i) printf(" %.9g", ret[i]);
    printf("\n");
    /* the same six episodes as rows of an index, in reverse order (ocd_rollout_indexed: independent populations in one
     * launch, the reference's Pool over init groups, run_mpc_ord.py:83-90): bit for bit the flat call's returns */
    {
        int32_t idx[6][3], *idx_pin;
        float back[6];
        for (int e = 0; e < 6; ++e) { idx[e][0] = (5 - e) / 3; idx[e][1] = (5 - e) % 3; idx[e][2] = 5 - e; }
        if (hipHostMalloc((void **)&idx_pin, sizeof idx, 0)) return 1;
        memcpy(idx_pin, idx, sizeof idx);
        if (ocd_rollout_indexed(scn, init_dev, 3, w_dev, 2, idx_pin, 6, ret_dev, NULL, NULL, NULL) != OCD_OK) { fprintf(stderr, "indexed rollout: %s\n", ocd_last_error()); return 1; }
        if (ocd_stream_synchronize(NULL) != OCD_OK) return 1;
        hipMemcpy(back, ret_dev, sizeof back, 2);
        for (int e = 0; e < 6; ++e)
            if (memcmp(&back[e], &ret[5 - e], sizeof(float))) { fprintf(stderr, "indexed episode %d differs from the flat call\n", e); return 1; }
        printf("indexed rollout equals the flat call on 6 episodes\n");
        /* an index row that names no candidate is an ERROR, not a clamp (ABI 3).  Pinned host memory: refused before
         * anything is launched, the row named in the message */
        idx_pin[4 * 3 + 0] = 2;                                    /* row 4: candidate 2 of 2 */
        int32_t st = ocd_rollout_indexed(scn, init_dev, 3, w_dev, 2, idx_pin, 6, ret_dev, NULL, NULL, NULL);
        if (st != OCD_ERR_INVALID_ARG || !strstr(ocd_last_error(), "row 4")) { fprintf(stderr, "bad pinned index: status %d (%s)\n", st, ocd_last_error()); return 1; }
        /* device memory: the kernel finds it -- that episode's return is NaN, the others are untouched, and the handle
         * reports the row after the wait */
        int32_t *idx_dev;
        int64_t bad_row = -2;
        if (hipMalloc((void **)&idx_dev, sizeof idx)) return 1;
        idx[4][1] = -1;                                            /* row 4: init row -1 */
        hipMemcpy(idx_dev, idx, sizeof idx, 1);
        if (ocd_scenario_index_error(scn, &bad_row) != OCD_OK || bad_row != -1) { fprintf(stderr, "index error before any device index\n"); return 1; }
        if (ocd_rollout_indexed(scn, init_dev, 3, w_dev, 2, idx_dev, 6, ret_dev, NULL, NULL, NULL) != OCD_OK) { fprintf(stderr, "device index: %s\n", ocd_last_error()); return 1; }
        if (ocd_stream_synchronize(NULL) != OCD_OK) return 1;
        hipMemcpy(back, ret_dev, sizeof back, 2);
        for (int e = 0; e < 6; ++e)
            if (e == 4 ? back[e] == back[e] : memcmp(&back[e], &ret[5 - e], sizeof(float))) { fprintf(stderr, "device index, episode %d: %g\n", e, back[e]); return 1; }
        st = ocd_scenario_index_error(scn, &bad_row);
        if (st != OCD_ERR_INVALID_ARG || bad_row != 4) { fprintf(stderr, "index error: status %d row %lld\n", st, (long long)bad_row); return 1; }
        if (ocd_scenario_index_error(scn, &bad_row) != OCD_OK) { fprintf(stderr, "the index error was not cleared by reporting it\n"); return 1; }
        /* unreported, the next indexed launch on the handle refuses */
        if (ocd_rollout_indexed(scn, init_dev, 3, w_dev, 2, idx_dev, 6, ret_dev, NULL, NULL, NULL) != OCD_OK) return 1;
        if (ocd_stream_synchronize(NULL) != OCD_OK) return 1;
        idx[4][1] = 1;
        hipMemcpy(idx_dev, idx, sizeof idx, 1);
        st = ocd_rollout_indexed(scn, init_dev, 3, w_dev, 2, idx_dev, 6, ret_dev, NULL, NULL, NULL);
        if (st != OCD_ERR_INVALID_ARG || !strstr(ocd_last_error(), "row 4")) { fprintf(stderr, "sticky index error: status %d (%s)\n", st, ocd_last_error()); return 1; }
        if (ocd_rollout_indexed(scn, init_dev, 3, w_dev, 2, idx_dev, 6, ret_dev, NULL, NULL, NULL) != OCD_OK) return 1;
        if (ocd_stream_synchronize(NULL) != OCD_OK || ocd_scenario_index_error(scn, NULL) != OCD_OK) return 1;
        hipFree(idx_dev);
        printf("out-of-range index rows are errors: pinned index refused, device index NaN + reported\n");
    }
    /* one native CMA-ES generation: population 4, the launch and the wait as function pointers, pinned buffers */
    ocd_cma *es = NULL;
    const double x0[7] = {-0.0624, 0, 0, 0, -0.0749, -0.6244, -0.6244};
    if (ocd_cma_create(7, x0, 0.05, 4, 3u, &es) != 0) return 1;
    float *w_pin, *ret_pin;
    if (hipHostMalloc((void **)&w_pin, 4 * 7 * sizeof(float), 0) || hipHostMalloc((void **)&ret_pin, 4 * 3 * sizeof(float), 0)) return 1;
    double X[4 * 7], cost[4], hist_w[4 * 7], hist_c[4], secs[8];
    int32_t nonf[1], flags[OCD_CMA_N_STOP], pending = 0;
    int64_t done = 0;
    ocd_cma_run_args a;
    memset(&a, 0, sizeof a);
    a.scn = scn; a.init_dev = init_dev; a.N = 3; a.S = 1; a.w_pinned = w_pin; a.ret_pinned = ret_pin; a.stream = NULL;
    a.rollout = (ocd_cma_rollout_fn)ocd_rollout_episodes; a.sync = ocd_stream_synchronize;
    a.normalise_variant = 1; a.max_generations = 1;
    const double opts[OCD_CMA_N_STOP] = {1, INFINITY, 1e-11, 1e-12, 1e-11, 1e3, 1e14, 1e20, 1e9, 1, 0, 0};
    memcpy(a.stop_opts, opts, sizeof opts);
    a.X = X; a.cost = cost; a.hist_w = hist_w; a.hist_cost = hist_c; a.seconds = secs; a.nonfinite = nonf;
    if (ocd_cma_run(es, &a, &done, flags, &pending) != 0) { fprintf(stderr, "ocd_cma_run failed: %s\n", ocd_last_error()); return 1; }
    printf("generation done %lld pending %d maxiter %d costs", (long long)done, pending, flags[0]);
    for (int i = 0; i < 4; ++i) printf(" %.17g", cost[i]);
    printf("\nweights");
    for (int i = 0; i < 4 * 7; ++i) printf(" %.9g", w_pin[i]);
    printf("\n");
    ocd_cma_destroy(es);
    hipFree(init_dev); hipFree(w_dev); hipFree(ret_dev);
    ocd_scenario_destroy(scn);
    return 0;
}
