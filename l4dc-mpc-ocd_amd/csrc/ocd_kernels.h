// ocd_kernels.h -- host/device shared declarations of the planner kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocd.h"

namespace ocd {

enum { OCD_MODE_ROLLOUT = 0, OCD_MODE_PLAN = 1 };

// How the lanes of a wavefront exchange the terms of the horizon recurrences
enum {
    V_LDS = 0,   // K wavefronts per workgroup (one per control initialisation), segments of H lanes,
                 // zero-padded LDS windows: any H, the throughput variant
    V_ROW = 1,   // K wavefronts per workgroup, one trajectory per 16-lane DPP row (H <= 16): row_shr/row_shl
    V_SEG = 2,   // ONE wavefront per workgroup holding all K initialisations of its trajectories in
                 // segments of H lanes (K*H <= 64): wave_shr/wave_shl moves, argmin inside the wavefront
    V_CHUNK = 3  // ONE wavefront per workgroup, a lane owns a chunk of S consecutive horizon steps, H/S lanes
                 // per (trajectory, initialisation): long horizons at throughput (ocd_chunk_kernel.hip)
};

// Terminal-value table of the planner (leaf_evaluation); device pointers owned by the scenario handle
struct LeafParams {
    const float *grid;        // [n0 + n1 + n2]
    const float *values;      // [n0, n1, n2]
    int32_t n[3];
    int32_t proj_kind;
    int32_t grid_in_lds;      // 1: the launch reserved n0 + n1 + n2 floats of LDS behind the variant's own: the kernel stages
                              //    the cell boundaries there (the corner search reads them 4-8 times per pass, dependently)
};

// Kernel argument block (passed by value: lives in the kernarg segment, read
// through scalar loads, so scenario constants cost no vector registers).
struct KernelParams {
    ocd_scenario_desc d;
    const float *ego_states;   // ROLLOUT: init_states [N,4];  PLAN / ROLLOUT-from-state: world_state [B,C,4]
    const float *weights;      // ROLLOUT: [P,D];  PLAN: [B,D] or [D]
    const float *other_plans;  // [C-1,H,2] or nullptr (constant-velocity model)
    const float *init_speed;   // PLAN: [B] the car's own speed for the extra_inits (naive_planner.py:114) or nullptr
    float *returns_out;        // ROLLOUT [n]
    float *traj_out;           // ROLLOUT [n,T+1,C,4] or nullptr
    float *ctrl_out;           // ROLLOUT [n,T,2] or nullptr
    float *plans_out;          // PLAN [B,H,2]
    float *best_loss_out;      // PLAN [B] or nullptr
    int32_t *best_init_out;    // PLAN [B] or nullptr
    float *all_plans_out;      // PLAN [B,K,H,2] or nullptr
    float *all_losses_out;     // PLAN [B,K] or nullptr
    LeafParams leaf;           // leaf.values == nullptr: no terminal value
    long long n_problems;      // trajectories handled by this launch
    long long ep_begin;        // ROLLOUT: flat index of the first episode
    long long N;               // ROLLOUT: number of init states
    int32_t S;                 // ROLLOUT: samples per (candidate, init)
    int32_t mode;
    int32_t weights_per_problem;
    int32_t K;                 // control initialisations
    int32_t T;                 // control steps to run (ROLLOUT); 1 in PLAN mode
    int32_t t0;                // world step index of the first control step (scripted plans, teleport)
    int32_t from_state;        // ROLLOUT: 1 = every problem starts from its own world_state [B,C,4]
    int32_t sample_fixed;      // from_state: which world.reset() outcome (teleported car) applies
    int32_t reset_phase;       // ROLLOUT: world.reset() calls before episode 0 of this batch (teleport cycle)
    int32_t segs_used;         // trajectories per wavefront; 0 = let the launcher choose
    int32_t no_skips;          // diagnostics: 1 = always evaluate collision and fence features
    int32_t scan_mode;         // 0 = automatic, 1 = V_LDS, 2 = V_ROW, 3 = V_SEG, 4 = V_CHUNK
    int32_t no_unify;          // diagnostics: 1 = never use the one-feature-per-lane evaluation
    // the two knobs as lane masks (all ones / zero), so the kernels test them with plain scalar and/or
    unsigned long long force_full, force_full_any;
    int32_t no_latency_build;  // diagnostics: 1 = never pick the LAT builds of V_ROW / V_SEG
    int32_t chunk_size;        // V_CHUNK: 0 = pick the compiled chunk size by cost, else force this one
    int32_t n_cus;             // compute units of the device (launch shape heuristics)
    unsigned long long *debug; // diagnostic builds only (OCD_STAMPS): per-wavefront cycle totals, else nullptr
    int32_t *launch_info;      // HOST pointer or nullptr: the launcher records its choice here (ocd_scenario_last_launch)
    int32_t dry_run;           // 1 = choose and record, launch nothing (ocd_scenario_plan_launch: no device needed)
    // ROLLOUT, indexed (ocd_rollout_indexed): episode `prob` = (weight row, init row, reset number) of ep_index[prob],
    // instead of the flat (p, n, s) decomposition of ep_begin + prob.  A row outside [0, P_rows) / [0, N) / reset < 0 is a
    // caller bug: an index in HOST memory is refused before the launch (ocd_api.hip); one in device memory is caught here --
    // the episode reads row 0 (memory-safe), starts from a NaN ego (its return is NaN, never a plausible number) and
    // index_error[0] gets 1 + the row's position (ocd_scenario_index_error reports it)
    const int32_t *ep_index;   // [n_problems, 3] or nullptr
    long long P_rows;          // rows of `weights` (indexed rollouts only)
    int32_t *index_error;      // device-visible pinned word of the handle, or nullptr
    // fence_shape * fence_width < 1/80: smooth_threshold is 0/0 on the road in the reference itself; such a handle runs the
    // generic kernels only, every feature of every lane, BOTH sides of the fence as merging.py:80-81 writes them
    int32_t two_sided;
};

// Which (candidate row, init row, entry of the teleport cycle) episode `prob` of a rollout launch runs (both planner
// kernels): the flat index e = ep_begin + prob = (p * N + n) * S + s of ocd_rollout_episodes, or row `prob` of the
// caller's episode index (ocd_rollout_indexed: independent populations evaluated by one launch).
// Returns false for an index row out of range (the caller poisons the episode: NaN ego state).
__device__ __forceinline__ bool episode_rows(const KernelParams &p, long long prob, long long &p_, long long &n_, int &tp_idx)
{
    const int period = p.d.teleport_period;
    bool ok = true;
    if (p.ep_index) {
        const int32_t *ix = p.ep_index + 3 * prob;
        long long pr = ix[0], nr = ix[1], reset = ix[2];
        ok = pr >= 0 && pr < p.P_rows && nr >= 0 && nr < p.N && reset >= 0;
        if (!ok) {
            pr = 0; nr = 0; reset = 0;
            if (p.index_error) p.index_error[0] = (int32_t)(prob < 0x7ffffffe ? prob + 1 : 0x7fffffff);
        }
        p_ = pr; n_ = nr;
        tp_idx = (int)(reset % (period > 0 ? period : p.S));
    } else {
        const long long e_glob = p.ep_begin + prob;           // flat (p, n, s) index
        const long long s_ = e_glob % p.S;
        n_ = (e_glob / p.S) % p.N;
        p_ = e_glob / ((long long)p.S * p.N);
        // ReplanningCarWorld.reset() toggles the removed car on EVERY reset (replanning_world.py:24-27):
        // episode e of a sequential evaluation is reset number reset_phase + e
        tp_idx = period > 0 ? (int)((p.reset_phase + e_glob) % period) : (int)s_;
    }
    return ok;
}

// the launchers' only way to start a planner kernel: nothing is launched in a dry run
#define OCD_LAUNCH(KERNEL, GRID, BLOCK, LDS, STREAM, P) \
    do { if (!(P).dry_run) hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, P); } while (0)
static inline hipError_t launch_status(const KernelParams &p) { return p.dry_run ? hipSuccess : hipGetLastError(); }

// what a launch chose: {scan mode 1..4, chunk size S (V_CHUNK) else 0, trajectories per wavefront, workgroups,
// wavefronts per SIMD the build is compiled for (1 = the latency build LAT, 0 = unconstrained), H specialised (0 =
// run-time H), terminal value 0/1, wavefronts per workgroup}
static inline void note_launch(const KernelParams &p, int mode, int chunk, int segs, unsigned blocks, int build_waves,
                               int ht, int leaf, int waves_per_group)
{
    if (!p.launch_info) return;
    int32_t *o = p.launch_info;
    o[0] = mode; o[1] = chunk; o[2] = segs; o[3] = (int32_t)blocks; o[4] = build_waves; o[5] = ht; o[6] = leaf;
    o[7] = waves_per_group;
}

// (horizon H, scripted cars NO, lanes L) triples with a planner kernel specialised on H (loops fully
// unrolled, all three exchange variants).  Horizons 5/6: the reference's own settings; 10/15/25:
// BASELINE.json configs 2-5; 3: the reference's planner known-answer test.  Every other horizon in
// [1, OCD_MAX_HORIZON] runs the generic kernel of its (NO, L) pair (run-time H, V_LDS).
#define OCD_KERNEL_TABLE(X)                                                                   \
    X(3, 0, 0) X(5, 0, 0)                                                                     \
    X(5, 1, 3) X(6, 1, 3) X(10, 1, 3) X(15, 1, 3) X(25, 1, 3)                                 \
    X(5, 2, 2) X(10, 2, 2) X(15, 2, 2)                                                        \
    X(5, 2, 3) X(10, 2, 3) X(25, 2, 3)

// (scripted cars NO, lanes L) pairs: generic planner kernel, reward kernel, objective kernel.
// L = 0 is the target-speed test reward, L = -1 the linear target-speed reward (any number of cars: the others
// do not enter either).
#define OCD_PAIR_TABLE(X)                                                                     \
    X(0, 0) X(1, 0) X(2, 0) X(3, 0)                                                           \
    X(0, -1) X(1, -1)                                                                         \
    X(1, 1) X(1, 2) X(1, 3) X(1, 4)                                                           \
    X(2, 1) X(2, 2) X(2, 3) X(2, 4)                                                           \
    X(3, 1) X(3, 2) X(3, 3) X(3, 4)

// (horizon H, scripted cars NO, lanes L, chunk S) with a chunked kernel (V_CHUNK); several S per shape; S need not
// divide H (the last lane of a segment then owns fewer steps).  Small S: more wavefronts of fewer instructions each --
// the best size is the smallest one whose wavefronts still fit one per SIMD (launch_chunk_dispatch picks by cost).
#define OCD_CHUNK_TABLE(X)                                                                    \
    X(10, 1, 3, 2) X(10, 1, 3, 5) X(15, 1, 3, 2) X(15, 1, 3, 3) X(15, 1, 3, 5) X(25, 1, 3, 2) X(25, 1, 3, 3) X(25, 1, 3, 5) \
    X(10, 2, 2, 2) X(10, 2, 2, 5) X(15, 2, 2, 2) X(15, 2, 2, 3) X(15, 2, 2, 5)                \
    X(10, 2, 3, 2) X(10, 2, 3, 5) X(25, 2, 3, 2) X(25, 2, 3, 3) X(25, 2, 3, 5)

// (H, NO, L) with V_ROW / V_SEG LAT builds that carry the terminal value: the reference's own horizons on the shape
// its value grids are made for (coarse_value_iteration.py: the three-lane finite-horizon world)
template <int HT, int NO, int L> struct leaf_specialised { static constexpr bool value = (HT == 5 || HT == 6) && NO == 1 && L == 3; };

hipError_t launch_mpc_dispatch(int H, int NO, int L, const KernelParams &p, hipStream_t st, bool *supported);
// V_CHUNK: *chunk = the chunk size it would use / used for this shape (0 = none): the compiled size that
// packs the batch into the fewest full rounds of wavefronts, or `want` (> 0) if that size is compiled;
// launches when `launch`
hipError_t launch_chunk_dispatch(int H, int NO, int L, const KernelParams &p, hipStream_t st, bool launch, int want,
                                 int *chunk);
hipError_t launch_reward(int NO, int L, const KernelParams &p, float *feats, float *rew, hipStream_t st, bool *supported);
// R(u) and dR/du for caller-supplied controls [B,H,2] (naive_planner.py:33-77)
hipError_t launch_objective(int NO, int L, const KernelParams &p, const float *controls, float *reward_out,
                            float *grad_out, float *traj_out, hipStream_t st, bool *supported);
// the reward evaluations of ocd_device.h side by side (ocd_debug_kernels.hip): out [B, 8, 5], valid [B, 8]
hipError_t launch_feature_variants(int NO, int L, const KernelParams &p, float *out, int32_t *valid, hipStream_t st, bool *supported);
hipError_t launch_math(const float *in, float *e, float *s, float *c, long long n, hipStream_t st);
hipError_t launch_packed_math(const float *num, const float *den, const float *x, float *div_scalar, float *div_packed,
                              float *exp_scalar, float *exp_packed, long long n_pairs, hipStream_t st);
hipError_t launch_guarded_division(const float *u, const float *n, const float *w, float *m_out, float *k_out, float *q_out,
                                   long long n_pairs, hipStream_t st);
hipError_t launch_dynamics(const float *states, const float *controls, float dt, float dt_sq, float friction,
                           float *out, long long n, hipStream_t st);

} // namespace ocd
