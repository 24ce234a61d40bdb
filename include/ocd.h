/*
 * include/ocd.h -- C ABI of the MI355X batched MPC planner ("ocd" = optimal
 * control design; the reference is avikj/L4DC-MPC-OCD).
 *
 * The reference has NO native / FFI boundary: it is pure Python on TensorFlow
 * and its seam is Python-level.  Every entry point below therefore names the
 * reference *Python* interface it replaces (file:line relative to the
 * reference tree); the ctypes binding a maintainer would add on the reference
 * side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, plain pointers and sizes, no torch / HIP types in signatures;
 *   - every data pointer of the device entry points is a DEVICE pointer
 *     (e.g. torch-ROCm tensor.data_ptr()); the caller owns every buffer, the
 *     library owns only the opaque scenario handle;
 *   - `hip_stream` is a hipStream_t passed as void* (NULL = default stream);
 *     calls are asynchronous on that stream, the caller synchronises;
 *   - return 0 on success, negative ocd_status on error; the message is
 *     available from ocd_last_error() (thread-local); nothing throws across
 *     the boundary;
 *   - all arithmetic is IEEE binary32 with the operation order documented in
 *     DESIGN.md section 3 ("arithmetic contract").
 */
#ifndef OCD_H
#define OCD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCD_ABI_VERSION 3   /* 3: lane_origin_y / lane_normal_y (the y-term of StraightLane.dist2median in the scored reward);
                              * ocd_rollout_indexed, ocd_debug_feature_variants; out-of-range index rows are an error */

#define OCD_MAX_CARS 4      /* ego + up to 3 scripted cars                    */
#define OCD_MAX_OTHERS 3
#define OCD_MAX_LANES 4
#define OCD_MAX_FEATURES 8  /* D = n_lanes + 4                                */
#define OCD_MAX_PLAN 8      /* scripted controls of a FixedPlanCar            */
#define OCD_MAX_SAMPLES 4   /* world.reset() repetitions per (candidate,init) */
#define OCD_MAX_HORIZON 32  /* planning horizon H handled by the HIP path     */
#define OCD_MAX_CTRL_INITS 6

typedef enum ocd_status {
    OCD_OK = 0,
    OCD_ERR_INVALID_ARG = -1,
    OCD_ERR_UNSUPPORTED = -2,   /* configuration outside the compiled kernels */
    OCD_ERR_HIP = -3,           /* a HIP runtime call failed                  */
    OCD_ERR_NO_DEVICE = -4
} ocd_status;

/* Which car.reward_fn the planning car uses. */
typedef enum ocd_reward_kind {
    /* ThreeLaneTestCar.features + LinearRewardCar.reward_fn
     * (experiments/merging.py:32-83, interact_drive/car/linear_reward_car.py:49-55):
     * D = n_lanes + 4 features, r = sum_d w_d * phi_d. */
    OCD_REWARD_LANE_FEATURES = 0,
    /* TargetSpeedPlannerCar.reward_fn = -(v - target_speed)^2
     * (interact_drive/planner/tests/targetSpeedRewardMaximizerCar.py:49-57);
     * only used to pin the planner against the reference's own known-answer
     * tests (interact_drive/planner/tests/test_naivePlanner.py:21-63). */
    OCD_REWARD_TARGET_SPEED = 1,
    /* LinearTargetSpeedPlannerCar (interact_drive/reward_design/tests/linearTargetSpeedPlannerCar.py:11-44):
     * D = 2 features [v, (v - target_speed)^2], r = w_0 * v + w_1 * (v - target)^2 with the LinearRewardCar
     * weights.  The planning car of the reference's inverse-optimal-control tests
     * (reward_design/tests/test_first_order_ioc.py:29-60: weights (2, -1), target 0, v = 1, friction 0 "leads to
     * zero controls"): the last planner known answer the reference holds.  n_lanes is ignored (0). */
    OCD_REWARD_LINEAR_TARGET_SPEED = 2
} ocd_reward_kind;

/* Number of reward features D (the length of a weight vector) of a descriptor's reward kind. */
#define OCD_N_FEATURES(reward_kind, n_lanes) \
    ((reward_kind) == OCD_REWARD_LANE_FEATURES ? (n_lanes) + 4 : ((reward_kind) == OCD_REWARD_LINEAR_TARGET_SPEED ? 2 : 0))

/*
 * Static description of one driving scenario: everything the reference's
 * scenario factories hard-code (interact_drive/reward_design/mpc_ord.py:162-207,
 * experiments/local_opt_scenario.py:6-55, experiments/replanning_world.py:45-95,
 * experiments/merging.py:86-99) plus the planner arguments
 * (interact_drive/planner/naive_planner.py:19-30).  Car 0 is the planning
 * ("ego") car; cars 1..n_cars-1 are scripted.
 */
typedef struct ocd_scenario_desc {
    int32_t abi_version;        /* OCD_ABI_VERSION */
    int32_t reward_kind;        /* ocd_reward_kind */
    int32_t n_cars;             /* C in [1, OCD_MAX_CARS] */
    int32_t n_lanes;            /* L in [0, OCD_MAX_LANES]; world.lanes (world.py:143-159) */
    int32_t horizon;            /* H, NaivePlanner.horizon */
    int32_t n_iter;             /* I, SGD steps per control initialisation (naive_planner.py:20) */
    int32_t extra_inits;        /* 0: K=3 inits, 1: K=6 (naive_planner.py:107-116) */
    int32_t check_plans;        /* PlannerCar.check_plans (planner_car.py:58-80) */
    int32_t episode_len;        /* T, MPC_ORD.designer_horizon (mpc_ord.py:95) */
    int32_t n_samples;          /* S, MPC_ORD.num_samples (mpc_ord.py:87) */
    int32_t teleport_step;      /* ReplanningCarWorld.critical_t (replanning_world.py:18,32); 0 = never */
    int32_t teleport_car[OCD_MAX_SAMPLES]; /* car index moved away by the i-th reset of the teleport cycle
                                            * (replanning_world.py:24-27,33); see teleport_period */
    float teleport_state[4];    /* (10,0,0,0) (replanning_world.py:34) */

    float dt;                   /* CarWorld.dt (world.py:19) */
    float dt_sq;                /* fp32(dt ** 2): the reference squares the Python float in double
                                 * before the fp32 multiply (simulation_utils.py:14) */
    float learning_rate;        /* SGD learning rate (naive_planner.py:20,28) */
    float ego_friction;         /* Car.friction of the planning car (car.py:33) */
    float target_speed;         /* ThreeLaneTestCar.target_speed (merging.py:29) */
    float lane_center[OCD_MAX_LANES]; /* StraightLane.p[0] per lane (world.py:150-151,157-158) */
    float fence_lo;             /* fp32(0.05*num_lanes - 0.05): threshold - width (math_utils.py:89, merging.py:80) */
    float fence_width;          /* 0.05 */
    float fence_shape;          /* c / width = 100 (math_utils.py:85).  fence_shape * fence_width is the reference's c = 5; below
                                 * 1/80 smooth_threshold is 0/0 = NaN on a band of the road in the reference itself: such a
                                 * descriptor is accepted and answered like the reference (NaN), by the generic kernels with
                                 * both sides of the fence evaluated as merging.py:80-81 writes them (no specialised build) */
    float bump_half_x;          /* 0.08 (merging.py:72) */
    float bump_half_y;          /* 0.15 (merging.py:73) */

    float other_init[OCD_MAX_OTHERS][4];      /* scripted cars' init_state */
    float other_friction[OCD_MAX_OTHERS];     /* 0 FixedVelocityCar (fixed_velocity_car.py:23), 0.2 FixedPlanCar */
    int32_t other_plan_len[OCD_MAX_OTHERS];   /* len(FixedPlanCar.plan); 0 = always other_default */
    float other_plan[OCD_MAX_OTHERS][OCD_MAX_PLAN][2];
    float other_default[OCD_MAX_OTHERS][2];   /* FixedPlanCar.default_control / FixedControlCar.control */

    float designer_weights[OCD_MAX_FEATURES]; /* MPC_ORD.designer_weights, fp32, already normalised (mpc_ord.py:24) */

    /* ---- ABI 2 ---- */
    /* ReplanningCarWorld.reset() toggles the removed car on EVERY reset (replanning_world.py:24-27), so
     * in a sequential evaluation the episode with flat index e is reset number reset_phase + e and
     * loses car teleport_car[(reset_phase + e) % teleport_period] ("reset_phase": ocd_scenario_set_option).
     * 0 = index teleport_car by the sample s instead (identical whenever n_samples == teleport_period). */
    int32_t teleport_period;    /* in [0, OCD_MAX_SAMPLES]; 2 for ReplanningCarWorld */
    /* What PlannerCar._get_next_control assumes for a scripted car beyond its plan when check_plans is
     * set (planner_car.py:66-75): default_control if the car has a plan and a default_control, else
     * (0, 0) -- also for a FixedControlCar, whose REAL control is other_default. */
    float other_assumed_default[OCD_MAX_OTHERS][2];

    /* ---- ABI 3 ---- */
    /* StraightLane.dist2median is r = (x - p[0]) * n[0] + (y - p[1]) * n[1] with the lane normal n = (-1, 0.0)
     * (world.py:184-187,216-217): the second term is +-0 for every finite y and NaN for y = +-inf / NaN.  The SCORED
     * reward -- car.reward_fn(past_state, ...) of mpc_ord.py:99, ocd_reward_batch -- carries it, so an episode whose
     * ego leaves the finite numbers scores NaN exactly as the reference's does; the planner's objective and its
     * gradient (naive_planner.py:33-77) keep r = (x - p[0]) * -1 (identical bits for every finite y).
     * lane_origin_y = StraightLane.p[1] (-5 in both worlds, world.py:150,157); lane_normal_y = n[1] must be 0:
     * lanes run along y, anything else is OCD_ERR_UNSUPPORTED. */
    float lane_origin_y;
    float lane_normal_y;
} ocd_scenario_desc;

typedef struct ocd_scenario ocd_scenario; /* opaque: validated descriptor + device-side constants */

/* Library / device probes. */
int32_t ocd_abi_version(void);
/* Number of visible HIP devices, or a negative ocd_status. */
int32_t ocd_device_count(void);
/* Thread-local description of the last error returned on this thread. */
const char *ocd_last_error(void);

/* Validate a descriptor and build the handle the kernels read their constants
 * from.  Replaces the reference's scenario factories + NaivePlanner.__init__
 * (naive_planner.py:19-30, planner_car.py:49-52).  Any planning horizon in
 * [1, OCD_MAX_HORIZON] runs; the lane-feature reward takes 1..OCD_MAX_OTHERS scripted cars (its collision feature
 * needs one) and 1..OCD_MAX_LANES lanes, the target-speed test reward 0..OCD_MAX_OTHERS scripted cars. */
int32_t ocd_scenario_create(const ocd_scenario_desc *desc, ocd_scenario **out);
void ocd_scenario_destroy(ocd_scenario *scn);

/* Per-handle options (no process-global state; set them before launching on other threads).
 * Tuning knobs -- results never depend on them:
 *   "segs_per_wave": trajectories packed into one wavefront, 0 = automatic;
 *   "scan_mode": how the horizon recurrences exchange terms and where the K control initialisations
 *                live: 0 = automatic, 1 = LDS windows (K wavefronts per workgroup), 2 = DPP row shifts
 *                (H <= 16, K wavefronts per workgroup), 3 = all K initialisations in one wavefront
 *                (K*H <= 64, wavefront shifts, no workgroup barrier), 4 = a lane owns a chunk of consecutive
 *                horizon steps (long horizons at throughput); a mode the scenario cannot use falls back to 1;
 *   "chunk_size": horizon steps per lane of scan_mode 4 (0 = automatic among the compiled sizes);
 *   "concurrent_launches": G launches of this handle are in flight on G streams at a time (the lockstep runs' groups,
 *                ocd_cma.h: ocd_cma_run_many): the launch rules plan each for 1 / G of the device's compute units, so
 *                that launches of at most one wavefront per SIMD sit side by side (0 / 1 = the whole device, default);
 *   "no_unified_features": 1 = never evaluate a lane's single active feature through the shared path;
 *   "no_feature_skips": 1 = evaluate the collision and fence features even where they are provably
 *                       zero (diagnostics; default 0);
 *   "no_latency_build": 1 = never pick the branch-free builds of scan modes 2 / 3 that launches with at
 *                       most one wavefront per SIMD use (diagnostics; default 0).
 * Episode bookkeeping:
 *   "reset_phase": world.reset() calls made before episode 0 of the next ocd_rollout_episodes batch
 *                  (see ocd_scenario_desc.teleport_period); default 0. */
int32_t ocd_scenario_set_option(ocd_scenario *scn, const char *name, int32_t value);

/* Diagnostics: what the most recent planner launch of this handle chose (no counterpart in the reference).
 * info = {scan mode 1..4 (see "scan_mode"), chunk size (mode 4, else 0), trajectories per wavefront, workgroups,
 *         wavefronts per SIMD the build was compiled for (1 = the latency build, 0 = unconstrained),
 *         horizon the kernel is specialised on (0 = run-time horizon), terminal value 0/1, wavefronts per workgroup};
 * all zero before the first launch. */
int32_t ocd_scenario_last_launch(const ocd_scenario *scn, int32_t info[8]);

/* The same record for a launch of n_problems trajectories on a device of n_cus compute units (0 = 256, MI355X) WITHOUT
 * launching anything: pure host logic, needs no device (the launch rules are unit-tested through it). */
int32_t ocd_scenario_plan_launch(const ocd_scenario *scn, int64_t n_problems, int32_t n_cus, int32_t info[8]);

/*
 * Terminal value of the planner: NaivePlanner(leaf_evaluation=ValueFeature(...).interpolate_value(t))
 * (naive_planner.py:20,69-70; reward_design/value_interpolation.py:28-61).  When set, the reward of
 * the LAST horizon step is the trilinear interpolation of `values` at the coarse state
 * proj(world_state) instead of car.reward_fn: proj_kind 0 = (x, y, v) of the planning car,
 * 1 = (x, y, v * sin(heading)) (the coarse state of coarse_value_iteration.py:117-124).  Outside
 * [grid[0], grid[-1]] in any dimension the value is NaN and its gradient ZERO: the reference's traced function
 * returns the constant float('nan') there (value_interpolation.py:59-60), so the plan's objective is NaN while the
 * controls keep the finite gradients of the other horizon steps.
 *   grid0/1/2 [n0]/[n1]/[n2]  ascending cell boundaries (ValueFeature.disc_grid), HOST pointers
 *   values    [n0, n1, n2]    one time slice of v_grids (v_grids[t]), HOST pointer
 * The handle copies the table (to every device it launches on).  values == NULL removes it.
 * Not to be called while launches that use the handle are still in flight (the device copy is freed).
 */
int32_t ocd_scenario_set_leaf_value(ocd_scenario *scn, const float *grid0, int32_t n0,
                                    const float *grid1, int32_t n1, const float *grid2, int32_t n2,
                                    const float *values, int32_t proj_kind);

/*
 * One receding-horizon plan for each of B independent world states.
 * Replaces NaivePlanner.generate_plan(init_state, weights, other_controls)
 * (naive_planner.py:81-164): K control initialisations x n_iter plain-SGD
 * ascent steps on the H x 2 controls, one more evaluation of the objective,
 * first-index argmin of the loss.
 *
 *   world_state  [B, C, 4]   (x, y, v, heading) per car, car 0 = ego
 *   weights      [B, D] when weights_per_problem != 0, else [D] shared;
 *                ignored (may be NULL) for OCD_REWARD_TARGET_SPEED
 *   other_plans  [C-1, H, 2] controls the planner assumes for the scripted
 *                cars (naive_planner.py:53-59), or NULL = constant velocity
 *                (naive_planner.py:60-65)
 *   plans_out    [B, H, 2]   the selected control sequence
 *   best_loss_out[B]         -R of the selected initialisation   (may be NULL)
 *   best_init_out[B]         index k of the selected initialisation (may be NULL)
 *   all_plans_out[B, K, H, 2], all_losses_out [B, K]  per-initialisation
 *                results before the argmin (may be NULL; used by parity tests)
 */
int32_t ocd_plan_batch(const ocd_scenario *scn,
                       const float *world_state,
                       const float *weights, int32_t weights_per_problem,
                       const float *other_plans,
                       float *plans_out, float *best_loss_out, int32_t *best_init_out,
                       float *all_plans_out, float *all_losses_out,
                       int64_t B, void *hip_stream);

/*
 * ocd_plan_batch with the planning car's OWN current speed given apart from the state the plan starts from.
 * The extra_inits control initialisations coast at friction * self.car.state[2] ** 2 (naive_planner.py:112-116):
 * the speed of the car object, not of the `init_state` argument.  The two coincide whenever the planner is
 * called from CarWorld.step (init_state = None), which is what ocd_plan_batch and the rollouts assume;
 * generate_plan(init_state=<another state>) with extra_inits needs this entry point.
 *   init_speed [B]  self.car.state[2] per problem (device pointer), or NULL = the ego speed in world_state
 * Everything else as ocd_plan_batch.
 */
int32_t ocd_plan_batch_from(const ocd_scenario *scn,
                            const float *world_state, const float *init_speed,
                            const float *weights, int32_t weights_per_problem,
                            const float *other_plans,
                            float *plans_out, float *best_loss_out, int32_t *best_init_out,
                            float *all_plans_out, float *all_losses_out,
                            int64_t B, void *hip_stream);

/*
 * Full receding-horizon episodes: for every (candidate p, init n, sample s)
 * with flat index e = (p*N + n)*S + s in [ep_begin, ep_end): reset the world,
 * then T times { optional teleport; score the PRE-step state with the
 * designer weights; plan; apply the first planned control through the real
 * dynamics; step the scripted cars }.  Replaces MPC_ORD.eval_weights_for_init
 * (mpc_ord.py:67-106) driven over CarWorld.step (world.py:79-109),
 * PlannerCar._get_next_control (planner_car.py:54-85) and Car.step
 * (car.py:76-87).
 *
 *   init_states  [N, 4]  ego initial states (fp32)
 *   cand_weights [P, D]  planner weights per candidate, fp32, ALREADY
 *                normalised the way the reference does it on the host
 *                (mpc_ord.py:71,120, linear_reward_car.py:47)
 *   returns_out  [ep_end-ep_begin]           fp32 sample_reward per episode
 *   traj_out     [ep_end-ep_begin, T+1, C, 4] world states (may be NULL)
 *   ctrl_out     [ep_end-ep_begin, T, 2]      applied ego controls (may be NULL)
 */
int32_t ocd_rollout_episodes(const ocd_scenario *scn,
                             const float *init_states,
                             const float *cand_weights,
                             int64_t P, int64_t N,
                             int64_t ep_begin, int64_t ep_end,
                             float *returns_out, float *traj_out, float *ctrl_out,
                             void *hip_stream);

/*
 * Independent populations in ONE launch: the reference runs one optimisation per init group in a multiprocessing.Pool
 * (experiments/run_mpc_ord.py:83-90; 28 chosen weight vectors x 32 held-out inits in generalization_data.py:78-107), each
 * process looping MPC_ORD.eval_weights_for_init (mpc_ord.py:67-106) over ITS candidates and ITS init states.  Here the
 * episodes of all those loops are rows of one index and run as one batch:
 *
 *   init_states   [N_rows, 4]   every run's init states, concatenated
 *   cand_weights  [P_rows, D]   every run's candidates, concatenated (normalised fp32 as for ocd_rollout_episodes)
 *   episode_index [E, 3] int32  per episode: row of cand_weights, row of init_states, and the number of the world.reset()
 *                 call that starts it in ITS OWN sequential evaluation (= its flat index (p*N + n)*S + s inside its run,
 *                 plus that run's reset_phase): selects the entry of the teleport cycle exactly as ocd_rollout_episodes
 *                 does for episode e (reset % teleport_period, or reset % n_samples when the period is 0).
 *                 Device-addressable memory (device, or pinned / registered host).  A row outside [0, P_rows) / [0, N_rows)
 *                 or with a negative reset number is an ERROR, not clamped (ABI 3): an index in host memory is checked
 *                 before anything is launched -> OCD_ERR_INVALID_ARG naming the row; an index in device memory is checked by
 *                 the kernel -- that episode's return is NaN, and ocd_scenario_index_error (after the stream has been
 *                 waited for) or the next ocd_rollout_indexed on the handle returns OCD_ERR_INVALID_ARG naming the row.
 *   returns_out   [E]; traj_out [E, T+1, C, 4] or NULL; ctrl_out [E, T, 2] or NULL -- in index order.
 * Episode i is bit for bit the episode ocd_rollout_episodes computes for the same (candidate, init, reset).
 */
int32_t ocd_rollout_indexed(const ocd_scenario *scn,
                            const float *init_states, int64_t N_rows,
                            const float *cand_weights, int64_t P_rows,
                            const int32_t *episode_index, int64_t E,
                            float *returns_out, float *traj_out, float *ctrl_out,
                            void *hip_stream);

/* Device-memory indices of ocd_rollout_indexed: OCD_OK if no completed launch on this handle has met an out-of-range index
 * row since the last call, else OCD_ERR_INVALID_ARG with *bad_row (may be NULL) = the position of such a row in its index
 * (the error is cleared by reporting it).  Call it after the launch's stream has been waited for. */
int32_t ocd_scenario_index_error(const ocd_scenario *scn, int64_t *bad_row);

/*
 * The same episode loop started from arbitrary world states instead of
 * world.reset(): n_steps control steps from world step index `first_step`
 * (selects the scripted cars' plan entries, fixed_plan_car.py:25-31, and the
 * teleport step) with the reset outcome `sample`.  With n_steps = 1 this is
 * one CarWorld.step() (world.py:79-109): returns_out is the designer reward
 * of the pre-step state, ctrl_out the ego control PlannerCar._get_next_control
 * chose, traj_out[.,1] the state after Car.step.
 *
 *   world_state [B, C, 4]; weights [B, D] or [D] (weights_per_problem);
 *   returns_out [B]; traj_out [B, n_steps+1, C, 4] or NULL; ctrl_out [B, n_steps, 2] or NULL.
 */
int32_t ocd_rollout_from_state(const ocd_scenario *scn, const float *world_state,
                               const float *weights, int32_t weights_per_problem,
                               int32_t first_step, int32_t n_steps, int32_t sample,
                               float *returns_out, float *traj_out, float *ctrl_out,
                               int64_t B, void *hip_stream);

/*
 * The planner's objective and its gradient for caller-supplied controls: replaces
 * NaivePlanner.reward_func(init_state, controls, other_controls, weights) (naive_planner.py:33-77)
 * and tf.GradientTape over it (naive_planner.py:124-125,153).
 *   world_state [B, C, 4]; weights [B, D] or [D] (weights_per_problem);
 *   controls    [B, H, 2]  (clipped inside the dynamics exactly like the reference);
 *   other_plans [C-1, H, 2] or NULL (constant-velocity model);
 *   reward_out  [B]        R = sum over the horizon of car.reward_fn (or the terminal value)  (may be NULL)
 *   grad_out    [B, H, 2]  dR/dcontrols                                                      (may be NULL)
 *   traj_out    [B, H, 4]  the planning car's state after each horizon step                   (may be NULL)
 */
int32_t ocd_mpc_reward_batch(const ocd_scenario *scn, const float *world_state,
                             const float *weights, int32_t weights_per_problem,
                             const float *controls, const float *other_plans,
                             float *reward_out, float *grad_out, float *traj_out,
                             int64_t B, void *hip_stream);

/*
 * car_dynamics_step / next_car_state for B (state, control) pairs
 * (interact_drive/simulation_utils.py:9-21,73-123; get_dynamics_fn :321-326).
 *   states [B, 4], controls [B, 2], next_out [B, 4]; dt_sq = fp32(dt ** 2).
 */
int32_t ocd_dynamics_batch(const float *states, const float *controls, float dt, float dt_sq, float friction,
                           float *next_out, int64_t B, void *hip_stream);

/*
 * Reward features and reward of B world states (no planning).  Replaces
 * ThreeLaneTestCar.features / LinearRewardCar.reward_fn evaluated on a grid,
 * e.g. the visualiser's heat map (interact_drive/visualizer.py:211-238).
 *   world_state [B, C, 4]; weights [D] ; feats_out [B, D] (may be NULL);
 *   reward_out [B] (may be NULL).
 */
int32_t ocd_reward_batch(const ocd_scenario *scn,
                         const float *world_state, const float *weights,
                         float *feats_out, float *reward_out,
                         int64_t B, void *hip_stream);

/* Device exp / sin / cos of n floats: lets the tests prove the device math is
 * bit-identical to the oracle's restatement.  in, exp_out, sin_out, cos_out: [n]. */
int32_t ocd_debug_math(const float *in, float *exp_out, float *sin_out, float *cos_out,
                       int64_t n, void *hip_stream);

/* The two-wide ("packed", v_pk_*_f32) division and exponential cores of the reward features beside their scalar
 * forms, on caller-supplied operands: lets the tests prove on the device that the unpadded packed instruction
 * sequences the planner kernels run give the bits of the scalar ones (IEEE division; the contract's exp for
 * arguments <= 1), denormal numerators and quotients included.  All arrays hold 2 * n_pairs floats (pair i =
 * elements 2i, 2i+1); num/den or x may be NULL to skip that half; any output may be NULL. */
int32_t ocd_debug_packed_math(const float *num, const float *den, const float *x, float *div_scalar_out,
                              float *div_packed_out, float *exp_scalar_out, float *exp_packed_out, int64_t n_pairs,
                              void *hip_stream);

/* The shortened divisions of the reward features (csrc/ocd_devmath.h: recip_pair_guarded, quot2_by_recip with
 * refined_recip) on caller-supplied operands, so that the tests can hold them against correctly rounded division over the
 * whole operand range their guards admit (and see where they stop being exact beyond it).  All arrays hold 2 * n_pairs
 * floats.  u -> m_out = -1/u, k_out = (-m)/u;  (n, w) -> q_out = n / w through the refined reciprocal of w.  u or (n, w)
 * may be NULL to skip that half. */
int32_t ocd_debug_guarded_division(const float *u, const float *n, const float *w, float *m_out, float *k_out,
                                   float *q_out, int64_t n_pairs, void *hip_stream);

/* The kernels evaluate the reward features (merging.py:44-83) and their adjoint through several hand-written forms that
 * must agree bit for bit wherever their preconditions hold (csrc/ocd_device.h: reward_state -- the definition --,
 * reward_one with full / shortened divisions and with sub-skips, reward_fc, reward_fcc, the work-item form, reward_two).  This entry evaluates ALL of them
 * on B caller-supplied world states, so that the tests can hold them to each other form by form:
 *   world_state [B, C, 4]; weights [D];
 *   out   [B, 11, 5] per form (r, dr/dx, dr/dy, dr/dv, dr/dheading); forms: 0 reward_state, 1 reward_one (straight line, full
 *                    divisions), 2 reward_one (shortened divisions), 3 reward_one (sub-skips), 4 reward_fc, 5 reward_fc
 *                    (shortened), 6 reward_every, 7 reward_every (shortened), 8 work items (reward_base_grad + feature_item_grad
 *                    through an LDS list, the gradient passes of the chunked kernel's shared-SIMD builds; adjoint only: its r
 *                    repeats form 0; at most two scripted cars), 9 / 10 reward_two (two scripted cars: at most two active
 *                    features per lane -- the latency builds' steps with a lane inside both cars' boxes, round 6; full /
 *                    reciprocal quotients by the bump widths; adjoint only);
 *   valid [B, 11]    1 where the form's precondition holds for that state (one active feature, guards of the shortened
 *                    divisions, ...): only those entries are defined.
 * Lane-feature reward with (scripted cars, lanes) in {(1,2), (1,3), (2,2), (2,3), (3,3)}. */
int32_t ocd_debug_feature_variants(const ocd_scenario *scn, const float *world_state, const float *weights, float *out,
                                   int32_t *valid, int64_t B, void *hip_stream);

/* Block until everything queued on `hip_stream` (NULL = the default stream) has finished: hipStreamSynchronize behind
 * the C ABI, for host code that holds no HIP headers -- the native CMA-ES generation loop of include/ocd_cma.h waits for
 * its episode launch through this (mpc_ord.py:41: pycma's serial loop around the fitness callable). */
int32_t ocd_stream_synchronize(void *hip_stream);

/* Timing helper used by bench.py: runs `reps` back-to-back launches of
 * ocd_rollout_episodes on `hip_stream`, bracketed by HIP events recorded on
 * that same stream, and returns the mean milliseconds per launch in *ms_out
 * (host pointer).  Synchronises the stream. */
int32_t ocd_time_rollout(const ocd_scenario *scn,
                         const float *init_states, const float *cand_weights,
                         int64_t P, int64_t N, int64_t ep_begin, int64_t ep_end,
                         float *returns_out, int32_t reps, float *ms_out,
                         void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* OCD_H */
