/*
 * oracle/ocd_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement (plain C, IEEE binary32, one rounding per reference
 * TensorFlow op) of the reference's receding-horizon planner path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (l4dc-mpc-ocd_amd/) never does.
 *
 * Pinning status:
 *   - dynamics, bump / threshold primitives, planner: pinned by the
 *     reference's own known-answer tests (tests/test_oracle_kat.py);
 *   - ThreeLaneTestCar features, scenarios, episode returns: PARITY UNPINNED
 *     by the reference (it has no test or fixture for them and TensorFlow is
 *     not installable here); cross-checked instead against an independent
 *     torch-autograd restatement (tests/torch_restatement.py).
 */
#ifndef OCD_ORACLE_H
#define OCD_ORACLE_H

#include "../include/ocd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* math_utils.py / simulation_utils.py primitives, for the known-answer tests */
float ocd_oracle_expf(float x);
float ocd_oracle_sinf(float x);
float ocd_oracle_cosf(float x);
/* _f(x, shape)  (math_utils.py:7-31) */
float ocd_oracle_f(float x, float shape);
/* smooth_threshold(threshold, width, c)(x)  (math_utils.py:59-97); lo = fp32(threshold - width) */
float ocd_oracle_smooth_threshold(float x, float lo, float width, float shape);
/* smooth_bump(start, end)(x)  (math_utils.py:135-180) */
float ocd_oracle_smooth_bump(float x, float start, float end);
/* car_dynamics_step (simulation_utils.py:9-21): state[4] -> out[4] */
void ocd_oracle_dynamics_step(const float *state, const float *control, float dt, float dt_sq,
                              float friction, float *out);

/* ThreeLaneTestCar.features + reward (merging.py:32-83, linear_reward_car.py:49-55)
 * of one world state [C,4]; feats_out[D] and grad_out[4] (d reward / d ego state)
 * may be NULL.  Returns the reward.  Without grad_out this is reward_fn as an episode is SCORED (mpc_ord.py:99; the
 * lane offset carries dist2median's y-term, world.py:216-217: NaN for a non-finite y); with grad_out it is the planner's
 * form (no y-term; identical for every finite y). */
float ocd_oracle_reward(const ocd_scenario_desc *d, const float *world_state, const float *weights,
                        float *feats_out, float *grad_out);

/* mpc_reward (naive_planner.py:33-77) and its gradient w.r.t. the controls.
 * world_state [C,4], controls [H,2], other_plans [C-1,H,2] or NULL,
 * grad_out [H,2] or NULL, traj_out [H,4] ego post-step states or NULL. */
float ocd_oracle_mpc_reward(const ocd_scenario_desc *d, const float *world_state,
                            const float *weights, const float *controls,
                            const float *other_plans, float *grad_out, float *traj_out);

/* Terminal value used by ocd_oracle_mpc_reward and the CPU twins below (process-global, test
 * infrastructure): ocd_scenario_set_leaf_value of include/ocd.h; values == NULL removes it.
 * The pointers are kept, not copied. */
void ocd_oracle_set_leaf_value(const float *grid0, int32_t n0, const float *grid1, int32_t n1,
                               const float *grid2, int32_t n2, const float *values, int32_t proj_kind);

/* The planning car's own current speed per problem for the extra_inits control initialisations (naive_planner.py:114
 * reads self.car.state[2], not the init_state argument); NULL (default): the world state's ego speed.  Applies to
 * ocd_plan_batch_cpu; the pointer must stay valid until it is reset. */
void ocd_oracle_set_init_speed(const float *init_speed);

/* CPU twins of the device entry points of include/ocd.h (HOST pointers). */
int32_t ocd_plan_batch_cpu(const ocd_scenario_desc *d,
                           const float *world_state,
                           const float *weights, int32_t weights_per_problem,
                           const float *other_plans,
                           float *plans_out, float *best_loss_out, int32_t *best_init_out,
                           float *all_plans_out, float *all_losses_out,
                           int64_t B, int32_t n_threads);

int32_t ocd_rollout_episodes_cpu(const ocd_scenario_desc *d,
                                 const float *init_states,
                                 const float *cand_weights,
                                 int64_t P, int64_t N,
                                 int64_t ep_begin, int64_t ep_end,
                                 float *returns_out, float *traj_out, float *ctrl_out,
                                 int32_t n_threads, int32_t reset_phase);

int32_t ocd_rollout_from_state_cpu(const ocd_scenario_desc *d, const float *world_state,
                                   const float *weights, int32_t weights_per_problem,
                                   int32_t first_step, int32_t n_steps, int32_t sample,
                                   float *returns_out, float *traj_out, float *ctrl_out, int64_t B);

int32_t ocd_reward_batch_cpu(const ocd_scenario_desc *d,
                             const float *world_state, const float *weights,
                             float *feats_out, float *reward_out, int64_t B);

/* 1 when built with -DOCD_USE_LIBM (glibc expf/sinf/cosf), else 0 */
int32_t ocd_oracle_uses_libm(void);

#ifdef __cplusplus
}
#endif
#endif
