/*
 * ocd_cma.h -- host side of a CMA-ES generation around the episode kernel, in native code (libocd_cma.so, plain C,
 * no HIP): ask, tell and the float64 fitness reduction.
 *
 * Replaces (reference file:line, relative to the reference tree)
 *   cma.evolution_strategy.fmin2(self.eval_weights, ...)    interact_drive/reward_design/mpc_ord.py:33-45
 *       (pycma: an unpinned, un-vendored dependency, setup.py:6 -- the sampling sequence here is NOT pycma's;
 *        optimisation traces are "parity unpinned")
 *   the reduction of the episode returns to a cost       interact_drive/reward_design/mpc_ord.py:102,126-151
 * The Python binding is l4dc-mpc-ocd_amd/interact_drive/reward_design/cmaes.py (NativeCMAES); MPC_ORD.optimize_cmaes
 * drives it with whole generations per kernel launch.
 *
 * Conventions: host pointers, caller-owned buffers, 0 = OK / -1 = invalid argument, no global state; one handle is
 * not thread-safe, different handles are independent.
 */
#ifndef OCD_CMA_H
#define OCD_CMA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCD_CMA_MAX_DIM 64
#define OCD_CMA_ABI_VERSION 8      /* 8: run_many shares the per-run host work (tells, next deviates, history rows) out to host_threads threads; 7: run_many in groups on several streams (n_groups, streams); 6: the tutorial's weights ln((lambda+1)/2) - ln i over all ranks, active update on by default, set_active, weights; 2: tell returns the non-finite count; resample, stop_state, abi_version; 3: normalise_weights; 4: stop, run; 5: 12 stop rules (noeffectaxis, noeffectcoord), add_evals, run_many */
#define OCD_CMA_N_STOP 12          /* termination rules of ocd_cma_stop */

typedef struct ocd_cma ocd_cma;

/* (mu/mu_w, lambda)-CMA-ES with the default strategy parameters of Hansen's tutorial (arXiv 1604.00772, Table 1:
 * weights ln((lambda+1)/2) - ln i, negative weights in the rank-mu update = pycma's CMA_active default), started at x0
 * [n] with step size sigma0;
 * popsize 0 = 4 + floor(3 ln n) (pycma's default: 9 for the 7 weights of ThreeLaneTestCar); the normal deviates are
 * numpy.random.RandomState(seed).standard_normal's stream, bit for bit. */
int32_t ocd_cma_create(int32_t n, const double *x0, double sigma0, int32_t popsize, uint32_t seed, ocd_cma **out);
void ocd_cma_destroy(ocd_cma *es);
int32_t ocd_cma_popsize(const ocd_cma *es);
/* The active update (negative recombination weights, tutorial eqs. 46-47, 50-53) on / off; on after create (pycma's
 * default).  Only before the first tell: -1 afterwards. */
int32_t ocd_cma_set_active(ocd_cma *es, int32_t on);
/* w [lambda] <- the recombination weights (positive first); consts [8] <- mueff, cc, csigma, c1, cmu, dsigma, chiN,
 * the sum of all weights.  Either may be NULL. */
int32_t ocd_cma_weights(const ocd_cma *es, double *w, double *consts);
/* OCD_CMA_ABI_VERSION of the built library: the binding refuses a stale libocd_cma.so. */
int32_t ocd_cma_abi_version(void);

/* Draw the normal deviates of the next population ahead of time (they do not depend on the state of the search):
 * for callers with something to wait for -- the running episode kernel.  Optional; the stream is the same. */
int32_t ocd_cma_prepare(ocd_cma *es);
/* X [lambda, n] <- the next population: mean + sigma * C^(1/2) z (symmetric square root). */
int32_t ocd_cma_ask(ocd_cma *es, double *X);
/* Row k of X <- one more candidate for slot k of the population last asked for (n fresh deviates): what pycma's
 * ask_and_eval does with a candidate whose cost came back NaN (rejection sampling around mpc_ord.py:41). */
int32_t ocd_cma_resample(ocd_cma *es, int32_t k, double *X);
/* The update of one generation from the population last asked for (X must be what ask / resample wrote: the update
 * uses the steps behind those rows) and its costs fitness [lambda] (lower is better).  Costs are ranked with NaN LAST
 * (numpy's argsort order; +inf just before); returns the number of non-finite costs (>= 0), or -1 on bad arguments. */
int32_t ocd_cma_tell(ocd_cma *es, const double *X, const double *fitness);
/* Count n evaluations made outside tell: candidates redrawn after a NaN cost (pycma's ask_and_eval counts every
 * evaluation, so `maxfevals` and the evaluation counter include the rejected ones). */
int32_t ocd_cma_add_evals(ocd_cma *es, int64_t n);
/* Any of the outputs may be NULL: mean [n], sigma, C [n, n], best_x [n], best_f, generations, evaluations,
 * max_axis = the largest sqrt-eigenvalue of C (the stopping rule sigma * max_axis < tolx). */
int32_t ocd_cma_state(const ocd_cma *es, double *mean, double *sigma, double *C, double *best_x, double *best_f,
                      int64_t *gen, int64_t *counteval, double *max_axis);

/* What the termination rules read after a tell (see ocd_cma.c): sigma, axis lengths, counters, the best / median /
 * worst cost of the population last told, non-finite counts, sigma * max sqrt(C_ii), sigma * max |pc_i|, min sqrt(C_ii). */
int32_t ocd_cma_stop_state(const ocd_cma *es, double out[13]);

/* returns [P, N, S] fp32 sample rewards -> cost_out [P]: samples summed sequentially in fp32 (TensorFlow scalars,
 * mpc_ord.py:102), inits sequentially in float64 (mpc_ord.py:126,137), / S, negated (mpc_ord.py:139,151). */
int32_t ocd_fitness_from_returns(const float *returns, int64_t P, int64_t N, int64_t S, double *cost_out);

/* pycma's termination rules on the state ocd_cma_tell keeps (the best costs of the last 10 + 30 n / lambda generations,
 * every generation's best / median cost, ...): cma.evolution_strategy.fmin2's default options around mpc_ord.py:41,
 * restated (pycma is absent: parity unpinned; reward_design/cmaes.py:_Termination is the same logic in Python, used
 * by the numpy twin -- the tests compare the two).  opts and flags in the order
 *   maxiter, maxfevals, tolfun, tolfunhist, tolx, tolfacupx, tolconditioncov, tolupsigma, tolstagnation, tolflatfitness,
 *   noeffectaxis, noeffectcoord (pycma has no option value for the last two: opts[10], opts[11] are ignored);
 * tolstagnation compares the newest l generations' best / median costs with the l just before them (pycma:
 * histbest[:l] vs histbest[l:2l], newest first);
 * flags[i] = 1 where condition i holds; returns how many hold (0 = go on), -1 on bad arguments. */
int32_t ocd_cma_stop(ocd_cma *es, const double opts[OCD_CMA_N_STOP], int32_t flags[OCD_CMA_N_STOP]);

/* The episode launch and the stream wait the native generation loop calls -- function pointers, so that this library
 * stays free of HIP: ocd_rollout_episodes and ocd_stream_synchronize of include/ocd.h. */
typedef int32_t (*ocd_cma_rollout_fn)(const void *scn, const float *init_states, const float *cand_weights, int64_t P,
                                      int64_t N, int64_t ep_begin, int64_t ep_end, float *returns_out, float *traj_out,
                                      float *ctrl_out, void *hip_stream);
typedef int32_t (*ocd_cma_sync_fn)(void *hip_stream);

typedef struct ocd_cma_run_args {
    const void *scn;              /* ocd_scenario handle */
    const float *init_dev;        /* [N, 4] device */
    int64_t N, S;                 /* inits, samples per init */
    float *w_pinned;              /* [lambda, n] fp32, pinned host memory the device addresses (HIP maps it): the kernel reads it */
    float *ret_pinned;            /* [lambda * N * S] fp32, pinned host memory the kernel writes its returns into */
    void *stream;
    ocd_cma_rollout_fn rollout;
    ocd_cma_sync_fn sync;
    int32_t normalise_variant;    /* ocd_normalise_weights: the dot order that reproduces numpy here */
    int32_t reserved;
    int64_t max_generations;      /* at most this many generations in this call */
    double stop_opts[OCD_CMA_N_STOP]; /* ocd_cma_stop */
    double *X;                    /* [lambda, n] the population (also the one a pending-NaN generation is left in) */
    double *cost;                 /* [lambda] its costs */
    double *hist_w;               /* [max_generations, lambda, n] rows normalised once (history entries), or NULL */
    double *hist_cost;            /* [max_generations, lambda], or NULL */
    double *seconds;              /* [max_generations, 8]: whole generation, ask, normalise, launch, overlapped host work, wait, reduce, tell + stop; or NULL */
    int32_t *nonfinite;           /* [max_generations] non-finite costs told, or NULL */
} ocd_cma_run_args;

/* Generations of MPC_ORD.optimize_cmaes (mpc_ord.py:33-45 around cma's fmin2) without returning to the interpreter:
 * ask, normalise into w_pinned, launch, (next deviates + history rows while the GPU works), wait, reduce, tell, stop.
 * Returns when a termination condition holds (stop_flags), after max_generations, or -- *pending_nan = 1 -- with a
 * generation evaluated but NOT told because a cost is NaN: the caller redraws those candidates as pycma does
 * (ocd_cma_resample), tells, and calls again.  *generations_done = generations told in this call. */
int32_t ocd_cma_run(ocd_cma *es, const ocd_cma_run_args *a, int64_t *generations_done, int32_t stop_flags[OCD_CMA_N_STOP],
                    int32_t *pending_nan);

/* R independent optimisations in lockstep, ONE episode launch per generation: what the reference spreads over a
 * multiprocessing.Pool -- one process per init group (experiments/run_mpc_ord.py:83-90), each looping pycma's ask /
 * fitness / tell on its own population and init states (mpc_ord.py:33-45).  The episodes of all active runs are rows of one
 * index and go through ocd_rollout_indexed (include/ocd.h); every run sees exactly the call sequence ocd_cma_run gives it
 * alone, so its candidates, costs and history are bit for bit those of the run alone. */
#define OCD_CMA_MAX_RUNS 256
typedef int32_t (*ocd_cma_rollout_indexed_fn)(const void *scn, const float *init_states, int64_t N_rows,
                                              const float *cand_weights, int64_t P_rows, const int32_t *episode_index,
                                              int64_t E, float *returns_out, float *traj_out, float *ctrl_out, void *hip_stream);

typedef struct ocd_cma_many_args {
    const void *scn;              /* ocd_scenario handle */
    const float *init_dev;        /* [N_rows, 4] device: every run's init states, concatenated */
    int64_t N_rows, P_rows, S;    /* init rows, weight rows (sum of the popsizes), samples per init */
    int32_t R;                    /* runs, <= OCD_CMA_MAX_RUNS; all of the same dimension n */
    int32_t normalise_variant;    /* ocd_normalise_weights */
    const int64_t *run_n0, *run_N;/* [R] first init row and number of inits of run r */
    const int64_t *run_p0;        /* [R] first row of run r in w_pinned (it owns popsize_r rows) */
    const int32_t *run_reset_phase; /* [R] world.reset() calls before run r's episode 0 (teleport cycle), or NULL = 0 */
    float *w_pinned;              /* [P_rows, n] fp32, pinned host memory the device addresses */
    int32_t *index_pinned;        /* [sum_r popsize_r N_r S, 3] pinned: rewritten here whenever a run drops out */
    float *ret_pinned;            /* [same] pinned: the launch's returns, active runs in order */
    void *stream;
    ocd_cma_rollout_indexed_fn rollout;   /* ocd_rollout_indexed */
    ocd_cma_sync_fn sync;                 /* ocd_stream_synchronize */
    int64_t max_generations;      /* at most this many lockstep generations in this call */
    const double *stop_opts;      /* [OCD_CMA_N_STOP] ocd_cma_stop options, the same for every run */
    uint8_t *active;              /* [R] in / out: 1 = run r still takes part (cleared when a termination rule holds) */
    double *const *X;             /* [R] pointers: run r's population [popsize_r, n] */
    double *const *cost;          /* [R] pointers: its costs [popsize_r] */
    double *hist_w;               /* [max_generations, P_rows, n] rows normalised once (history entries), or NULL */
    double *hist_cost;            /* [max_generations, P_rows], or NULL */
    uint8_t *evaluated;           /* [max_generations, R] 1 where run r took part in generation g (its rows are valid), or NULL */
    double *seconds;              /* [max_generations, 8] as ocd_cma_run_args.seconds (all runs together), or NULL */
    int32_t *nonfinite;           /* [max_generations, R] non-finite costs told, or NULL */
    int64_t *episodes_launched;   /* [max_generations] episodes of each generation's launch, or NULL */
    int32_t *stop_flags;          /* [R, OCD_CMA_N_STOP] out: the rules that ended run r in this call */
    uint8_t *pending_nan;         /* [R] out: run r's last generation is evaluated (X, cost, history rows) but NOT told: a
                                   * cost is NaN -- the caller redraws (ocd_cma_resample), tells it, and calls again */
    /* ---- ABI 7 ---- */
    int32_t n_groups;             /* 0 / 1: one launch per generation on `stream`.  G > 1 (<= 8): the runs are dealt to G groups of
                                   * neighbouring runs, group k launches on streams[k]; each group cycles wait -> tell -> ask -> launch
                                   * by itself and the groups take turns, so one group's host work runs under the others' kernels
                                   * (for launches of at most one wavefront per SIMD: they run side by side) */
    int32_t host_threads;         /* ABI 8 (the field ABI 7 reserved): 0 / 1: the calling thread does all host work.  T > 1 (<= 16): T - 1
                                   * workers live for the duration of the call and share the runs' tells (and the work that overlaps the
                                   * kernel) with the caller -- each run is told by one thread on its own state: same results for every T */
    void *const *streams;         /* [n_groups] HIP streams, ordered after whatever produced init_dev */
} ocd_cma_many_args;

/* Returns after max_generations, when no run is active any more, or once a run got a NaN cost (pending_nan) and whatever
 * was already launched has been reduced and told (with several groups that can be one more generation of the other groups:
 * `evaluated` says who took part in which).  *generations_done = lockstep generations executed in this call. */
int32_t ocd_cma_run_many(ocd_cma *const *es, const ocd_cma_many_args *a, int64_t *generations_done);

/* K fitness evaluations of one fixed population back to back -- launch (a->rollout on the P rows of a->w_pinned, taken
 * as they are), wait (a->sync), float64 reduction into cost_out [P] -- without the interpreter between them: what a
 * generation of ocd_cma_run spends outside ask / tell, and what bench.py times as its step in one process.  Uses scn,
 * init_dev, N, S, w_pinned, ret_pinned, stream, rollout, sync of the argument block. */
int32_t ocd_eval_generations(const ocd_cma_run_args *a, int64_t P, int64_t K, double *cost_out, double *seconds_out);

/* W [P, D] float64 candidate weights -> out [P, D] fp32 as the planning car gets them: three float64 normalisations
 * (mpc_ord.py:120,71; linear_reward_car.py:45-47) and the fp32 cast.  `variant` names the summation order of the
 * dot product behind np.linalg.norm on this machine (0: mul + add left to right, 1: fma left to right); the binding
 * (scenarios.planner_weights_fp32_batch) self-checks against numpy and falls back to numpy if neither matches. */
int32_t ocd_normalise_weights(const double *W, int64_t P, int64_t D, int32_t variant, float *out);

#ifdef __cplusplus
}
#endif
#endif
