"""Second opinion on EPISODES: a float64 torch-autograd restatement of the reference's fitness loop (TEST ONLY).

Written from the reference's Python, NOT from the C oracle, NOT from the kernels and NOT from
l4dc_mpc_ocd_amd.scenarios: every scenario constant below is typed from the reference file it cites, the world
keeps the reference's own mutable state (the `unlucky_car_idx` toggle of ReplanningCarWorld.reset, FixedPlanCar.t),
and the weight vectors go through the reference's own chain of numpy normalisations.  Only the arithmetic is
batched: all episodes of a fixture advance together as leading tensor dimensions [E] (episodes) and [E, K]
(control initialisations), which changes no value -- episodes never interact.

What is restated (file:line relative to the reference tree):
  MPC_ORD.__init__ / eval_weights / eval_weights_for_init      interact_drive/reward_design/mpc_ord.py:16-31,67-151
  CarWorld.reset / step, lane geometry                           interact_drive/world.py:72-109,143-159,162-218
  ReplanningCarWorld.reset / step                                experiments/replanning_world.py:11-36
  Car.reset / step, set_next_control                             interact_drive/car/car.py:70-123
  FixedPlanCar (t, plan[t] else default_control)                 interact_drive/car/fixed_plan_car.py:19-39
  FixedControlCar / FixedVelocityCar (friction 0, control 0)     interact_drive/car/fixed_control_car.py:26-36, fixed_velocity_car.py:18-24
  LinearRewardCar (weights setter, reward_fn)                    interact_drive/car/linear_reward_car.py:34-55
  PlannerCar._get_next_control (other_plans from index 0)        interact_drive/car/planner_car.py:54-85
  NaivePlanner.mpc_reward / generate_plan                        interact_drive/planner/naive_planner.py:33-77,107-164
  car_dynamics_step                                              interact_drive/simulation_utils.py:9-21
  ThreeLaneTestCar.features, smooth_bump, smooth_threshold, _f   experiments/merging.py:51-83, interact_drive/math_utils.py:28-31,87-95,166-178
  scenario factories                                             mpc_ord.py:162-207, local_opt_scenario.py:6-55, replanning_world.py:38-95, merging.py:86-99
  eval horizons, samples, tuned weights                          experiments/run_mpc_ord.py:19-44

TensorFlow's gradient conventions only matter at exact ties (minimum / maximum / reduce_min / reduce_max / clip at
the bound); the forms below follow them anyway: `where(a <= b, a, b)` hands a tie to the first argument, amin /
amax split a tie equally.
"""
import math

import numpy as np
import torch

PI_2 = np.pi / 2


# ----------------------------------------------------------------------------------------------------------------
# scenario constants, typed from the reference
# ----------------------------------------------------------------------------------------------------------------
class _Lane:
    """StraightLane (world.py:162-204): median through p and q, unit normal n, shifted() along n."""

    def __init__(self, p, q, w):
        self.p, self.q, self.w = np.asarray(p, dtype=np.float64), np.asarray(q, dtype=np.float64), w
        m = (self.q - self.p) / np.linalg.norm(self.q - self.p)
        self.n = np.asarray([-m[1], m[0]])

    def shifted(self, k):
        return _Lane(self.p + self.n * self.w * k, self.q + self.n * self.w * k, self.w)


def _three_lanes():                                   # ThreeLaneCarWorld, world.py:149-151
    lane = _Lane((0.0, -5.), (0.0, 10.), 0.1)
    return [lane.shifted(1), lane, lane.shifted(-1)]


def _two_lanes():                                     # TwoLaneCarWorld, world.py:157-158
    lane = _Lane((-0.05, -5.), (-0.05, 10.), 0.1)
    return [lane, lane.shifted(-1)]


def _fixed_velocity(init):                            # FixedVelocityCar: control (0, 0), friction 0.
    return dict(init=np.array(init, dtype=np.float64), friction=0.0, plan=None, control=(0.0, 0.0), default=None)


def _fixed_plan(init, plan, default):                 # FixedPlanCar: Car's default friction 0.2 (car.py:33)
    return dict(init=np.array(init, dtype=np.float64), friction=0.2, plan=[tuple(u) for u in plan], control=None,
                default=tuple(default))


def finite_horizon(horizon=5, extra_inits=False):
    """finite_horizon_env (mpc_ord.py:162-207); run_mpc_ord.py:29-36."""
    return dict(
        name="finite_horizon", lanes=_three_lanes(), num_lanes=3, target_speed=1.0, friction=0.2, horizon=horizon,
        car_weights=np.array([-5, 0., 0., 0., -6., -50, -50]),
        n_iter=200 if horizon == 6 else 100, extra_inits=extra_inits, check_plans=False,
        others=[_fixed_velocity([0, -0.6, 0.5, np.pi / 2])],
        eval_horizon=15, num_samples=1, critical_t=None,
        init_dist=((0, 0.04, (-0.1, 0.1)), (-0.9, 0.02, (-0.95, -0.85)), (0.8, 0.03, (0.7, 0.9))),
        tuned_weights=np.array([-0.21963165, -0.01184596, 0.34379187, -0.04687411, -0.06364365,
                                -0.54138792, -0.7308079]))


def local_opt(horizon=5, extra_inits=False):
    """local_opt_env (local_opt_scenario.py:6-55; its horizon is the literal 5); run_mpc_ord.py:20-27."""
    weights = np.array([-5, 0., 0., -10, 0, -50, -50])
    return dict(
        name="local_opt", lanes=_three_lanes(), num_lanes=3, target_speed=1.0, friction=0.2, horizon=horizon,
        car_weights=weights / np.linalg.norm(weights),
        n_iter=100, extra_inits=extra_inits, check_plans=False,
        others=[_fixed_velocity([0, -0.9, 1., np.pi / 2])],
        eval_horizon=15, num_samples=1, critical_t=None,
        init_dist=((-0.1, 0.005, (-0.12, -0.08)), (-0.9, 0.04, (-1., -0.8)), (1.0, 0.03, (0.9, 1.1))),
        tuned_weights=np.array([-0.09686739, 0.25720383, -0.58355971, -0.23075428, -0.41237239,
                                -0.4758984, -0.36625558]))


def replanning(horizon=5):
    """setup_world (replanning_world.py:38-95): og_weights normalised in float32; run_mpc_ord.py:37-43."""
    og_weights = np.array([-3, 0, 0, -2, -10, -10], dtype=np.float32)
    og_weights /= np.linalg.norm(og_weights)
    return dict(
        name="replanning", lanes=_two_lanes(), num_lanes=2, target_speed=1.2, friction=0.2, horizon=horizon,
        car_weights=og_weights, n_iter=100, extra_inits=False, check_plans=True,
        others=[_fixed_plan([0., -0.7, 0.8, np.pi / 2], [(0., 0.), (0.7, 2.7), (0., 0.), (0.0, -2.7)], (0.0, 0.0)),
                _fixed_plan([0., -0.7, 0.8, np.pi / 2], [(0., 0.), (0.7, -2.7), (0., 0.), (0.0, 2.7)], (0.0, 0.0))],
        eval_horizon=20, num_samples=2, critical_t=4,
        init_dist=((-0.0, 0.02, (-0.005, 0.005)), (-0.9, 0.04, (-1., -0.8)), (1.0, 0.05, (0.8, 1.2))),
        tuned_weights=np.array([-0.55899817, -0.4436692, -0.3724511, -0.19964276, -0.5438697, 0.12770043]))


def merging(horizon=5):
    """setup_world (merging.py:86-99).  The reference never evaluates this world through MPC_ORD: 15 steps as in its
    main() (merging.py:120), one sample; no init distribution and no tuned weights exist."""
    return dict(
        name="merging", lanes=_three_lanes(), num_lanes=3, target_speed=1.0, friction=0.2, horizon=horizon,
        car_weights=np.array([-1, 0., 0., -10., -10., -10, -5]),
        n_iter=100, extra_inits=False, check_plans=False,
        others=[_fixed_velocity([0.1, -1.8, 0.8, np.pi / 2]), _fixed_velocity([0.1, -1.3, 0.8, np.pi / 2])],
        eval_horizon=15, num_samples=1, critical_t=None, init_dist=None, tuned_weights=None,
        default_init=np.array([0, -1.8, 0.8, np.pi / 2]))


SCENARIOS = dict(finite_horizon=finite_horizon, local_opt=local_opt, replanning=replanning, merging=merging)


def init_state_of_seed(spec, env_seed):
    """get_init_state (mpc_ord.py:171-181 and its two copies): three truncated normals under np.random.seed.
    The stream is scipy-version dependent -- fixtures store the states themselves."""
    import scipy.stats
    np.random.seed(seed=env_seed)
    out = []
    for mean, std, rang in spec["init_dist"]:
        a, b = (rang[0] - mean) / std, (rang[1] - mean) / std
        out.append(np.squeeze(scipy.stats.truncnorm.rvs(a, b) * std + mean))
    return np.array(out + [np.pi / 2])


# ----------------------------------------------------------------------------------------------------------------
# weights: the reference's numpy chain
# ----------------------------------------------------------------------------------------------------------------
def car_weights_fp32(weights):
    """LinearRewardCar: tf.Variable(weights / np.linalg.norm(weights), dtype=tf.float32) (linear_reward_car.py:34),
    and the setter of `car.weights` does the same assign (:45-47)."""
    w = np.asarray(weights)
    return (w / np.linalg.norm(w)).astype(np.float32)


def designer_weights_of(spec):
    """MPC_ORD.__init__: car.weights / np.linalg.norm(car.weights) on the float32 `weights_tf.numpy()` (mpc_ord.py:24)."""
    car_w = car_weights_fp32(spec["car_weights"])
    return car_w / np.linalg.norm(car_w)


def planner_weights_of(candidate):
    """eval_weights (mpc_ord.py:115-120) -> eval_weights_for_init (:69-71) -> car.weights = weights (:83)."""
    w = candidate
    if isinstance(w, list):
        w = np.array(w)
    if w.ndim == 2:
        w = w[0]
    w = w / np.linalg.norm(w)
    if w.ndim == 2:
        w = w[0]
    w = w / np.linalg.norm(w)
    return car_weights_fp32(w)


# ----------------------------------------------------------------------------------------------------------------
# arithmetic (batched over leading dimensions)
# ----------------------------------------------------------------------------------------------------------------
class _Constants:
    """How a Python-float constant of the reference meets a tensor.  TensorFlow converts it to the tensor's dtype,
    float32 -- `dt`, `dt ** 2` (squared in Python first), `friction`, 0.08, `threshold - width` ... are float32
    numbers in the traced graph.  A float32 torch run does the same conversion by itself.  A float64 run has two
    meanings: `exact` (the default: the Python doubles as typed -- the mathematically intended expressions), and
    `float32` (K.round32 = True: the reference's OWN constants, the float32 roundings, carried in double arithmetic:
    "the reference's graph without rounding error").  The second is what a float64 build of any float32
    implementation computes, so two such builds can be compared far below 1e-4 at horizons where the 1e-8 between
    fp32(0.1) and 0.1 is amplified past it (H >= 10)."""
    round32 = False

    def __call__(self, x):
        return float(np.float32(x)) if self.round32 else x


K = _Constants()


def _f(x, shape):                                                       # math_utils.py:28-31
    pos = x > 0
    x_clipped = torch.where(pos, x, torch.zeros_like(x) + 0.01)
    return torch.where(pos, torch.exp(-1 / (K(shape) * x_clipped)), torch.zeros_like(x))


def _smooth_threshold(x, threshold, width, c=5.):                       # math_utils.py:85-95
    shape = c / width
    x_diff = x - K(threshold - width)
    return _f(x_diff, shape) / (_f(x_diff, shape) + _f(K(width) - x_diff, shape))


def _smooth_bump(x, start, end):                                        # math_utils.py:166-178
    width = (end - start) / 2
    center = (start + end) / 2
    x_norm = (x - center) / width
    cond = torch.square(x_norm) < 1
    x_norm_clipped = torch.where(cond, x_norm, torch.zeros_like(x_norm))
    return torch.where(cond, torch.exp(-1 / (1 - x_norm_clipped ** 2) + 1), torch.zeros_like(x_norm))


def _dynamics(s, acc, ang_vel, dt, friction):                           # simulation_utils.py:9-21
    x, y, v, angle = s[..., 0], s[..., 1], s[..., 2], s[..., 3]
    four = torch.full_like(acc, 4.)
    acc = torch.where(acc <= four, acc, four)                           # tf.minimum(acc, 4.)
    acc = torch.where(acc >= -2 * four, acc, -2 * four)                 # tf.maximum(., -2*4.)
    ang_vel = torch.where(ang_vel <= four, ang_vel, four)
    ang_vel = torch.where(ang_vel >= -four, ang_vel, -four)
    total_acc = acc - K(friction) * v ** 2
    distance_travelled = v * K(dt) + 0.5 * total_acc * K(dt ** 2)
    return torch.stack([x + torch.cos(angle) * distance_travelled, y + torch.sin(angle) * distance_travelled,
                        v + total_acc * K(dt), angle + ang_vel * K(dt)], dim=-1)


class _Sim:
    def __init__(self, spec, dtype):
        self.spec, self.dtype = spec, dtype
        self.dt = 0.1                                                   # CarWorld(dt=0.1), world.py:19
        ts = np.float32(spec["target_speed"])                           # merging.py:29
        self.target_speed = float(ts)
        self.speed_bound = float(4 * ts ** 2)                           # merging.py:59, float32 scalar arithmetic
        self.n_others = len(spec["others"])
        self.keep_plans = None                                          # a list: (plans, losses, dR/du) per control step

    def features(self, ego, others):
        """ThreeLaneTestCar.features (merging.py:51-83); ego [..., 4], others [..., n_others, 4] -> [..., D]."""
        sp = self.spec
        feats, lane_dists = [], []
        velocity = ego[..., 2] * torch.sin(ego[..., 3])
        sq = (velocity - self.target_speed) ** 2
        bound = torch.full_like(sq, self.speed_bound)
        feats.append(torch.where(sq <= bound, sq, bound))
        for lane in sp["lanes"]:
            r = (ego[..., 0] - K(lane.p[0])) * K(lane.n[0]) + (ego[..., 1] - K(lane.p[1])) * K(lane.n[1])   # world.py:216-218
            d = r ** 2 * 10
            lane_dists.append(d)
            feats.append(d)
        feats.append(torch.amin(torch.stack(lane_dists, dim=0), dim=0))
        collision = []
        for j in range(self.n_others):                                  # the ego's own bump is computed and dropped
            o = others[..., j, :]
            collision.append(_smooth_bump(ego[..., 0], o[..., 0] - K(0.08), o[..., 0] + K(0.08))
                             * _smooth_bump(ego[..., 1], o[..., 1] - K(0.15), o[..., 1] + K(0.15)))
        feats.append(torch.amax(torch.stack(collision, dim=0), dim=0))
        thr, x = 0.05 * sp["num_lanes"], ego[..., 0]
        feats.append((_smooth_threshold(x, thr, width=0.05) + _smooth_threshold(-x, thr, width=0.05)) * torch.abs(x))
        return torch.stack(feats, dim=-1)

    def mpc_reward(self, ego, others, controls, other_controls, weights):
        """mpc_reward (naive_planner.py:44-77).  ego [E, 1, 4], others [E, 1, n, 4], controls [E, K, H, 2],
        other_controls [n, H, 2] or None, weights [E, 1, D] -> R [E, K]."""
        dt, r = self.dt, 0
        for t in range(self.spec["horizon"]):
            ego = _dynamics(ego, controls[:, :, t, 0], controls[:, :, t, 1], dt, self.spec["friction"])
            v, angle = others[..., 2], others[..., 3]
            if other_controls is not None:
                acc, ang_vel = other_controls[:, t, 0], other_controls[:, t, 1]
                update = torch.stack([torch.cos(angle) * (v * K(dt) + 0.5 * acc * K(dt ** 2)),
                                      torch.sin(angle) * (v * K(dt) + 0.5 * acc * K(dt ** 2)),
                                      acc * K(dt) + torch.zeros_like(v), ang_vel * K(dt) + torch.zeros_like(v)], dim=-1)
            else:
                update = torch.stack([torch.cos(angle) * v * K(dt), torch.sin(angle) * v * K(dt),
                                      torch.zeros_like(v), torch.zeros_like(v)], dim=-1)
            others = others + update
            r = r + torch.sum(weights * self.features(ego, others), dim=-1)
        return r

    def generate_plan(self, ego, others, other_controls, weights):
        """generate_plan (naive_planner.py:107-164).  ego [E, 4], others [E, n, 4], weights [E, D].
        Returns plans [E, K, H, 2], losses [E, K], best [E]."""
        sp, dtype = self.spec, self.dtype
        E, H = ego.shape[0], sp["horizon"]
        cols = [(torch.zeros(E, dtype=dtype), 0.0), (torch.zeros(E, dtype=dtype), K(-5 * 0.13)),
                (torch.zeros(E, dtype=dtype), K(5 * 0.13))]
        if sp["extra_inits"]:
            coast = K(sp["friction"]) * ego[:, 2] ** 2                  # self.car.friction * self.car.state[2] ** 2
            cols += [(coast, 0.0), (coast, K(-5 * 0.13)), (coast, K(5 * 0.13))]
        u = torch.stack([torch.stack([a, torch.full((E,), w, dtype=dtype)], dim=-1) for a, w in cols], dim=1)
        u = u[:, :, None, :].repeat(1, 1, H, 1)                         # [E, K, H, 2]
        e1, o1, w1 = ego[:, None, :], others[:, None, :, :], weights[:, None, :]
        lr = 0.1                                                        # NaivePlanner(learning_rate=0.1)
        for _ in range(sp["n_iter"]):
            u = u.detach().requires_grad_(True)
            loss = -self.mpc_reward(e1, o1, u, other_controls, w1)
            (g,) = torch.autograd.grad(loss.sum(), u)
            u = u - K(lr) * g                                           # keras SGD, no momentum
        u = u.detach()
        losses = -self.mpc_reward(e1, o1, u, other_controls, w1)
        if self.keep_plans is not None:                                 # the objective's gradient at the END points too
            ug = u.clone().requires_grad_(True)
            (g,) = torch.autograd.grad(self.mpc_reward(e1, o1, ug, other_controls, w1).sum(), ug)
            self.keep_plans.append((u.numpy().copy(), losses.numpy().copy(), g.numpy().copy()))
        best = torch.zeros(E, dtype=torch.long)                         # losses.index(min(losses)): first index wins
        cur = losses[:, 0]
        for k in range(1, losses.shape[1]):
            better = losses[:, k] < cur
            best = torch.where(better, torch.full_like(best, k), best)
            cur = torch.where(better, losses[:, k], cur)
        return u, losses, best


class _WorldBookkeeping:
    """The mutable integers of the reference's world that decide WHAT an episode simulates: which car
    ReplanningCarWorld removes (toggled on every reset, replanning_world.py:24-27; the factory resets once, :93)."""

    def __init__(self, spec):
        self.replanning = spec["critical_t"] is not None
        self.unlucky_car_idx = 1                                        # ReplanningCarWorld.__init__
        self.reset()                                                    # setup_world(): world.reset()

    def reset(self):
        if self.replanning:
            self.unlucky_car_idx = 2 if self.unlucky_car_idx == 1 else 1
        return self.unlucky_car_idx if self.replanning else 0


def run(spec, init_states, candidates, dtype=torch.float64, constants="exact", keep_plans=False, nudge=0.0):
    """constants="float32": the reference's own (float32-rounded) constants in a float64 run, see _Constants.
    keep_plans: also return, per control step, what generate_plan ended on for EVERY control initialisation --
        final_plans [E, T, K, H, 2], final_losses [E, T, K] and final_grad [E, T, K, H, 2] = dR/du at those end points
        (one objective + gradient evaluation per (world state, control sequence): nothing iterated, so two float64
        implementations can be compared on them at any horizon).
    nudge: multiply every ego init state by (1 + nudge) AFTER its float32 cast (1e-13: a millionth of float32's
        resolution).  The generator uses it to let torch ITSELF say which episodes a float64 run determines: where a
        hundred SGD steps amplify 1e-13 to 1e-3 no second implementation can be expected to land on the same numbers.

    Every episode MPC_ORD would run for `candidates` (list of weight vectors as pycma hands them over) on
    `init_states` [N, 4], in the reference's order: eval_weights per candidate -> inits -> samples.

    Returns a dict of numpy arrays (E = P * N * S episodes, flat index (p * N + n) * S + s):
      states [E, T+1, C, 4]   world.state after reset, then the third value of every world.step()
      past   [E, T, C, 4]     the first value of every world.step() (after a teleport, if any)
      controls [E, T, 2]      the planning car's applied control
      chosen [E, T]           index of the control initialisation generate_plan kept
      margin [E, T]           gap between the best and the second-best loss (how decided the argmin was)
      sample_reward [E]       mpc_ord.py:93-99
      designer_reward [P, N]  mpc_ord.py:86-104 (sum over samples)
      cost [P]                eval_weights' return value (mpc_ord.py:126-151)
      removed [E]             car index the world teleported away (0 = none)
      planner_w32 [P, D], designer_w32 [D]
    """
    assert constants in ("exact", "float32")
    K.round32 = constants == "float32"
    try:
        return _run(spec, init_states, candidates, dtype, keep_plans, nudge)
    finally:
        K.round32 = False


def _run(spec, init_states, candidates, dtype, keep_plans=False, nudge=0.0):
    sim = _Sim(spec, dtype)
    if keep_plans:
        sim.keep_plans = []
    book = _WorldBookkeeping(spec)
    init_states = np.asarray(init_states, dtype=np.float64)
    P, N, S, T = len(candidates), len(init_states), spec["num_samples"], spec["eval_horizon"]
    C = 1 + sim.n_others
    designer_w = designer_weights_of(spec)
    planner_w, removed, ego0, w_rows = [], [], [], []
    for p in range(P):
        pw = planner_weights_of(np.asarray(candidates[p], dtype=np.float64))
        planner_w.append(pw)
        for n in range(N):
            init32 = init_states[n].astype(np.float32)                  # tf.constant(init, dtype=tf.float32)
            for _ in range(S):
                removed.append(book.reset())                            # world.reset() of this sample
                ego0.append(init32)
                w_rows.append(pw)
    E = P * N * S
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), dtype=dtype)  # noqa: E731
    ego = t(np.stack(ego0)) * (1 + nudge)                               # Car.reset: state = init_state
    others = t(np.stack([np.asarray(o["init"]).astype(np.float32) for o in spec["others"]]))[None].repeat(E, 1, 1)
    weights = t(np.stack(w_rows))
    dw = t(designer_w)
    removed_t = torch.tensor(removed, dtype=torch.long)
    other_controls = None
    if spec["check_plans"]:                                             # planner_car.py:58-80: always from plan[0]
        rows = []
        for o in spec["others"]:
            row = []
            for j in range(spec["horizon"]):
                if o["plan"] is not None:
                    if j < len(o["plan"]):
                        row.append(o["plan"][j])
                    elif o["default"] is not None:
                        row.append(o["default"])
                    else:
                        row.append((0., 0.))
                else:
                    row.append((0., 0.))
            rows.append(row)
        other_controls = t(np.asarray(rows, dtype=np.float32))
    car_t = [0] * sim.n_others                                          # FixedPlanCar.t after reset
    states, past, controls, chosen, margin = [torch.cat([ego[:, None], others], dim=1)], [], [], [], []
    sample_reward = torch.zeros(E, dtype=dtype)
    world_t = 0
    for _ in range(T):
        world_t += 1                                                    # ReplanningCarWorld.step
        if spec["critical_t"] is not None and world_t == spec["critical_t"]:
            gone = t(np.array([10., 0., 0., 0.], dtype=np.float32))
            for j in range(sim.n_others):
                hit = (removed_t == j + 1)[:, None]
                others = torch.cat([others[:, :j], torch.where(hit, gone, others[:, j])[:, None], others[:, j + 1:]], dim=1)
        past.append(torch.cat([ego[:, None], others], dim=1))           # past_state = self.state
        plans, losses, best = sim.generate_plan(ego, others, other_controls, weights)
        u0 = plans[torch.arange(E), best][:, 0, :]                      # tf.identity(self.plan[0])
        srt = torch.sort(losses, dim=1).values
        chosen.append(best)
        margin.append(srt[:, 1] - srt[:, 0])
        # reward of the PRE-step state with the designer's weights (mpc_ord.py:99)
        sample_reward = sample_reward + torch.sum(dw * sim.features(past[-1][:, 0], past[-1][:, 1:]), dim=-1)
        new_others = []
        for j, o in enumerate(spec["others"]):
            if o["plan"] is not None:                                   # FixedPlanCar: control set at reset / last step
                u = o["plan"][car_t[j]] if car_t[j] < len(o["plan"]) else o["default"]
            else:
                u = o["control"]
            a = torch.full((E,), float(np.float32(u[0])), dtype=dtype)
            w = torch.full((E,), float(np.float32(u[1])), dtype=dtype)
            new_others.append(_dynamics(others[:, j], a, w, sim.dt, o["friction"]))
            car_t[j] += 1
        ego = _dynamics(ego, u0[:, 0], u0[:, 1], sim.dt, spec["friction"])
        others = torch.stack(new_others, dim=1)
        controls.append(u0)
        states.append(torch.cat([ego[:, None], others], dim=1))
    sr = sample_reward.numpy()
    designer_reward = sr.reshape(P, N, S).sum(axis=2)                   # designer_reward += sample_reward
    total = np.zeros(P)
    for p in range(P):                                                  # eval_weights: python float accumulation
        acc = 0
        for n in range(N):
            acc += designer_reward[p, n]
        total[p] = acc / S
    extra = {}
    if keep_plans:
        extra = dict(final_plans=np.stack([k[0] for k in sim.keep_plans], axis=1),
                     final_losses=np.stack([k[1] for k in sim.keep_plans], axis=1),
                     final_grad=np.stack([k[2] for k in sim.keep_plans], axis=1))
    return dict(extra, states=torch.stack(states, dim=1).numpy(), past=torch.stack(past, dim=1).numpy(),
                controls=torch.stack(controls, dim=1).numpy(), chosen=torch.stack(chosen, dim=1).numpy().astype(np.int32),
                margin=torch.stack(margin, dim=1).numpy(), sample_reward=sr, designer_reward=designer_reward,
                cost=-total, removed=np.asarray(removed, dtype=np.int32),
                planner_w32=np.stack(planner_w), designer_w32=np.asarray(designer_w, dtype=np.float32))
