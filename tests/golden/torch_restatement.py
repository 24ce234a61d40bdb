"""Independent restatement of the reference's planner objective in torch (TEST ONLY).

Written from the reference's Python (not from the C oracle): same expressions,
torch ops in place of tf ops, torch.autograd in place of tf.GradientTape.  It
is the second opinion on the oracle's hand-derived adjoint and on the feature
definitions the reference's own tests do not pin.

Reference expressions mirrored:
  interact_drive/math_utils.py:28-31 (_f), :87-95 (smooth_threshold), :166-178 (smooth_bump)
  interact_drive/simulation_utils.py:9-21 (car_dynamics_step)
  experiments/merging.py:44-83 (features), interact_drive/world.py:216-218 (dist2median)
  interact_drive/planner/naive_planner.py:44-77 (mpc_reward)
"""
import numpy as np
import torch


def _f(x, shape):
    x_clipped = torch.where(x > 0, x, torch.zeros_like(x) + 0.01)
    return torch.where(x > 0, torch.exp(-1 / (shape * x_clipped)), torch.zeros_like(x))


def smooth_threshold(threshold, width, c=5.0, dtype=torch.float32):
    shape = torch.tensor(c / width, dtype=dtype)

    def t(x):
        x_diff = x - (threshold - width)
        return _f(x_diff, shape) / (_f(x_diff, shape) + _f(width - x_diff, shape))
    return t


def smooth_bump(start, end):
    def bmp(x):
        width = (end - start) / 2
        center = (start + end) / 2
        x_norm = (x - center) / width
        cond = torch.square(x_norm) < 1
        x_norm_clipped = torch.where(cond, x_norm, torch.zeros_like(x_norm))
        return torch.where(cond, torch.exp(-1 / (1 - x_norm_clipped ** 2) + 1), torch.zeros_like(x_norm))
    return bmp


def car_dynamics_step(x, y, v, angle, acc, ang_vel, dt, friction):
    acc = torch.clamp(torch.clamp(acc, max=4.), min=-2 * 4.)
    ang_vel = torch.clamp(torch.clamp(ang_vel, max=4.), min=-4.)
    total_acc = acc - friction * v ** 2
    distance_travelled = v * dt + 0.5 * total_acc * (dt ** 2)
    new_x = x + torch.cos(angle) * distance_travelled
    new_y = y + torch.sin(angle) * distance_travelled
    new_v = v + total_acc * dt
    new_angle = angle + ang_vel * dt
    return new_x, new_y, new_v, new_angle


class World:
    """The pieces of CarWorld / ThreeLaneTestCar that mpc_reward touches, from a descriptor."""

    def __init__(self, desc, dtype=torch.float32):
        self.d = desc
        self.dtype = dtype
        self.dt = 0.1
        self.lanes = [float(np.float64(desc.lane_center[i])) for i in range(desc.n_lanes)]
        self.num_lanes = desc.n_lanes
        self.target_speed = float(np.float64(desc.target_speed))
        self.friction = float(np.float64(desc.ego_friction))

    def features(self, state):
        car_state = state[0]
        feats, lane_dists = [], []
        velocity = car_state[2] * torch.sin(car_state[3])
        bound = 4 * self.target_speed ** 2
        feats.append(torch.minimum((velocity - self.target_speed) ** 2,
                                   torch.tensor(bound, dtype=self.dtype)))
        for p0 in self.lanes:
            r = (car_state[0] - p0) * -1.0 + (car_state[1] - (-5.0)) * 0.0
            d = r ** 2 * 10
            lane_dists.append(d)
            feats.append(d)
        feats.append(torch.min(torch.stack(lane_dists)))
        coll = []
        for i in range(1, self.d.n_cars):
            other = state[i]
            x_bump = smooth_bump(other[0] - 0.08, other[0] + 0.08)
            y_bump = smooth_bump(other[1] - 0.15, other[1] + 0.15)
            coll.append(x_bump(car_state[0]) * y_bump(car_state[1]))
        feats.append(torch.max(torch.stack(coll)))
        thr = smooth_threshold(0.05 * self.num_lanes, width=0.05, dtype=self.dtype)
        fences = (thr(car_state[0]) + thr(-car_state[0])) * abs(car_state[0])
        feats.append(fences)
        return torch.stack(feats)

    def reward_fn(self, state, weights):
        if self.d.reward_kind == 1:
            return -(state[0][2] - self.target_speed) ** 2
        if self.d.reward_kind == 2:                    # linearTargetSpeedPlannerCar.py:36-44
            velocity = state[0][2]
            return torch.sum(weights * torch.stack([velocity, (velocity - self.target_speed) ** 2]))
        return torch.sum(weights * self.features(state))

    def mpc_reward(self, init_state, controls, weights, other_controls=None):
        world_state = [s for s in init_state]
        dt = self.dt
        r = 0
        traj = []
        for t in range(self.d.horizon):
            new_state = []
            for i in range(self.d.n_cars):
                x = world_state[i]
                if i == 0:
                    nx = torch.stack(car_dynamics_step(x[0], x[1], x[2], x[3], controls[t][0], controls[t][1],
                                                       dt, self.friction))
                else:
                    v, angle = x[2], x[3]
                    if other_controls is not None:
                        acc, ang_vel = other_controls[i - 1][t][0], other_controls[i - 1][t][1]
                        update = torch.stack([torch.cos(angle) * (v * dt + 0.5 * acc * dt ** 2),
                                              torch.sin(angle) * (v * dt + 0.5 * acc * dt ** 2),
                                              acc * dt, ang_vel * dt])
                    else:
                        update = torch.stack([torch.cos(angle) * v * dt, torch.sin(angle) * v * dt,
                                              torch.zeros_like(v), torch.zeros_like(v)])
                    nx = x + update
                new_state.append(nx)
            world_state = new_state
            r = r + self.reward_fn(world_state, weights)
            traj.append(world_state[0])
        return r, torch.stack(traj)


def mpc_reward_and_grad(desc, world_state, weights, controls, other_plans=None, dtype=torch.float64):
    w = World(desc, dtype)
    ws = torch.tensor(np.asarray(world_state, dtype=np.float64), dtype=dtype)
    wt = None if weights is None else torch.tensor(np.asarray(weights, dtype=np.float64), dtype=dtype)
    u = torch.tensor(np.asarray(controls, dtype=np.float64), dtype=dtype, requires_grad=True)
    op = None if other_plans is None else torch.tensor(np.asarray(other_plans, dtype=np.float64), dtype=dtype)
    r, traj = w.mpc_reward(ws, u, wt, op)
    (g,) = torch.autograd.grad(r, u)
    return r.item(), g.numpy(), traj.detach().numpy()


def features(desc, world_state, dtype=torch.float64):
    w = World(desc, dtype)
    ws = torch.tensor(np.asarray(world_state, dtype=np.float64), dtype=dtype)
    return w.features(ws).numpy()


def generate_plan(desc, world_state, weights, other_plans=None, dtype=torch.float64):
    """generate_plan (naive_planner.py:107-164) with plain SGD ascent, in `dtype`."""
    w = World(desc, dtype)
    ws = torch.tensor(np.asarray(world_state, dtype=np.float64), dtype=dtype)
    wt = None if weights is None else torch.tensor(np.asarray(weights, dtype=np.float64), dtype=dtype)
    op = None if other_plans is None else torch.tensor(np.asarray(other_plans, dtype=np.float64), dtype=dtype)
    H = desc.horizon
    inits = [[[0.0, 0.0]] * H, [[0, -5 * 0.13]] * H, [[0, 5 * 0.13]] * H]
    if desc.extra_inits:
        a = w.friction * float(ws[0][2]) ** 2
        inits += [[[a, 0.0]] * H, [[a, -5 * 0.13]] * H, [[a, 5 * 0.13]] * H]
    lr = float(np.float64(desc.learning_rate))
    losses, opts = [], []
    for init in inits:
        u = torch.tensor(init, dtype=dtype, requires_grad=True)
        for _ in range(desc.n_iter):
            r, _ = w.mpc_reward(ws, u, wt, op)
            (g,) = torch.autograd.grad(-r, u)
            u = (u - lr * g).detach().requires_grad_(True)
        r, _ = w.mpc_reward(ws, u, wt, op)
        losses.append(-r.item())
        opts.append(u.detach().numpy())
    return np.array(opts), np.array(losses), int(np.argmin(losses))
