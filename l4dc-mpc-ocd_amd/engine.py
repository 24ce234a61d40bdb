"""Device-side plumbing around the C ABI: buffers, streams, handles.

PyTorch-ROCm is used for device allocation, host<->device copies and stream
handles only; every computation is a call into csrc/libocd_hip.so.  There is
no CPU fallback: constructing an Engine without the library or without a GPU
raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import abi
from .scenarios import Scenario


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class DeviceOps:
    """Scenario-independent device entry points (dynamics, math probes) on one GPU."""

    def __init__(self, device: Optional[str] = None):
        self.lib = abi.load_hip_library()
        if not torch.cuda.is_available():
            raise RuntimeError("no MI355X visible: the planner runs on the GPU only (no CPU fallback)")
        self.device = torch.device(device or f"cuda:{torch.cuda.current_device()}")

    # ------------------------------------------------------------------ helpers
    def _to_dev(self, a, dtype=torch.float32) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _call(self, fn, *args):
        with torch.cuda.device(self.device):
            abi.check(self.lib, fn(*args))

    def dynamics_batch(self, states, controls, dt: float, friction: float):
        """next_car_state for [B,4] states and [B,2] controls (simulation_utils.py:73-123)."""
        st = self._to_dev(states).reshape(-1, 4)
        u = self._to_dev(controls).reshape(-1, 2)
        if u.shape[0] == 1 and st.shape[0] > 1:
            u = u.expand(st.shape[0], 2).contiguous()
        out = torch.empty_like(st)
        self._call(self.lib.ocd_dynamics_batch, _ptr(st), _ptr(u), float(np.float32(dt)),
                   float(np.float32(float(dt) ** 2)), float(np.float32(friction)), _ptr(out), st.shape[0],
                   self._stream())
        torch.cuda.synchronize(self.device)
        return out.cpu().numpy()

    def debug_math(self, x):
        xin = self._to_dev(x).reshape(-1)
        n = xin.numel()
        e = torch.empty_like(xin)
        s = torch.empty_like(xin)
        c = torch.empty_like(xin)
        self._call(self.lib.ocd_debug_math, _ptr(xin), _ptr(e), _ptr(s), _ptr(c), n, self._stream())
        torch.cuda.synchronize(self.device)
        return e.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy()


    def debug_packed_math(self, num, den, x):
        """(div_scalar, div_packed, exp_scalar, exp_packed) of 2 * n_pairs operands (include/ocd.h:
        ocd_debug_packed_math): the planner's two-wide division / exp cores beside their scalar forms."""
        nn, dd, xx = (self._to_dev(a).reshape(-1) for a in (num, den, x))
        if not (nn.numel() == dd.numel() == xx.numel()) or nn.numel() % 2:
            raise ValueError("num, den and x must hold the same even number of floats")
        outs = [torch.empty_like(nn) for _ in range(4)]
        self._call(self.lib.ocd_debug_packed_math, _ptr(nn), _ptr(dd), _ptr(xx), *[_ptr(o) for o in outs],
                   nn.numel() // 2, self._stream())
        torch.cuda.synchronize(self.device)
        return tuple(o.cpu().numpy() for o in outs)


    def debug_guarded_division(self, u, n, w):
        """(m, k, q): m = -1/u and k = (-m)/u by recip_pair_guarded, q = n / w by quot2_by_recip, of 2 * n_pairs
        operands each (include/ocd.h: ocd_debug_guarded_division)."""
        uu, nn, ww = (self._to_dev(a).reshape(-1) for a in (u, n, w))
        if not (uu.numel() == nn.numel() == ww.numel()) or uu.numel() % 2:
            raise ValueError("u, n and w must hold the same even number of floats")
        outs = [torch.empty_like(uu) for _ in range(3)]
        self._call(self.lib.ocd_debug_guarded_division, _ptr(uu), _ptr(nn), _ptr(ww), *[_ptr(o) for o in outs],
                   uu.numel() // 2, self._stream())
        torch.cuda.synchronize(self.device)
        return tuple(o.cpu().numpy() for o in outs)


_default_ops: Optional[DeviceOps] = None


def default_ops() -> DeviceOps:
    global _default_ops
    if _default_ops is None:
        _default_ops = DeviceOps()
    return _default_ops


class Engine(DeviceOps):
    """One scenario handle bound to one GPU."""

    def __init__(self, scenario: Scenario, device: Optional[str] = None):
        super().__init__(device)
        self.scenario = scenario
        self.desc = scenario.desc
        h = C.c_void_p()
        abi.check(self.lib, self.lib.ocd_scenario_create(C.byref(self.desc), C.byref(h)))
        self._h = h
        plans = scenario.other_plans()
        self._other_plans = None if plans is None else self._to_dev(plans)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.ocd_scenario_destroy(h)
            self._h = None

    def set_option(self, name: str, value: int) -> None:
        """Per-handle tuning / bookkeeping option (include/ocd.h: ocd_scenario_set_option)."""
        abi.check(self.lib, self.lib.ocd_scenario_set_option(self._h, name.encode(), int(value)))

    def last_launch(self) -> dict:
        """What the most recent planner launch of this handle chose (include/ocd.h: ocd_scenario_last_launch)."""
        import ctypes as C
        info = (C.c_int32 * 8)()
        abi.check(self.lib, self.lib.ocd_scenario_last_launch(self._h, info))
        return abi.decode_launch(info)

    def plan_launch(self, n_problems: int, n_cus: int = 0) -> dict:
        """What a launch of n_problems trajectories would choose (include/ocd.h: ocd_scenario_plan_launch)."""
        return abi.plan_launch(self.lib, self._h, n_problems, n_cus)

    def set_leaf_value(self, disc_grid, values, proj_kind: int = 0) -> None:
        """Terminal value of the planner: ValueFeature(disc_grid, v_grids[t]) as leaf_evaluation
        (value_interpolation.py:28-61, naive_planner.py:69-70).  values=None removes it."""
        if values is None:
            abi.check(self.lib, self.lib.ocd_scenario_set_leaf_value(self._h, None, 0, None, 0, None, 0, None, 0))
            return
        g = [np.ascontiguousarray(np.asarray(a, dtype=np.float32)) for a in disc_grid]
        v = np.ascontiguousarray(np.asarray(values, dtype=np.float32))
        if len(g) != 3 or v.shape != tuple(len(a) for a in g):
            raise ValueError("disc_grid must hold 3 axes and values must have shape (n0, n1, n2)")
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        abi.check(self.lib, self.lib.ocd_scenario_set_leaf_value(
            self._h, fp(g[0]), len(g[0]), fp(g[1]), len(g[1]), fp(g[2]), len(g[2]), fp(v), int(proj_kind)))

    # ------------------------------------------------------------------ entry points
    def plan_batch(self, world_state, weights=None, other_plans="scenario", want_all: bool = False,
                   to_numpy: bool = True, init_speed=None) -> Dict[str, object]:
        """NaivePlanner.generate_plan for a batch of world states [B, C, 4].  init_speed [B]: the planning car's own
        current speed where it differs from the state the plan starts from (extra_inits, naive_planner.py:114)."""
        d = self.desc
        ws = self._to_dev(world_state).reshape(-1, d.n_cars, 4)
        B = ws.shape[0]
        H, K = d.horizon, d.n_ctrl_inits
        w = None if weights is None else self._to_dev(weights)
        per = int(w is not None and w.dim() == 2)
        if w is not None and per and w.shape[0] != B:
            raise ValueError(f"weights has {w.shape[0]} rows for {B} world states")
        if isinstance(other_plans, str):
            op = self._other_plans
        else:
            op = None if other_plans is None else self._to_dev(other_plans).reshape(d.n_cars - 1, H, 2)
        plans = torch.empty((B, H, 2), dtype=torch.float32, device=self.device)
        loss = torch.empty((B,), dtype=torch.float32, device=self.device)
        best = torch.empty((B,), dtype=torch.int32, device=self.device)
        all_plans = torch.empty((B, K, H, 2), dtype=torch.float32, device=self.device) if want_all else None
        all_losses = torch.empty((B, K), dtype=torch.float32, device=self.device) if want_all else None
        vs = None
        if init_speed is not None:
            vs = self._to_dev(init_speed).reshape(-1)
            if vs.shape[0] != B:
                raise ValueError(f"init_speed has {vs.shape[0]} entries for {B} world states")
        self._call(self.lib.ocd_plan_batch_from, self._h, _ptr(ws), _ptr(vs), _ptr(w), per, _ptr(op), _ptr(plans),
                   _ptr(loss), _ptr(best), _ptr(all_plans), _ptr(all_losses), B, self._stream())
        out = dict(plans=plans, best_loss=loss, best_init=best)
        if want_all:
            out.update(all_plans=all_plans, all_losses=all_losses)
        if to_numpy:
            torch.cuda.synchronize(self.device)
            out = {k: v.cpu().numpy() for k, v in out.items()}
        return out

    def rollout(self, init_states, cand_weights, ep_begin: int = 0, ep_end: Optional[int] = None,
                want_traj: bool = False, to_numpy: bool = True) -> Dict[str, object]:
        """Episodes e = (p*N + n)*S + s in [ep_begin, ep_end): MPC_ORD.eval_weights_for_init per sample.

        cand_weights: [P, D] fp32, already normalised (scenarios.planner_weights_fp32).
        """
        d = self.desc
        init = self._to_dev(init_states).reshape(-1, 4)
        N = init.shape[0]
        if cand_weights is None:
            w, P = None, 1
        else:
            w = self._to_dev(cand_weights).reshape(-1, d.n_features)
            P = w.shape[0]
        E = P * N * d.n_samples
        if ep_end is None:
            ep_end = E
        n = ep_end - ep_begin
        T = d.episode_len
        ret = torch.empty((max(n, 0),), dtype=torch.float32, device=self.device)
        traj = torch.empty((n, T + 1, d.n_cars, 4), dtype=torch.float32, device=self.device) if want_traj else None
        ctrl = torch.empty((n, T, 2), dtype=torch.float32, device=self.device) if want_traj else None
        self._call(self.lib.ocd_rollout_episodes, self._h, _ptr(init), _ptr(w), P, N, ep_begin, ep_end,
                   _ptr(ret), _ptr(traj), _ptr(ctrl), self._stream())
        out = dict(returns=ret)
        if want_traj:
            out.update(traj=traj, ctrl=ctrl)
        if to_numpy:
            torch.cuda.synchronize(self.device)
            out = {k: v.cpu().numpy() for k, v in out.items()}
        return out

    def rollout_indexed(self, init_states, cand_weights, episode_index, want_traj: bool = False,
                        to_numpy: bool = True) -> Dict[str, object]:
        """Independent populations in one launch (include/ocd.h: ocd_rollout_indexed; the reference's Pool over init
        groups, run_mpc_ord.py:83-90): episode i runs candidate row episode_index[i, 0] on init row [i, 1] as reset
        number [i, 2] of its own sequential evaluation.  init_states [N_rows, 4], cand_weights [P_rows, D] (fp32,
        normalised), episode_index [E, 3] int32."""
        d = self.desc
        init = self._to_dev(init_states).reshape(-1, 4)
        w = self._to_dev(cand_weights).reshape(-1, max(d.n_features, 1))
        idx = np.ascontiguousarray(np.asarray(episode_index, dtype=np.int32).reshape(-1, 3))
        if idx.size and (idx[:, 0].min() < 0 or idx[:, 0].max() >= w.shape[0] or idx[:, 1].min() < 0 or
                         idx[:, 1].max() >= init.shape[0] or idx[:, 2].min() < 0):
            raise ValueError("episode_index names a candidate / init row that does not exist, or a negative reset number")
        idx_dev = torch.as_tensor(idx).to(self.device)
        E, T = idx.shape[0], d.episode_len
        ret = torch.empty((E,), dtype=torch.float32, device=self.device)
        traj = torch.empty((E, T + 1, d.n_cars, 4), dtype=torch.float32, device=self.device) if want_traj else None
        ctrl = torch.empty((E, T, 2), dtype=torch.float32, device=self.device) if want_traj else None
        self._call(self.lib.ocd_rollout_indexed, self._h, _ptr(init), init.shape[0], _ptr(w), w.shape[0], _ptr(idx_dev), E,
                   _ptr(ret), _ptr(traj), _ptr(ctrl), self._stream())
        out = dict(returns=ret)
        if want_traj:
            out.update(traj=traj, ctrl=ctrl)
        if to_numpy:
            torch.cuda.synchronize(self.device)
            out = {k: v.cpu().numpy() for k, v in out.items()}
        return out

    def rollout_from_state(self, world_state, weights, first_step: int, n_steps: int, sample: int = 0,
                           to_numpy: bool = True) -> Dict[str, object]:
        """n_steps CarWorld.step() calls from arbitrary world states [B, C, 4] (world step index first_step)."""
        d = self.desc
        ws = self._to_dev(world_state).reshape(-1, d.n_cars, 4)
        B = ws.shape[0]
        w = None if weights is None else self._to_dev(weights)
        per = int(w is not None and w.dim() == 2)
        ret = torch.empty((B,), dtype=torch.float32, device=self.device)
        traj = torch.empty((B, n_steps + 1, d.n_cars, 4), dtype=torch.float32, device=self.device)
        ctrl = torch.empty((B, n_steps, 2), dtype=torch.float32, device=self.device)
        self._call(self.lib.ocd_rollout_from_state, self._h, _ptr(ws), _ptr(w), per, first_step, n_steps, sample,
                   _ptr(ret), _ptr(traj), _ptr(ctrl), B, self._stream())
        out = dict(returns=ret, traj=traj, ctrl=ctrl)
        if to_numpy:
            torch.cuda.synchronize(self.device)
            out = {k: v.cpu().numpy() for k, v in out.items()}
        return out

    def time_rollout(self, init_dev: torch.Tensor, w_dev: torch.Tensor, ep_begin: int, ep_end: int,
                     ret_dev: torch.Tensor, reps: int) -> float:
        """Mean ms per launch over `reps` launches, HIP events on the launch stream (bench.py)."""
        d = self.desc
        N = init_dev.shape[0]
        P = w_dev.shape[0]
        ms = C.c_float(0.0)
        self._call(self.lib.ocd_time_rollout, self._h, _ptr(init_dev), _ptr(w_dev), P, N, ep_begin, ep_end,
                   _ptr(ret_dev), reps, C.byref(ms), self._stream())
        return float(ms.value)

    def mpc_reward_batch(self, world_state, weights, controls, other_plans="scenario", want_grad: bool = True,
                         want_traj: bool = False) -> Dict[str, np.ndarray]:
        """NaivePlanner.reward_func and its gradient for caller-supplied controls [B, H, 2]
        (naive_planner.py:33-77)."""
        d = self.desc
        ws = self._to_dev(world_state).reshape(-1, d.n_cars, 4)
        B = ws.shape[0]
        H = d.horizon
        u = self._to_dev(controls).reshape(-1, H, 2)
        if u.shape[0] == 1 and B > 1:
            u = u.expand(B, H, 2).contiguous()
        if u.shape[0] != B:
            raise ValueError(f"controls has {u.shape[0]} rows for {B} world states")
        w = None if weights is None else self._to_dev(weights)
        per = int(w is not None and w.dim() == 2)
        if w is not None and per and w.shape[0] != B:
            raise ValueError(f"weights has {w.shape[0]} rows for {B} world states")
        if w is not None and w.shape[-1] != d.n_features:
            raise ValueError(f"weights has {w.shape[-1]} features, the scenario {d.n_features}")
        if isinstance(other_plans, str):
            op = self._other_plans
        else:
            op = None if other_plans is None else self._to_dev(other_plans).reshape(d.n_cars - 1, H, 2)
        rew = torch.empty((B,), dtype=torch.float32, device=self.device)
        grad = torch.empty((B, H, 2), dtype=torch.float32, device=self.device) if want_grad else None
        traj = torch.empty((B, H, 4), dtype=torch.float32, device=self.device) if want_traj else None
        self._call(self.lib.ocd_mpc_reward_batch, self._h, _ptr(ws), _ptr(w), per, _ptr(u), _ptr(op), _ptr(rew),
                   _ptr(grad), _ptr(traj), B, self._stream())
        torch.cuda.synchronize(self.device)
        out = dict(reward=rew.cpu().numpy())
        if want_grad:
            out["grad"] = grad.cpu().numpy()
        if want_traj:
            out["traj"] = traj.cpu().numpy()
        return out

    def reward_batch(self, world_state, weights):
        d = self.desc
        ws = self._to_dev(world_state).reshape(-1, d.n_cars, 4)
        B = ws.shape[0]
        w = self._to_dev(weights)
        feats = torch.empty((B, d.n_features), dtype=torch.float32, device=self.device)
        rew = torch.empty((B,), dtype=torch.float32, device=self.device)
        self._call(self.lib.ocd_reward_batch, self._h, _ptr(ws), _ptr(w), _ptr(feats), _ptr(rew), B, self._stream())
        torch.cuda.synchronize(self.device)
        return feats.cpu().numpy(), rew.cpu().numpy()
