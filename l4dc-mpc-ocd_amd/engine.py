"""Device-side plumbing around the C ABI: buffers, streams, handles.

PyTorch-ROCm is used for device allocation, host<->device copies and stream
handles only; every computation is a call into csrc/libocd_hip.so.  There is
no CPU fallback: constructing an Engine without the library or without a GPU
raises.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Dict, Optional

import numpy as np
import torch

from . import abi
from .scenarios import Scenario


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# Calls whose arguments and results are small (one world.step() of the object-by-object API: a (C, 4) world state in, H
# controls out) skip the copies altogether: the kernels read their inputs from and write their outputs to PINNED host
# memory, which HIP maps into the device's address space -- a call is then "fill the staging arrays, launch, wait for an
# event on the call's own stream, read the arrays" instead of 3-6 synchronous memcpys and a device-wide synchronisation.
_SMALL_BYTES = 1 << 16
_NP = {torch.float32: np.float32, torch.int32: np.int32}


class _Staging(threading.local):
    """Per-thread pinned staging tensors (two threads may share one Engine: tests/test_gpu_threads.py) and one event."""

    def __init__(self):
        self.bufs = {}
        self.event = None


class DeviceOps:
    """Scenario-independent device entry points (dynamics, math probes) on one GPU."""

    def __init__(self, device: Optional[str] = None):
        self.lib = abi.load_hip_library()
        if not torch.cuda.is_available():
            raise RuntimeError("no MI355X visible: the planner runs on the GPU only (no CPU fallback)")
        self.device = torch.device(device or f"cuda:{torch.cuda.current_device()}")
        self._stage = _Staging()

    # ------------------------------------------------------------------ helpers
    def _to_dev(self, a, dtype=torch.float32) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)

    def _pinned(self, tag: str, shape, dtype) -> torch.Tensor:
        key = (tag, tuple(int(v) for v in shape), dtype)
        t = self._stage.bufs.get(key)
        if t is None:
            if len(self._stage.bufs) > 256:
                self._stage.bufs.clear()
            t = self._stage.bufs[key] = torch.empty(key[1], dtype=dtype).pin_memory()
        return t

    def _in(self, tag: str, a, shape, dtype=torch.float32, small: bool = True) -> torch.Tensor:
        """An input the kernel can read: a device tensor as it is; host data through pinned staging when small (no copy
        call), else uploaded."""
        if isinstance(a, torch.Tensor) and a.is_cuda:
            return a.to(device=self.device, dtype=dtype).contiguous().reshape(shape)
        arr = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
        arr = np.ascontiguousarray(arr, dtype=_NP[dtype]).reshape(shape)
        if small and arr.nbytes <= _SMALL_BYTES:
            t = self._pinned(tag, arr.shape, dtype)
            t.numpy()[...] = arr
            return t
        return torch.as_tensor(arr).to(self.device)

    def _out(self, tag: str, shape, dtype=torch.float32, small: bool = True) -> torch.Tensor:
        n = int(np.prod(shape)) * 4
        if small and n <= _SMALL_BYTES:
            return self._pinned(tag, shape, dtype)
        return torch.empty(tuple(int(v) for v in shape), dtype=dtype, device=self.device)

    def _wait(self) -> None:
        """Block until what this thread queued on its current stream of this device is done: an event on the call's
        own stream, not a device-wide synchronisation (other streams keep running)."""
        ev = self._stage.event
        if ev is None:
            ev = self._stage.event = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        ev.synchronize()

    @staticmethod
    def _host(t: torch.Tensor) -> np.ndarray:
        """After _wait(): the result as a numpy array the caller owns."""
        return t.cpu().numpy() if t.is_cuda else t.numpy().copy()

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _call(self, fn, *args):
        with torch.cuda.device(self.device):
            abi.check(self.lib, fn(*args))

    def dynamics_batch(self, states, controls, dt: float, friction: float):
        """next_car_state for [B,4] states and [B,2] controls (simulation_utils.py:73-123)."""
        st = self._in("dyn_s", states, (-1, 4))
        if isinstance(controls, torch.Tensor):
            controls = controls.detach().cpu().numpy()
        cu = np.asarray(controls, dtype=np.float32).reshape(-1, 2)
        if cu.shape[0] == 1 and st.shape[0] > 1:
            cu = np.broadcast_to(cu, (st.shape[0], 2))
        u = self._in("dyn_u", cu, (-1, 2))
        out = self._out("dyn_o", st.shape)
        self._call(self.lib.ocd_dynamics_batch, _ptr(st), _ptr(u), float(np.float32(dt)),
                   float(np.float32(float(dt) ** 2)), float(np.float32(friction)), _ptr(out), st.shape[0],
                   self._stream())
        self._wait()
        return self._host(out)

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

    def side_by_side_streams(self, want: int, init_state, probe_episodes: int = 64) -> list:
        """Up to `want` torch streams on which launches of this handle run SIDE BY SIDE on the chip -- found by
        measurement, once per engine.  HIP maps its streams onto a few hardware queues (four by default) and two streams
        that share one run their kernels one after the other: two groups of the lockstep runs then take 2.4 ms per
        generation instead of 1.2 (profiles/r06_lockstep_streams.txt: the third and fourth stream of PyTorch's pool do).
        The probe launches `probe_episodes` episodes (the designer's weights from `init_state` [4]) on the streams
        chosen so far plus one candidate and keeps the candidate when that takes less than 1.45 x one launch alone.
        The caller has set "concurrent_launches" to `want`, so every probe launch is planned for its share of the chip."""
        import time
        have = getattr(self, "_sbs_streams", None)
        if have is None:
            have = self._sbs_streams = []
        if len(have) >= want:
            return have[:want]
        d = self.desc
        init = self._to_dev(np.asarray(init_state, dtype=np.float32).reshape(1, 4))
        w_row = np.array(d.designer_weights[:max(d.n_features, 1)], dtype=np.float32)
        P = max(1, int(probe_episodes) // max(1, d.n_samples))
        w = self._to_dev(np.tile(w_row.reshape(1, -1), (P, 1)))
        E = P * d.n_samples
        rets = {}

        def launch(st):
            ret = rets.get(st.cuda_stream)
            if ret is None:
                ret = rets[st.cuda_stream] = torch.empty((E,), dtype=torch.float32, device=self.device)
            self._call(self.lib.ocd_rollout_episodes, self._h, _ptr(init), _ptr(w), P, 1, 0, E, _ptr(ret), None, None, st.cuda_stream)

        def timed(streams):
            best = float("inf")
            for _ in range(2):
                t0 = time.perf_counter()
                for st in streams:
                    launch(st)
                for st in streams:
                    st.synchronize()
                best = min(best, time.perf_counter() - t0)
            return best

        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            if not have:
                st0 = torch.cuda.Stream(device=self.device)
                launch(st0); st0.synchronize()                     # (the first launch on a stream creates its queue)
                have.append(st0)
            alone = timed(have[:1])
            tries = 0
            while len(have) < want and tries < 3 * want + 4:
                tries += 1
                cand = torch.cuda.Stream(device=self.device)
                launch(cand); cand.synchronize()
                if timed(have + [cand]) < 1.45 * alone:
                    have.append(cand)
        self.side_by_side_probe = dict(alone_ms=alone * 1e3, tried=tries, found=len(have))
        return have[:want]

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
        sm = bool(to_numpy)                                        # results wanted as device tensors: no host staging
        ws = self._in("plan_ws", world_state, (-1, d.n_cars, 4), small=sm)
        B = ws.shape[0]
        H, K = d.horizon, d.n_ctrl_inits
        w = None
        if weights is not None:
            wdim = weights.dim() if isinstance(weights, torch.Tensor) else np.ndim(weights)
            w = self._in("plan_w", weights, (-1, d.n_features) if wdim == 2 else (-1,), small=sm)
        per = int(w is not None and w.dim() == 2)
        if w is not None and per and w.shape[0] != B:
            raise ValueError(f"weights has {w.shape[0]} rows for {B} world states")
        if isinstance(other_plans, str):
            op = self._other_plans
        else:
            op = None if other_plans is None else self._in("plan_op", other_plans, (d.n_cars - 1, H, 2), small=sm)
        plans = self._out("plan_p", (B, H, 2), small=sm)
        loss = self._out("plan_l", (B,), small=sm)
        best = self._out("plan_b", (B,), torch.int32, small=sm)
        all_plans = self._out("plan_ap", (B, K, H, 2), small=sm) if want_all else None
        all_losses = self._out("plan_al", (B, K), small=sm) if want_all else None
        vs = None
        if init_speed is not None:
            vs = self._in("plan_vs", init_speed, (-1,), small=sm)
            if vs.shape[0] != B:
                raise ValueError(f"init_speed has {vs.shape[0]} entries for {B} world states")
        self._call(self.lib.ocd_plan_batch_from, self._h, _ptr(ws), _ptr(vs), _ptr(w), per, _ptr(op), _ptr(plans),
                   _ptr(loss), _ptr(best), _ptr(all_plans), _ptr(all_losses), B, self._stream())
        out = dict(plans=plans, best_loss=loss, best_init=best)
        if want_all:
            out.update(all_plans=all_plans, all_losses=all_losses)
        if to_numpy:
            self._wait()
            out = {k: self._host(v) for k, v in out.items()}
        return out

    def rollout(self, init_states, cand_weights, ep_begin: int = 0, ep_end: Optional[int] = None,
                want_traj: bool = False, to_numpy: bool = True) -> Dict[str, object]:
        """Episodes e = (p*N + n)*S + s in [ep_begin, ep_end): MPC_ORD.eval_weights_for_init per sample.

        cand_weights: [P, D] fp32, already normalised (scenarios.planner_weights_fp32).
        """
        d = self.desc
        sm = bool(to_numpy)
        init = self._in("ro_i", init_states, (-1, 4), small=sm)
        N = init.shape[0]
        if cand_weights is None:
            w, P = None, 1
        else:
            w = self._in("ro_w", cand_weights, (-1, d.n_features), small=sm)
            P = w.shape[0]
        E = P * N * d.n_samples
        if ep_end is None:
            ep_end = E
        n = ep_end - ep_begin
        T = d.episode_len
        ret = self._out("ro_r", (max(n, 0),), small=sm)
        traj = self._out("ro_t", (n, T + 1, d.n_cars, 4), small=sm) if want_traj else None
        ctrl = self._out("ro_c", (n, T, 2), small=sm) if want_traj else None
        self._call(self.lib.ocd_rollout_episodes, self._h, _ptr(init), _ptr(w), P, N, ep_begin, ep_end,
                   _ptr(ret), _ptr(traj), _ptr(ctrl), self._stream())
        out = dict(returns=ret)
        if want_traj:
            out.update(traj=traj, ctrl=ctrl)
        if to_numpy:
            self._wait()
            out = {k: self._host(v) for k, v in out.items()}
        return out

    def rollout_indexed(self, init_states, cand_weights, episode_index, want_traj: bool = False,
                        to_numpy: bool = True, check_index: bool = True) -> Dict[str, object]:
        """Independent populations in one launch (include/ocd.h: ocd_rollout_indexed; the reference's Pool over init
        groups, run_mpc_ord.py:83-90): episode i runs candidate row episode_index[i, 0] on init row [i, 1] as reset
        number [i, 2] of its own sequential evaluation.  init_states [N_rows, 4], cand_weights [P_rows, D] (fp32,
        normalised), episode_index [E, 3] int32.  An index row that names no candidate / init row raises ValueError here
        (check_index=False leaves the check to the library: the kernel poisons that episode -- NaN return -- and
        ocd_scenario_index_error raises after the wait)."""
        d = self.desc
        init = self._to_dev(init_states).reshape(-1, 4)
        w = self._to_dev(cand_weights).reshape(-1, max(d.n_features, 1))
        idx = np.ascontiguousarray(np.asarray(episode_index, dtype=np.int32).reshape(-1, 3))
        if check_index and idx.size and (idx[:, 0].min() < 0 or idx[:, 0].max() >= w.shape[0] or idx[:, 1].min() < 0 or
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
            self._wait()
            self._call(self.lib.ocd_scenario_index_error, self._h, None)   # a device-side out-of-range row raises here
            out = {k: self._host(v) for k, v in out.items()}
        return out

    def rollout_from_state(self, world_state, weights, first_step: int, n_steps: int, sample: int = 0,
                           to_numpy: bool = True) -> Dict[str, object]:
        """n_steps CarWorld.step() calls from arbitrary world states [B, C, 4] (world step index first_step)."""
        d = self.desc
        sm = bool(to_numpy)
        ws = self._in("rfs_ws", world_state, (-1, d.n_cars, 4), small=sm)
        B = ws.shape[0]
        w = None
        if weights is not None:
            wdim = weights.dim() if isinstance(weights, torch.Tensor) else np.ndim(weights)
            w = self._in("rfs_w", weights, (-1, d.n_features) if wdim == 2 else (-1,), small=sm)
        per = int(w is not None and w.dim() == 2)
        ret = self._out("rfs_r", (B,), small=sm)
        traj = self._out("rfs_t", (B, n_steps + 1, d.n_cars, 4), small=sm)
        ctrl = self._out("rfs_c", (B, n_steps, 2), small=sm)
        self._call(self.lib.ocd_rollout_from_state, self._h, _ptr(ws), _ptr(w), per, first_step, n_steps, sample,
                   _ptr(ret), _ptr(traj), _ptr(ctrl), B, self._stream())
        out = dict(returns=ret, traj=traj, ctrl=ctrl)
        if to_numpy:
            self._wait()
            out = {k: self._host(v) for k, v in out.items()}
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
        ws = self._in("obj_ws", world_state, (-1, d.n_cars, 4))
        B = ws.shape[0]
        H = d.horizon
        u = self._to_dev(controls).reshape(-1, H, 2) if B > 1 else self._in("obj_u", controls, (-1, H, 2))
        if u.shape[0] == 1 and B > 1:
            u = u.expand(B, H, 2).contiguous()
        if u.shape[0] != B:
            raise ValueError(f"controls has {u.shape[0]} rows for {B} world states")
        w = None
        if weights is not None:
            if not isinstance(weights, torch.Tensor):
                weights = np.asarray(weights, dtype=np.float32)
            wdim = weights.dim() if isinstance(weights, torch.Tensor) else weights.ndim
            if (weights.shape[-1] if wdim else 0) != d.n_features:
                raise ValueError(f"weights has {weights.shape[-1] if wdim else 0} features, the scenario {d.n_features}")
            w = self._in("obj_w", weights, (-1, d.n_features) if wdim == 2 else (-1,))
            if wdim == 2 and w.shape[0] != B:
                raise ValueError(f"weights has {w.shape[0]} rows for {B} world states")
        per = int(w is not None and w.dim() == 2)
        if w is not None and per and w.shape[0] != B:
            raise ValueError(f"weights has {w.shape[0]} rows for {B} world states")
        if w is not None and w.shape[-1] != d.n_features:
            raise ValueError(f"weights has {w.shape[-1]} features, the scenario {d.n_features}")
        if isinstance(other_plans, str):
            op = self._other_plans
        else:
            op = None if other_plans is None else self._in("obj_op", other_plans, (d.n_cars - 1, H, 2))
        rew = self._out("obj_r", (B,))
        grad = self._out("obj_g", (B, H, 2)) if want_grad else None
        traj = self._out("obj_t", (B, H, 4)) if want_traj else None
        self._call(self.lib.ocd_mpc_reward_batch, self._h, _ptr(ws), _ptr(w), per, _ptr(u), _ptr(op), _ptr(rew),
                   _ptr(grad), _ptr(traj), B, self._stream())
        self._wait()
        out = dict(reward=self._host(rew))
        if want_grad:
            out["grad"] = self._host(grad)
        if want_traj:
            out["traj"] = self._host(traj)
        return out

    def feature_variants(self, world_state, weights):
        """(out [B, 11, 5], valid [B, 11]): every hand-written form of the reward evaluation on the same world states
        (include/ocd.h: ocd_debug_feature_variants) -- test support."""
        d = self.desc
        ws = self._to_dev(world_state).reshape(-1, d.n_cars, 4)
        B = ws.shape[0]
        w = self._to_dev(weights).reshape(-1)
        out = torch.empty((B, 11, 5), dtype=torch.float32, device=self.device)
        valid = torch.empty((B, 11), dtype=torch.int32, device=self.device)
        self._call(self.lib.ocd_debug_feature_variants, self._h, _ptr(ws), _ptr(w), _ptr(out), _ptr(valid), B, self._stream())
        self._wait()
        return out.cpu().numpy(), valid.cpu().numpy().astype(bool)

    def reward_batch(self, world_state, weights):
        d = self.desc
        ws = self._in("rw_ws", world_state, (-1, d.n_cars, 4))
        B = ws.shape[0]
        w = self._in("rw_w", weights, (-1,))
        feats = self._out("rw_f", (B, d.n_features))
        rew = self._out("rw_r", (B,))
        self._call(self.lib.ocd_reward_batch, self._h, _ptr(ws), _ptr(w), _ptr(feats), _ptr(rew), B, self._stream())
        self._wait()
        return self._host(feats), self._host(rew)
