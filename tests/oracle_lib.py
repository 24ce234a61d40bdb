"""ctypes binding of the CPU oracle (oracle/libocd_oracle.so).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

import sys
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from l4dc_mpc_ocd_amd.abi import ScenarioDesc  # noqa: E402  (struct layout only)

_I = C.POINTER(C.c_int32)


def _to_double(ctype):
    """c_float -> c_double through nested ctypes array types (for the `make fp64` build)."""
    if ctype is C.c_float:
        return C.c_double
    if hasattr(ctype, "_length_") and hasattr(ctype, "_type_"):
        return _to_double(ctype._type_) * ctype._length_
    return ctype


class ScenarioDesc64(C.Structure):
    """ocd_scenario_desc as the fp64 sensitivity build sees it (-Dfloat=double)."""
    _fields_ = [(n, _to_double(t)) for n, t in ScenarioDesc._fields_]


def _copy_desc(src, dst_cls):
    dst = dst_cls()

    def cp(a, b, name):
        v = getattr(a, name)
        if hasattr(v, "_length_"):
            tgt = getattr(b, name)

            def rec(x, y):
                for i in range(len(x)):
                    if hasattr(x[i], "_length_"):
                        rec(x[i], y[i])
                    else:
                        y[i] = x[i]
            rec(v, tgt)
        else:
            setattr(b, name, v)
    for n, _ in src._fields_:
        cp(src, dst, n)
    return dst


def _ip(a):
    return None if a is None else a.ctypes.data_as(_I)


class Oracle:
    def __init__(self, path: str, real=np.float32):
        self.lib = lib = C.CDLL(path)
        self.real = real
        creal = C.c_float if real == np.float32 else C.c_double
        self._desc_cls = ScenarioDesc if real == np.float32 else ScenarioDesc64
        _F = C.POINTER(creal)
        _D = C.POINTER(self._desc_cls)
        self._F = _F
        C_c_float = creal                      # scalar float arguments follow the build's real type
        for n in ("ocd_oracle_expf", "ocd_oracle_sinf", "ocd_oracle_cosf"):
            getattr(lib, n).restype = C_c_float
            getattr(lib, n).argtypes = [C_c_float]
        lib.ocd_oracle_f.restype = C_c_float
        lib.ocd_oracle_f.argtypes = [C_c_float, C_c_float]
        lib.ocd_oracle_smooth_threshold.restype = C_c_float
        lib.ocd_oracle_smooth_threshold.argtypes = [C_c_float] * 4
        lib.ocd_oracle_smooth_bump.restype = C_c_float
        lib.ocd_oracle_smooth_bump.argtypes = [C_c_float] * 3
        lib.ocd_oracle_dynamics_step.restype = None
        lib.ocd_oracle_dynamics_step.argtypes = [_F, _F, C_c_float, C_c_float, C_c_float, _F]
        lib.ocd_oracle_reward.restype = C_c_float
        lib.ocd_oracle_reward.argtypes = [_D, _F, _F, _F, _F]
        lib.ocd_oracle_mpc_reward.restype = C_c_float
        lib.ocd_oracle_mpc_reward.argtypes = [_D, _F, _F, _F, _F, _F, _F]
        lib.ocd_plan_batch_cpu.restype = C.c_int32
        lib.ocd_plan_batch_cpu.argtypes = [_D, _F, _F, C.c_int32, _F, _F, _F, _I, _F, _F, C.c_int64, C.c_int32]
        lib.ocd_rollout_episodes_cpu.restype = C.c_int32
        lib.ocd_rollout_episodes_cpu.argtypes = [_D, _F, _F, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                                 _F, _F, _F, C.c_int32, C.c_int32]
        lib.ocd_oracle_set_init_speed.restype = None
        lib.ocd_oracle_set_init_speed.argtypes = [_F]
        lib.ocd_oracle_set_leaf_value.restype = None
        lib.ocd_oracle_set_leaf_value.argtypes = [_F, C.c_int32, _F, C.c_int32, _F, C.c_int32, _F, C.c_int32]
        lib.ocd_rollout_from_state_cpu.restype = C.c_int32
        lib.ocd_rollout_from_state_cpu.argtypes = [_D, _F, _F, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                   _F, _F, _F, C.c_int64]
        lib.ocd_reward_batch_cpu.restype = C.c_int32
        lib.ocd_reward_batch_cpu.argtypes = [_D, _F, _F, _F, _F, C.c_int64]
        lib.ocd_oracle_uses_libm.restype = C.c_int32

    def _fp(self, a):
        return None if a is None else a.ctypes.data_as(self._F)

    def _d(self, desc):
        """byref() of the descriptor in this build's layout."""
        if self._desc_cls is ScenarioDesc:
            return C.byref(desc)
        self._last_desc = _copy_desc(desc, self._desc_cls)
        return C.byref(self._last_desc)

    # --- scalar primitives -------------------------------------------------
    def expf(self, x): return float(self.lib.ocd_oracle_expf(float(x)))
    def sinf(self, x): return float(self.lib.ocd_oracle_sinf(float(x)))
    def cosf(self, x): return float(self.lib.ocd_oracle_cosf(float(x)))
    def f(self, x, shape=5.0): return float(self.lib.ocd_oracle_f(float(x), float(shape)))

    def smooth_threshold(self, x, threshold, width, c=5.0):
        lo = np.float32(threshold - width)
        return float(self.lib.ocd_oracle_smooth_threshold(float(x), float(lo), float(width), float(np.float32(c / width))))

    def smooth_bump(self, x, start, end):
        return float(self.lib.ocd_oracle_smooth_bump(float(x), float(start), float(end)))

    def exp_array(self, x):
        x = np.ascontiguousarray(x, dtype=self.real)
        return np.array([self.lib.ocd_oracle_expf(float(v)) for v in x], dtype=self.real)

    def dynamics_step(self, state, control, dt, friction):
        st = np.ascontiguousarray(state, dtype=self.real)
        u = np.ascontiguousarray(control, dtype=self.real)
        out = np.empty(4, dtype=self.real)
        self.lib.ocd_oracle_dynamics_step(self._fp(st), self._fp(u), float(dt), float(np.float32(float(dt) ** 2)),
                                          float(friction), self._fp(out))
        return out

    # --- reward / objective ----------------------------------------------
    def reward(self, desc, world_state, weights, want_grad=False):
        ws = np.ascontiguousarray(world_state, dtype=self.real)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=self.real)
        feats = np.zeros(max(desc.n_features, 1), dtype=self.real)
        grad = np.zeros(4, dtype=self.real) if want_grad else None
        r = self.lib.ocd_oracle_reward(self._d(desc), self._fp(ws), self._fp(w), self._fp(feats), self._fp(grad))
        return self.real(r), feats[: desc.n_features], grad

    def mpc_reward(self, desc, world_state, weights, controls, other_plans=None, want_grad=True):
        ws = np.ascontiguousarray(world_state, dtype=self.real)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=self.real)
        u = np.ascontiguousarray(controls, dtype=self.real)
        op = None if other_plans is None else np.ascontiguousarray(other_plans, dtype=self.real)
        H = desc.horizon
        grad = np.zeros((H, 2), dtype=self.real) if want_grad else None
        traj = np.zeros((H, 4), dtype=self.real)
        r = self.lib.ocd_oracle_mpc_reward(self._d(desc), self._fp(ws), self._fp(w), self._fp(u), self._fp(op), self._fp(grad), self._fp(traj))
        return self.real(r), grad, traj

    # --- CPU twins of the device entry points ----------------------------
    def plan_batch(self, desc, world_state, weights, other_plans=None, n_threads=0, init_speed=None):
        ws = np.ascontiguousarray(world_state, dtype=self.real)
        C_, H, K = desc.n_cars, desc.horizon, desc.n_ctrl_inits
        ws = ws.reshape(-1, C_, 4)
        B = ws.shape[0]
        w = None if weights is None else np.ascontiguousarray(weights, dtype=self.real)
        per = int(w is not None and w.ndim == 2)
        op = None if other_plans is None else np.ascontiguousarray(other_plans, dtype=self.real)
        plans = np.zeros((B, H, 2), dtype=self.real)
        loss = np.zeros(B, dtype=self.real)
        best = np.zeros(B, dtype=np.int32)
        all_plans = np.zeros((B, K, H, 2), dtype=self.real)
        all_losses = np.zeros((B, K), dtype=self.real)
        vs = None if init_speed is None else np.ascontiguousarray(init_speed, dtype=self.real).reshape(-1)
        assert vs is None or vs.shape[0] == B
        self.lib.ocd_oracle_set_init_speed(self._fp(vs))
        try:
            st = self.lib.ocd_plan_batch_cpu(self._d(desc), self._fp(ws), self._fp(w), per, self._fp(op), self._fp(plans),
                                             self._fp(loss), _ip(best), self._fp(all_plans), self._fp(all_losses), B, n_threads)
        finally:
            self.lib.ocd_oracle_set_init_speed(None)
        if st != 0:
            raise RuntimeError(f"ocd_plan_batch_cpu -> {st}")
        return dict(plans=plans, best_loss=loss, best_init=best, all_plans=all_plans, all_losses=all_losses)

    def set_leaf_value(self, disc_grid, values, proj_kind=0):
        """Terminal value for every later call (process-global in the oracle); values=None removes it."""
        if values is None:
            self._leaf = None
            self.lib.ocd_oracle_set_leaf_value(None, 0, None, 0, None, 0, None, 0)
            return
        g = [np.ascontiguousarray(a, dtype=self.real) for a in disc_grid]
        v = np.ascontiguousarray(values, dtype=self.real)
        self._leaf = (g, v)                      # the oracle keeps the pointers
        self.lib.ocd_oracle_set_leaf_value(self._fp(g[0]), len(g[0]), self._fp(g[1]), len(g[1]),
                                           self._fp(g[2]), len(g[2]), self._fp(v), int(proj_kind))

    def rollout(self, desc, init_states, cand_weights, ep_begin=0, ep_end=None, want_traj=False, n_threads=0,
                reset_phase=0):
        init = np.ascontiguousarray(init_states, dtype=self.real).reshape(-1, 4)
        N = init.shape[0]
        if cand_weights is None:
            w, P = None, 1
        else:
            w = np.ascontiguousarray(cand_weights, dtype=self.real).reshape(-1, desc.n_features)
            P = w.shape[0]
        E = P * N * desc.n_samples
        if ep_end is None:
            ep_end = E
        n = ep_end - ep_begin
        T, C_ = desc.episode_len, desc.n_cars
        ret = np.zeros(n, dtype=self.real)
        traj = np.zeros((n, T + 1, C_, 4), dtype=self.real) if want_traj else None
        ctrl = np.zeros((n, T, 2), dtype=self.real) if want_traj else None
        st = self.lib.ocd_rollout_episodes_cpu(self._d(desc), self._fp(init), self._fp(w), P, N, ep_begin, ep_end,
                                               self._fp(ret), self._fp(traj), self._fp(ctrl), n_threads, reset_phase)
        if st != 0:
            raise RuntimeError(f"ocd_rollout_episodes_cpu -> {st}")
        return dict(returns=ret, traj=traj, ctrl=ctrl)

    def rollout_from_state(self, desc, world_state, weights, first_step, n_steps, sample=0):
        ws = np.ascontiguousarray(world_state, dtype=self.real).reshape(-1, desc.n_cars, 4)
        B = ws.shape[0]
        w = None if weights is None else np.ascontiguousarray(weights, dtype=self.real)
        per = int(w is not None and w.ndim == 2)
        ret = np.zeros(B, dtype=self.real)
        traj = np.zeros((B, n_steps + 1, desc.n_cars, 4), dtype=self.real)
        ctrl = np.zeros((B, n_steps, 2), dtype=self.real)
        st = self.lib.ocd_rollout_from_state_cpu(self._d(desc), self._fp(ws), self._fp(w), per, first_step, n_steps, sample,
                                                 self._fp(ret), self._fp(traj), self._fp(ctrl), B)
        if st != 0:
            raise RuntimeError(f"ocd_rollout_from_state_cpu -> {st}")
        return dict(returns=ret, traj=traj, ctrl=ctrl)

    def reward_batch(self, desc, world_state, weights):
        ws = np.ascontiguousarray(world_state, dtype=self.real).reshape(-1, desc.n_cars, 4)
        B = ws.shape[0]
        w = np.ascontiguousarray(weights, dtype=self.real)
        feats = np.zeros((B, desc.n_features), dtype=self.real)
        rew = np.zeros(B, dtype=self.real)
        st = self.lib.ocd_reward_batch_cpu(self._d(desc), self._fp(ws), self._fp(w), self._fp(feats), self._fp(rew), B)
        if st != 0:
            raise RuntimeError(f"ocd_reward_batch_cpu -> {st}")
        return feats, rew


_cache = {}


def build(target: str = "all") -> None:
    subprocess.run(["make", "-C", ORACLE_DIR, target], check=True, capture_output=True)


def load(variant: str = "") -> Oracle:
    """variant: '' (own exp/sin/cos, the parity oracle) or 'libm'."""
    if variant in _cache:
        return _cache[variant]
    name = "libocd_oracle.so" if not variant else f"libocd_oracle_{variant}.so"
    real = np.float64 if variant == "fp64" else np.float32
    path = os.path.join(ORACLE_DIR, name)
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("ocd_oracle.c", "ocd_oracle.h", "ocd_refmath.h")]
    srcs.append(os.path.join(ROOT, "include", "ocd.h"))
    stale = (not os.path.exists(path)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(path) for s in srcs)
    if stale:
        try:
            build("all" if not variant else variant)
        except Exception:
            if not os.path.exists(path):
                raise
    _cache[variant] = Oracle(path, real)
    return _cache[variant]
