"""ctypes mirror of include/ocd.h and the loader of the HIP C-ABI library.

The product path never falls back to a CPU implementation: if the HIP library
is missing or does not export a symbol declared in include/ocd.h, loading
raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

OCD_ABI_VERSION = 3
OCD_MAX_CARS = 4
OCD_MAX_OTHERS = 3
OCD_MAX_LANES = 4
OCD_MAX_FEATURES = 8
OCD_MAX_PLAN = 8
OCD_MAX_SAMPLES = 4
OCD_MAX_HORIZON = 32
OCD_MAX_CTRL_INITS = 6

OCD_OK = 0
OCD_ERR_INVALID_ARG = -1
OCD_ERR_UNSUPPORTED = -2
OCD_ERR_HIP = -3
OCD_ERR_NO_DEVICE = -4

OCD_REWARD_LANE_FEATURES = 0
OCD_REWARD_TARGET_SPEED = 1
OCD_REWARD_LINEAR_TARGET_SPEED = 2


class ScenarioDesc(C.Structure):
    """struct ocd_scenario_desc (include/ocd.h); field order is the ABI."""

    _fields_ = [
        ("abi_version", C.c_int32),
        ("reward_kind", C.c_int32),
        ("n_cars", C.c_int32),
        ("n_lanes", C.c_int32),
        ("horizon", C.c_int32),
        ("n_iter", C.c_int32),
        ("extra_inits", C.c_int32),
        ("check_plans", C.c_int32),
        ("episode_len", C.c_int32),
        ("n_samples", C.c_int32),
        ("teleport_step", C.c_int32),
        ("teleport_car", C.c_int32 * OCD_MAX_SAMPLES),
        ("teleport_state", C.c_float * 4),
        ("dt", C.c_float),
        ("dt_sq", C.c_float),
        ("learning_rate", C.c_float),
        ("ego_friction", C.c_float),
        ("target_speed", C.c_float),
        ("lane_center", C.c_float * OCD_MAX_LANES),
        ("fence_lo", C.c_float),
        ("fence_width", C.c_float),
        ("fence_shape", C.c_float),
        ("bump_half_x", C.c_float),
        ("bump_half_y", C.c_float),
        ("other_init", (C.c_float * 4) * OCD_MAX_OTHERS),
        ("other_friction", C.c_float * OCD_MAX_OTHERS),
        ("other_plan_len", C.c_int32 * OCD_MAX_OTHERS),
        ("other_plan", ((C.c_float * 2) * OCD_MAX_PLAN) * OCD_MAX_OTHERS),
        ("other_default", (C.c_float * 2) * OCD_MAX_OTHERS),
        ("designer_weights", C.c_float * OCD_MAX_FEATURES),
        ("teleport_period", C.c_int32),
        ("other_assumed_default", (C.c_float * 2) * OCD_MAX_OTHERS),
        ("lane_origin_y", C.c_float),
        ("lane_normal_y", C.c_float),
    ]

    @property
    def n_features(self) -> int:
        if self.reward_kind == OCD_REWARD_LANE_FEATURES:
            return self.n_lanes + 4
        return 2 if self.reward_kind == OCD_REWARD_LINEAR_TARGET_SPEED else 0

    @property
    def n_ctrl_inits(self) -> int:
        return 6 if self.extra_inits else 3


# Every symbol include/ocd.h declares: (name, restype, argtypes)
_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int32)
_VP = C.c_void_p
HIP_SYMBOLS = [
    ("ocd_abi_version", C.c_int32, []),
    ("ocd_device_count", C.c_int32, []),
    ("ocd_last_error", C.c_char_p, []),
    ("ocd_scenario_create", C.c_int32, [C.POINTER(ScenarioDesc), C.POINTER(_VP)]),
    ("ocd_scenario_destroy", None, [_VP]),
    ("ocd_scenario_set_option", C.c_int32, [_VP, C.c_char_p, C.c_int32]),
    ("ocd_scenario_last_launch", C.c_int32, [_VP, _I]),
    ("ocd_scenario_plan_launch", C.c_int32, [_VP, C.c_int64, C.c_int32, _I]),
    ("ocd_scenario_set_leaf_value", C.c_int32, [_VP, _F, C.c_int32, _F, C.c_int32, _F, C.c_int32, _F, C.c_int32]),
    ("ocd_plan_batch", C.c_int32,
     [_VP, _VP, _VP, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    ("ocd_plan_batch_from", C.c_int32,
     [_VP, _VP, _VP, _VP, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    ("ocd_rollout_episodes", C.c_int32,
     [_VP, _VP, _VP, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _VP, _VP, _VP, _VP]),
    ("ocd_rollout_indexed", C.c_int32,
     [_VP, _VP, C.c_int64, _VP, C.c_int64, _VP, C.c_int64, _VP, _VP, _VP, _VP]),
    ("ocd_scenario_index_error", C.c_int32, [_VP, C.POINTER(C.c_int64)]),
    ("ocd_rollout_from_state", C.c_int32,
     [_VP, _VP, _VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _VP, _VP, _VP, C.c_int64, _VP]),
    ("ocd_mpc_reward_batch", C.c_int32,
     [_VP, _VP, _VP, C.c_int32, _VP, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    ("ocd_dynamics_batch", C.c_int32, [_VP, _VP, C.c_float, C.c_float, C.c_float, _VP, C.c_int64, _VP]),
    ("ocd_reward_batch", C.c_int32, [_VP, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    ("ocd_stream_synchronize", C.c_int32, [_VP]),
    ("ocd_debug_math", C.c_int32, [_VP, _VP, _VP, _VP, C.c_int64, _VP]),
    ("ocd_debug_packed_math", C.c_int32, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    ("ocd_debug_guarded_division", C.c_int32, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    ("ocd_debug_feature_variants", C.c_int32, [_VP, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    ("ocd_time_rollout", C.c_int32,
     [_VP, _VP, _VP, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _VP, C.c_int32,
      C.POINTER(C.c_float), _VP]),
]

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_PKG_DIR, "csrc", "libocd_hip.so")

_lib: Optional[C.CDLL] = None


class OcdError(RuntimeError):
    """A C-ABI call returned a negative ocd_status."""

    def __init__(self, status: int, message: str):
        super().__init__(f"ocd status {status}: {message}")
        self.status = status


def load_hip_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libocd_hip.so and bind every declared symbol.  Raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("OCD_HIP_LIB") or None      # experiments: an alternative build of the library
    p = path or HIP_LIB_PATH
    if not os.path.exists(p) and path is None:
        # a source-only checkout: compile the library (hipcc cross-compiles gfx950 without a GPU).
        # This builds the product; it is not a fallback -- if it fails, loading fails.
        import subprocess
        r = subprocess.run(["make", "-C", os.path.dirname(p), "all"], capture_output=True, text=True)
        if r.returncode != 0:
            raise FileNotFoundError(f"{p} is missing and `make -C {os.path.dirname(p)}` failed:\n{r.stderr[-2000:]}")
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the MI355X planner has no CPU fallback)")
    # PyTorch-ROCm ships its own libamdhip64; import it first so that this library binds to the
    # SAME HIP runtime (same SONAME, first one loaded wins) and torch's device pointers and
    # streams are valid in it.  Two runtimes in one process do not share devices or memory.
    import torch  # noqa: F401
    lib = C.CDLL(p)
    for name, restype, argtypes in HIP_SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.ocd_abi_version()
    if got != OCD_ABI_VERSION:
        raise RuntimeError(f"libocd_hip.so ABI {got} != header ABI {OCD_ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def kernel_source_sha() -> str:
    """sha256 (first 16 hex digits) over the sources the device code is built from -- csrc/*.hip, csrc/*.h and
    include/ocd.h, in name order: the identity of the kernels.  Profiles record it (tools/rocprof_summary.py) and
    bench.py replays a profile's counters only beside kernels built from the same sources."""
    import glob
    import hashlib
    csrc = os.path.join(_PKG_DIR, "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")))
    files.append(os.path.join(os.path.dirname(_PKG_DIR), "include", "ocd.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def planner_kernel_name(desc, launch: dict) -> str:
    """The demangled name of the planner kernel a launch record (decode_launch) stands for, as rocprofv3 prints it:
    ocd::mpc_kernel<HT, NO, L, V, LEAF, LAT> / ocd::mpc_chunk_kernel<HT, NO, L, S, LAT, OCC3> (csrc/ocd_kernels.hip,
    csrc/ocd_chunk_kernel.hip)."""
    b = lambda v: "true" if v else "false"  # noqa: E731
    no = desc.n_cars - 1
    lanes = desc.n_lanes if desc.reward_kind == OCD_REWARD_LANE_FEATURES else (0 if desc.reward_kind == OCD_REWARD_TARGET_SPEED else -1)
    ht, build = launch["specialised_horizon"], launch["build_wavefronts_per_simd"]
    if launch["scan_mode"] == 4:
        return f"void ocd::mpc_chunk_kernel<{ht}, {no}, {lanes}, {launch['chunk']}, {b(build == 1)}, {b(build >= 3)}>(ocd::KernelParams)"
    v = {1: 0, 2: 1, 3: 2}.get(launch["scan_mode"], 0)
    return f"void ocd::mpc_kernel<{ht}, {no}, {lanes}, {v}, {b(launch['terminal_value'])}, {b(build == 1)}>(ocd::KernelParams)"


def check(lib: C.CDLL, status: int) -> None:
    if status != OCD_OK:
        msg = lib.ocd_last_error()
        raise OcdError(status, msg.decode() if msg else "")


LAUNCH_MAPPINGS = {0: "none", 1: "lds_windows", 2: "dpp_rows", 3: "one_wavefront", 4: "chunked"}


def decode_launch(info) -> dict:
    """The int32[8] record of ocd_scenario_last_launch / ocd_scenario_plan_launch (include/ocd.h) as a dict."""
    return {"scan_mode": info[0], "mapping": LAUNCH_MAPPINGS.get(info[0], "?"), "chunk": info[1],
            "trajectories_per_wavefront": info[2], "workgroups": info[3], "build_wavefronts_per_simd": info[4],
            "specialised_horizon": info[5], "terminal_value": bool(info[6]), "wavefronts_per_workgroup": info[7]}


def plan_launch(lib, handle, n_problems: int, n_cus: int = 0) -> dict:
    """What a launch of n_problems trajectories would choose on a device of n_cus compute units (0 = 256); no device needed."""
    info = (C.c_int32 * 8)()
    check(lib, lib.ocd_scenario_plan_launch(handle, int(n_problems), int(n_cus), info))
    return decode_launch(info)
