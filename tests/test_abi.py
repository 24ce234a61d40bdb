"""The C-ABI library loads on a machine without a GPU, exports every symbol include/ocd.h declares,
its struct layout matches the ctypes mirror, and compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from l4dc_mpc_ocd_amd import abi, scenarios

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ocd.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ocd_[a-z_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = abi.load_hip_library()
    names = declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ocd.h but not exported by libocd_hip.so"
    assert sorted(n for n, _, _ in abi.HIP_SYMBOLS) == names       # the Python binding covers the whole header
    assert lib.ocd_abi_version() == abi.OCD_ABI_VERSION


def test_struct_layout_matches_the_header(tmp_path):
    fields = [f for f, _ in abi.ScenarioDesc._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void){',
            'printf("%zu\\n", sizeof(ocd_scenario_desc));']
    prog += [f'printf("%zu\\n", offsetof(ocd_scenario_desc, {f}));' for f in fields]
    prog += ['return 0;}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == C.sizeof(abi.ScenarioDesc)
    assert out[1:] == [getattr(abi.ScenarioDesc, f).offset for f in fields]


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_gpu_means_error_not_fallback():
    lib = abi.load_hip_library()
    scn = scenarios.finite_horizon(horizon=5)
    h = C.c_void_p()
    assert lib.ocd_scenario_create(C.byref(scn.desc), C.byref(h)) == abi.OCD_OK     # validation is host-only
    ret = np.zeros(3, dtype=np.float32)
    st = lib.ocd_rollout_episodes(h, ret.ctypes.data, ret.ctypes.data, 1, 3, 0, 3, ret.ctypes.data, None, None, None)
    assert st == abi.OCD_ERR_NO_DEVICE and b"no CPU fallback" in lib.ocd_last_error()
    st = lib.ocd_plan_batch(h, ret.ctypes.data, ret.ctypes.data, 0, None, ret.ctypes.data, None, None, None, None, 1, None)
    assert st == abi.OCD_ERR_NO_DEVICE
    lib.ocd_scenario_destroy(h)
    from l4dc_mpc_ocd_amd.engine import Engine
    with pytest.raises(RuntimeError):
        Engine(scn)


def test_descriptor_validation_messages():
    lib = abi.load_hip_library()
    h = C.c_void_p()
    for field, value, word in [("horizon", 99, b"horizon"), ("n_samples", 0, b"n_samples"), ("abi_version", 7, b"abi_version"),
                               ("n_lanes", 0, b"lane")]:
        d = scenarios.finite_horizon(horizon=5).desc
        setattr(d, field, value)
        assert lib.ocd_scenario_create(C.byref(d), C.byref(h)) == abi.OCD_ERR_INVALID_ARG
        assert word in lib.ocd_last_error()
    assert lib.ocd_scenario_create(None, C.byref(h)) == abi.OCD_ERR_INVALID_ARG


def test_product_never_imports_the_oracle():
    """The oracle is a checker: nothing under the package or the C sources may reference oracle/."""
    pkg = os.path.join(ROOT, "l4dc-mpc-ocd_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in text and "libocd_oracle" not in text and "ocd_refmath" not in text, f


def test_argument_checks_precede_any_device_work():
    """Bad arguments are rejected with OCD_ERR_INVALID_ARG and a message, GPU or not."""
    lib = abi.load_hip_library()
    scn = scenarios.replanning(horizon=5)
    h = C.c_void_p()
    assert lib.ocd_scenario_create(C.byref(scn.desc), C.byref(h)) == abi.OCD_OK
    buf = np.zeros(64, dtype=np.float32)
    p = buf.ctypes.data
    assert lib.ocd_plan_batch(h, None, p, 0, None, p, None, None, None, None, 1, None) == abi.OCD_ERR_INVALID_ARG
    assert lib.ocd_plan_batch(h, p, None, 0, None, p, None, None, None, None, 1, None) == abi.OCD_ERR_INVALID_ARG
    assert b"weights" in lib.ocd_last_error()
    assert lib.ocd_plan_batch(h, p, p, 0, None, p, None, None, None, None, -1, None) == abi.OCD_ERR_INVALID_ARG
    assert lib.ocd_plan_batch(h, p, p, 0, None, p, None, None, None, None, 0, None) == abi.OCD_OK      # empty batch
    # episode range outside [0, P*N*S)
    assert lib.ocd_rollout_episodes(h, p, p, 2, 3, 0, 13, p, None, None, None) == abi.OCD_ERR_INVALID_ARG
    assert b"episode range" in lib.ocd_last_error()
    assert lib.ocd_rollout_episodes(h, p, p, 2, 3, 5, 5, p, None, None, None) == abi.OCD_OK               # empty range
    assert lib.ocd_rollout_from_state(h, p, p, 0, 0, 3, 9, p, None, None, 1, None) == abi.OCD_ERR_INVALID_ARG  # sample 9
    assert lib.ocd_dynamics_batch(None, p, 0.1, 0.01, 0.2, p, 4, None) == abi.OCD_ERR_INVALID_ARG
    assert lib.ocd_scenario_set_option(h, b"segs_per_wave", 99) == abi.OCD_ERR_INVALID_ARG
    assert lib.ocd_scenario_set_option(h, None, 1) == abi.OCD_ERR_INVALID_ARG
    assert lib.ocd_scenario_set_option(None, b"scan_mode", 1) == abi.OCD_ERR_INVALID_ARG
    assert lib.ocd_scenario_set_option(h, b"scan_mode", 3) == abi.OCD_OK              # per handle, no global state
    assert lib.ocd_scenario_set_option(h, b"reset_phase", 1) == abi.OCD_OK
    for knob in (b"no_latency_build", b"no_feature_skips", b"no_unified_features", b"chunk_size"):
        assert lib.ocd_scenario_set_option(h, knob, 1 if knob != b"chunk_size" else 5) == abi.OCD_OK
        assert lib.ocd_scenario_set_option(h, knob, 0) == abi.OCD_OK
    info = (C.c_int32 * 8)(*([7] * 8))
    assert lib.ocd_scenario_last_launch(h, info) == abi.OCD_OK and list(info) == [0] * 8     # nothing launched yet
    assert lib.ocd_scenario_last_launch(None, info) == abi.OCD_ERR_INVALID_ARG
    assert lib.ocd_scenario_last_launch(h, None) == abi.OCD_ERR_INVALID_ARG
    assert lib.ocd_mpc_reward_batch(h, p, p, 0, None, None, p, p, None, 1, None) == abi.OCD_ERR_INVALID_ARG  # controls NULL
    # terminal-value table: argument checks run on the host
    g = np.array([0.0, 1.0, 2.0], dtype=np.float32)
    gp = g.ctypes.data_as(C.POINTER(C.c_float))
    vals = np.zeros(27, dtype=np.float32)
    vp = vals.ctypes.data_as(C.POINTER(C.c_float))
    assert lib.ocd_scenario_set_leaf_value(h, gp, 3, gp, 3, gp, 3, vp, 7) == abi.OCD_ERR_INVALID_ARG   # proj_kind
    assert lib.ocd_scenario_set_leaf_value(h, gp, 1, gp, 3, gp, 3, vp, 0) == abi.OCD_ERR_INVALID_ARG   # one grid point
    bad = np.array([0.0, 2.0, 1.0], dtype=np.float32)
    assert lib.ocd_scenario_set_leaf_value(h, bad.ctypes.data_as(C.POINTER(C.c_float)), 3, gp, 3, gp, 3, vp, 0) \
        == abi.OCD_ERR_INVALID_ARG and b"ascending" in lib.ocd_last_error()
    assert lib.ocd_scenario_set_leaf_value(h, gp, 3, gp, 3, gp, 3, vp, 1) == abi.OCD_OK
    assert lib.ocd_scenario_set_leaf_value(h, None, 0, None, 0, None, 0, None, 0) == abi.OCD_OK        # remove
    lib.ocd_scenario_destroy(h)
    lib.ocd_scenario_destroy(None)                                                                        # no-op
