"""The boundary from the C side: tests/c_client/abi_client.c includes include/ocd.h and include/ocd_cma.h, dlopens the
two libraries and drives them with plain pointers -- no Python, no torch in the process.  Without a GPU it must load,
validate a scenario and get OCD_ERR_NO_DEVICE from a compute call; on a GPU its returns and its native CMA-ES generation
are compared with the CPU oracle, bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from l4dc_mpc_ocd_amd import abi, scenarios, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "l4dc-mpc-ocd_amd", "csrc")
INITS = np.array([[0.01, -0.9, 0.8, np.pi / 2], [-0.03, -0.88, 0.82, np.pi / 2], [0.05, -0.92, 0.78, np.pi / 2]]).astype(np.float32)
CANDS = np.array([[-5, 0, 0, 0, -6, -50, -50],
                  [-0.21963165, -0.01184596, 0.34379187, -0.04687411, -0.06364365, -0.54138792, -0.7308079]])


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    exe = tmp_path_factory.mktemp("c_client") / "abi_client"
    subprocess.run(["gcc", "-std=c11", "-D_GNU_SOURCE", "-O1", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "tests", "c_client", "abi_client.c"),
                    "-o", str(exe), "-ldl", "-lm"], check=True)

    def run(*mode):
        return subprocess.run([str(exe), os.path.join(CSRC, "libocd_hip.so"), os.path.join(CSRC, "libocd_cma.so"), *mode],
                              capture_output=True, text=True, timeout=300)
    return run


def line(out, key):
    rows = [ln.split()[1:] for ln in out.splitlines() if ln.split()[:1] == [key]]
    assert len(rows) == 1, (key, out)
    return rows[0]


def test_c_descriptor_is_the_host_mirrors(client):
    """finite_horizon_env(horizon=5) typed into the C struct by hand gives the bytes scenarios.finite_horizon builds."""
    r = client("desc")
    assert r.returncode == 0, r.stderr
    got = bytes.fromhex(line(r.stdout, "desc")[0])
    want = bytes(scenarios.finite_horizon(horizon=5).desc)
    assert len(got) == C.sizeof(abi.ScenarioDesc)
    assert got == want


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="checks the no-GPU behaviour")
def test_c_client_without_a_gpu_gets_an_error_not_a_fallback(client):
    r = client()
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"abi {abi.OCD_ABI_VERSION} " in r.stdout
    assert f"no-device status {abi.OCD_ERR_NO_DEVICE}:" in r.stdout and "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_c_client_episodes_and_native_generation_match_the_oracle(client, oracle):
    r = client("gpu")
    assert r.returncode == 0, r.stdout + r.stderr
    scn = scenarios.finite_horizon(horizon=5)
    w32 = np.stack([scenarios.planner_weights_fp32(c) for c in CANDS])
    want = oracle.rollout(scn.desc, INITS, w32)["returns"]
    got = np.array([float(x) for x in line(r.stdout, "returns")], dtype=np.float32)
    assert np.array_equal(got, want.ravel())
    assert "indexed rollout equals the flat call on 6 episodes" in r.stdout      # ocd_rollout_indexed, reverse order, in C
    # ABI 3: an index row that names no candidate / init row is OCD_ERR_INVALID_ARG (host index: before the launch; device
    # index: NaN return + ocd_scenario_index_error), never a clamp
    assert "out-of-range index rows are errors: pinned index refused, device index NaN + reported" in r.stdout
    # the generation: the rows the C loop normalised into pinned memory, the costs it told
    gen = line(r.stdout, "generation")
    assert gen[:6] == ["done", "1", "pending", "0", "maxiter", "1"] and gen[6] == "costs"
    costs = np.array([float(x) for x in gen[7:]])
    rows = np.array([float(x) for x in line(r.stdout, "weights")], dtype=np.float32).reshape(4, 7)
    ref_ret = oracle.rollout(scn.desc, INITS, rows)["returns"]
    assert np.array_equal(costs, sharding.fitness_from_returns(ref_ret, 4, 3, 1))
    assert np.all(np.abs(np.linalg.norm(rows.astype(np.float64), axis=1) - 1) < 1e-6)
