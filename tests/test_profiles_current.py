"""profiles/pmc_counters.json must describe THE kernels of this tree: it records the hash of the kernel sources it was
taken from (abi.kernel_source_sha: csrc/*.hip, csrc/*.h, include/ocd.h), and bench.py replays its counters and its
rocprof kernel averages only beside kernels built from those sources (a changed kernel must
not silently inherit old counters).  A kernel change therefore needs `tools/profile_round.sh` re-run on the GPU box."""
import json
import os

from l4dc_mpc_ocd_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = ("cfg2", "cfg3", "cfg4", "cfg5", "cfg4_share8", "cfg5_share8", "reference_h5", "reference_h6_extra")


def test_the_committed_profile_is_of_these_kernel_sources():
    with open(os.path.join(ROOT, "profiles", "pmc_counters.json")) as f:
        rec = json.load(f)
    assert rec["_kernel_source_sha"] == abi.kernel_source_sha(), (
        "profiles/pmc_counters.json was taken on other kernel sources: re-run tools/profile_round.sh on the GPU box")
    assert rec["_git_commit"] not in ("", "unknown")
    for k in WORKLOADS:
        r = rec[k]
        assert "ocd::mpc_" in r["kernel_name"] and r["kernel_calls"] >= 10 and r["episodes_per_launch"] > 0
        assert 0 < r["kernel_steady_avg_us"] <= r["kernel_avg_us"] * 1.01
        assert os.path.exists(os.path.join(ROOT, r["profile"])), r["profile"]
        # the text summary next to it names the same kernel
        assert r["kernel_name"] in open(os.path.join(ROOT, r["profile"])).read()


def test_kernel_names_follow_the_launch_records():
    from l4dc_mpc_ocd_amd import scenarios
    d = scenarios.local_opt(horizon=10).desc
    lat_seg = dict(scan_mode=3, chunk=0, specialised_horizon=10, build_wavefronts_per_simd=1, terminal_value=False)
    assert abi.planner_kernel_name(d, lat_seg) == "void ocd::mpc_kernel<10, 1, 3, 2, false, true>(ocd::KernelParams)"
    d5 = scenarios.merging(horizon=25).desc
    occ3 = dict(scan_mode=4, chunk=5, specialised_horizon=25, build_wavefronts_per_simd=3, terminal_value=False)
    assert abi.planner_kernel_name(d5, occ3) == "void ocd::mpc_chunk_kernel<25, 2, 3, 5, false, true>(ocd::KernelParams)"
    with open(os.path.join(ROOT, "profiles", "pmc_counters.json")) as f:
        rec = json.load(f)
    assert rec["cfg3"]["kernel_name"] == abi.planner_kernel_name(d, lat_seg)
    assert rec["cfg5"]["kernel_name"] == abi.planner_kernel_name(d5, occ3)
