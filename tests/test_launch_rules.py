"""The launch rules of the planner (which lane mapping, chunk size, packing and build a batch gets:
launch_mpc / launch_chunk_dispatch) pinned through ocd_scenario_plan_launch -- pure host logic
of the C-ABI library, no device needed.  The expected rows are the measured winners of profiles/r03_sweep_sizes.txt."""
import ctypes as C

import pytest

from l4dc_mpc_ocd_amd import abi, scenarios


@pytest.fixture(scope="module")
def lib():
    return abi.load_hip_library()


def plan(lib, scn, n, n_cus=0, leaf=False, **opts):
    h = C.c_void_p()
    abi.check(lib, lib.ocd_scenario_create(C.byref(scn.desc), C.byref(h)))
    try:
        for k, v in opts.items():
            abi.check(lib, lib.ocd_scenario_set_option(h, k.encode(), v))
        if leaf:
            import numpy as np
            g = [np.linspace(-1, 1, 4, dtype=np.float32) for _ in range(3)]
            vals = np.zeros(64, dtype=np.float32)
            fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
            abi.check(lib, lib.ocd_scenario_set_leaf_value(h, fp(g[0]), 4, fp(g[1]), 4, fp(g[2]), 4, fp(vals), 1))
        return abi.plan_launch(lib, h, n, n_cus)
    finally:
        lib.ocd_scenario_destroy(h)


def row(ll):
    return (ll["mapping"], ll["chunk"], ll["trajectories_per_wavefront"], ll["workgroups"], ll["build_wavefronts_per_simd"])


# (scenario, horizon, trajectories) -> (mapping, chunk size, trajectories per wavefront, workgroups, build)
RULES = [
    # config 2 / config 3 and multiples (H = 10, one scripted car)
    ("finite_horizon", 10, 128, ("dpp_rows", 0, 1, 128, 1)),
    ("local_opt", 10, 2048, ("one_wavefront", 0, 2, 1024, 1)),
    ("local_opt", 10, 3072, ("chunked", 2, 3, 1024, 1)),
    ("local_opt", 10, 4096, ("chunked", 2, 4, 1024, 1)),
    ("local_opt", 10, 8192, ("chunked", 2, 4, 2048, 0)),
    ("local_opt", 10, 32768, ("chunked", 2, 4, 8192, 3)),
    # config 4's shape (H = 15, two scripted cars): per-GPU share, intermediate sizes, whole
    ("replanning", 15, 1024, ("one_wavefront", 0, 1, 1024, 1)),
    ("replanning", 15, 2048, ("chunked", 2, 2, 1024, 1)),
    ("replanning", 15, 4096, ("chunked", 3, 4, 1024, 1)),
    ("replanning", 15, 6144, ("chunked", 5, 6, 1024, 1)),
    ("replanning", 15, 8192, ("chunked", 3, 4, 2048, 0)),
    ("replanning", 15, 16384, ("chunked", 3, 4, 4096, 3)),
    # config 5's shape (H = 25): the chunked kernel at every size, the smallest chunk that fits one wavefront per SIMD
    ("merging", 25, 1, ("chunked", 2, 1, 1, 1)),
    ("merging", 25, 1024, ("chunked", 2, 1, 1024, 1)),
    ("merging", 25, 2048, ("chunked", 3, 2, 1024, 1)),
    ("merging", 25, 4096, ("chunked", 5, 4, 1024, 1)),
    ("merging", 25, 32768, ("chunked", 5, 4, 8192, 3)),
    # the reference's own horizons: DPP rows (K wavefronts per workgroup, on ONE compute unit) while a compute unit holds
    # its workgroups one wavefront per SIMD -- floor(4 / K) of them: 256 trajectories at K = 3 --, then all initialisations in
    # one wavefront (round 6: 258 ... 341 trajectories took the rows and 1.95 ms instead of 1.15, profiles/r06_two_streams.txt)
    ("finite_horizon", 5, 3, ("dpp_rows", 0, 1, 3, 1)),
    ("finite_horizon", 5, 256, ("dpp_rows", 0, 1, 256, 1)),
    ("finite_horizon", 5, 257, ("one_wavefront", 0, 1, 257, 1)),
    ("finite_horizon", 5, 324, ("one_wavefront", 0, 1, 324, 1)),
    ("finite_horizon", 5, 2048, ("one_wavefront", 0, 2, 1024, 1)),
    ("finite_horizon", 6, 256, ("dpp_rows", 0, 1, 256, 1)),
    ("finite_horizon", 6, 300, ("one_wavefront", 0, 1, 300, 1)),
    # one wavefront per trajectory beyond one per SIMD (no chunked kernel at this horizon): no longer the latency build, which
    # claims its SIMD (round 6)
    ("finite_horizon", 5, 8192, ("one_wavefront", 0, 4, 2048, 0)),
    # a horizon without a specialised kernel: run-time H, LDS windows
    ("finite_horizon", 12, 100, ("lds_windows", 0, 1, 100, 0)),
]


@pytest.mark.parametrize("name,H,n,expect", RULES, ids=[f"{r[0]}-H{r[1]}-n{r[2]}" for r in RULES])
def test_launch_rule(lib, name, H, n, expect):
    ll = plan(lib, scenarios.SCENARIOS[name](horizon=H), n)
    assert row(ll) == expect, ll
    assert ll["specialised_horizon"] == (0 if expect[0] == "lds_windows" and H == 12 else H)
    assert not ll["terminal_value"]


def test_six_control_initialisations_share_one_wavefront(lib):
    """extra_inits (K = 6, naive_planner.py:112-116): six wavefronts of a DPP-rows workgroup would crowd the four
    SIMDs of one compute unit (the reference's validation shapes ran 1.7 x slower that way, tools/small_shapes.py);
    all K initialisations go into ONE wavefront while K * H <= 64."""
    for scn in (scenarios.finite_horizon(horizon=6, extra_inits=True), scenarios.local_opt(horizon=5, extra_inits=True)):
        assert scn.desc.n_ctrl_inits == 6
        assert row(plan(lib, scn, 27)) == ("one_wavefront", 0, 1, 27, 1)
        assert row(plan(lib, scn, 1024)) == ("one_wavefront", 0, 1, 1024, 1)
    # K = 3 keeps the DPP rows (three wavefronts on three SIMDs of a compute unit)
    assert row(plan(lib, scenarios.finite_horizon(horizon=6), 27))[0] == "dpp_rows"
    # K * H > 64: no one-wavefront mapping exists, the rows stay
    assert row(plan(lib, scenarios.finite_horizon(horizon=15, extra_inits=True), 27))[0] == "dpp_rows"


def test_forced_shapes_and_fallbacks(lib):
    scn = scenarios.local_opt(horizon=10)
    assert row(plan(lib, scn, 500, scan_mode=1))[0] == "lds_windows"
    assert row(plan(lib, scn, 500, scan_mode=4, chunk_size=5)) == ("chunked", 5, 1, 500, 1)
    assert row(plan(lib, scn, 5000, scan_mode=4, chunk_size=5, no_latency_build=1))[4] == 0
    assert row(plan(lib, scn, 500, scan_mode=3, segs_per_wave=2))[:3] == ("one_wavefront", 0, 2)
    # a chunk size that is not compiled for the shape: like any mode the scenario cannot use, the LDS windows
    assert row(plan(lib, scn, 500, scan_mode=4, chunk_size=4))[0] == "lds_windows"
    # mode 2 needs H <= 16, mode 3 K * H <= 64: H = 25 falls back to the LDS windows
    scn25 = scenarios.merging(horizon=25)
    assert row(plan(lib, scn25, 64, scan_mode=2))[0] == "lds_windows"
    assert row(plan(lib, scn25, 64, scan_mode=3))[0] == "lds_windows"


def test_terminal_value_builds(lib):
    """The terminal value rides in the specialised DPP builds at the reference's horizons 5 / 6, else in the generic kernel."""
    a = plan(lib, scenarios.finite_horizon(horizon=5), 64, leaf=True)
    assert a["terminal_value"] and a["mapping"] == "dpp_rows" and a["specialised_horizon"] == 5
    b = plan(lib, scenarios.finite_horizon(horizon=5), 2048, leaf=True)
    assert b["terminal_value"] and b["mapping"] == "one_wavefront"
    c = plan(lib, scenarios.finite_horizon(horizon=10), 64, leaf=True)
    assert c["terminal_value"] and c["mapping"] == "lds_windows" and c["specialised_horizon"] == 0


def test_smaller_device(lib):
    """Rules scale with the number of SIMDs: on 64 compute units config 3's 2 048 trajectories no longer fit one
    wavefront per SIMD under any one-lane-per-step mapping."""
    scn = scenarios.local_opt(horizon=10)
    assert row(plan(lib, scn, 512, n_cus=64)) == ("one_wavefront", 0, 2, 256, 1)
    assert row(plan(lib, scn, 2048, n_cus=64))[0] == "chunked"


def test_argument_checks(lib):
    h = C.c_void_p()
    abi.check(lib, lib.ocd_scenario_create(C.byref(scenarios.local_opt(horizon=10).desc), C.byref(h)))
    info = (C.c_int32 * 8)()
    assert lib.ocd_scenario_plan_launch(h, 0, 0, info) == abi.OCD_ERR_INVALID_ARG
    assert lib.ocd_scenario_plan_launch(h, 10, -1, info) == abi.OCD_ERR_INVALID_ARG
    assert lib.ocd_scenario_plan_launch(None, 10, 0, info) == abi.OCD_ERR_INVALID_ARG
    assert lib.ocd_scenario_plan_launch(h, 10, 0, None) == abi.OCD_ERR_INVALID_ARG
    lib.ocd_scenario_destroy(h)


def test_lockstep_group_counts(lib):
    """How many launches a lockstep generation goes out as (mpc_ord._lockstep_groups): four while every quarter is a latency
    build that fits its quarter of the compute units one wavefront per SIMD, else two, else one -- planned with each group's
    share of the compute units, as "concurrent_launches" makes the launcher do (no device needed)."""
    from l4dc_mpc_ocd_amd.interact_drive.reward_design import mpc_ord

    class Planner:                                                 # Engine.plan_launch on a device-less handle
        def __init__(self, scn):
            self.h = C.c_void_p()
            abi.check(lib, lib.ocd_scenario_create(C.byref(scn.desc), C.byref(self.h)))

        def plan_launch(self, n, n_cus=0):
            return abi.plan_launch(lib, self.h, n, n_cus)

    ref = Planner(scenarios.finite_horizon(horizon=5))
    try:
        assert mpc_ord._lockstep_groups(ref, [27] * 28, cus=256) == 4      # the reference's 28 runs: 4 x 189 one-wavefront workgroups
        assert mpc_ord._lockstep_groups(ref, [27] * 12, cus=256) == 4
        assert mpc_ord._lockstep_groups(ref, [27] * 3, cus=256) == 2       # fewer runs than four groups
        assert mpc_ord._lockstep_groups(ref, [27], cus=256) == 1
        assert mpc_ord._lockstep_groups(ref, [27] * 40, cus=256) == 4      # 270 per quarter: two trajectories per wavefront (K H = 15 lanes each)
        assert mpc_ord._lockstep_groups(ref, [27] * 148, cus=256) == 4     # 999 per quarter: four per wavefront, 250 wavefronts
        assert mpc_ord._lockstep_groups(ref, [27] * 150, cus=256) == 2     # a quarter of 38 runs = 1 026 > 4 x 256; a half = 2 025 fits
        assert mpc_ord._lockstep_groups(ref, [27] * 160, cus=256) == 1     # 1 080 per quarter, 2 160 per half: past one wavefront per SIMD
        # every group's plan is a one-wavefront-per-workgroup latency build within its share
        for G, R in ((4, 28), (4, 148), (2, 150)):
            e = 27 * R // G
            p_ = ref.plan_launch(e, 256 // G)
            assert p_["build_wavefronts_per_simd"] == 1 and p_["workgroups"] * p_["wavefronts_per_workgroup"] <= 4 * (256 // G)
    finally:
        lib.ocd_scenario_destroy(ref.h)


def test_concurrent_launches_option(lib):
    """ "concurrent_launches" G is accepted for 0..16 and refused beyond (the plan itself is asked for with the share's
    compute units: ocd_scenario_plan_launch takes them as an argument)."""
    scn = scenarios.finite_horizon(horizon=5)
    h = C.c_void_p()
    abi.check(lib, lib.ocd_scenario_create(C.byref(scn.desc), C.byref(h)))
    try:
        for v in (0, 1, 2, 4, 16):
            assert lib.ocd_scenario_set_option(h, b"concurrent_launches", v) == abi.OCD_OK
        assert lib.ocd_scenario_set_option(h, b"concurrent_launches", 17) == abi.OCD_ERR_INVALID_ARG
        assert lib.ocd_scenario_set_option(h, b"concurrent_launches", -1) == abi.OCD_ERR_INVALID_ARG
        # a quarter of the chip: 189 trajectories no longer fit DPP rows (3 wavefronts per workgroup, 64 compute units)
        whole, quarter = abi.plan_launch(lib, h, 189, 256), abi.plan_launch(lib, h, 189, 64)
        assert whole["mapping"] == "dpp_rows" and quarter["mapping"] == "one_wavefront" and quarter["workgroups"] <= 256
    finally:
        lib.ocd_scenario_destroy(h)
