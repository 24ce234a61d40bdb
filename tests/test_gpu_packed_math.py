"""The unpadded packed-fp32 instruction chains of the reward features (csrc/ocd_devmath.h: div2_, exp_le1_2) against
their scalar forms ON THE DEVICE, through the debug entry ocd_debug_packed_math (the evidence
used to be tools/microbench/pk_hazard.hip).  hipcc pads every dependent pair of v_pk_* instructions with an s_nop; the
kernels' inline asm leaves it out, which is only right if the hardware interlocks -- shown here on 2^17 operand pairs
per case, bit for bit, including denormal numerators / quotients, huge and tiny denominators, and against numpy's
correctly rounded float32 division."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 1 << 18                       # floats = 2^17 pairs


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def operands(rng):
    """Numerators and denominators shaped like the planner's: O(1) bump arguments, gradient numerators sliding
    through the denormal range at the edge of a collision box (exp(-1/(1-n^2))), fence denominators up to 1e30."""
    num = rng.standard_normal(N).astype(np.float32)
    den = (rng.uniform(0.05, 2.0, N) * rng.choice([-1.0, 1.0], N)).astype(np.float32)
    k = N // 8
    num[:k] *= np.float32(1e-38) * rng.uniform(1e-7, 1.0, k).astype(np.float32)          # denormal numerators
    num[k:2 * k] = (rng.uniform(1.0, 2.0, k) * 2.0 ** rng.integers(-149, -120, k)).astype(np.float32)
    den[2 * k:3 * k] = (rng.uniform(1.0, 2.0, k) * 2.0 ** rng.integers(60, 126, k)).astype(np.float32)   # quotient underflows
    den[3 * k:4 * k] = (rng.uniform(1.0, 2.0, k) * 2.0 ** rng.integers(-126, -60, k)).astype(np.float32)  # quotient overflows / huge
    num[4 * k:5 * k] = -1.0                                                                     # the kernels' -1 / u
    den[4 * k:5 * k] = rng.uniform(1e-6, 1.0, k).astype(np.float32)
    num[5 * k:5 * k + 8] = [0.0, -0.0, 1.0, np.inf, -np.inf, np.nan, 1.0, 0.0]
    den[5 * k:5 * k + 8] = [1.0, 1.0, 0.0, 1.0, np.inf, 1.0, np.nan, 0.0]
    return num, den


@pytest.mark.parametrize("seed", [1, 2])
def test_packed_division_is_the_scalar_division(hip, seed):
    from l4dc_mpc_ocd_amd.engine import default_ops
    rng = np.random.default_rng(seed)
    num, den = operands(rng)
    x = rng.uniform(-100.0, 1.0, N).astype(np.float32)
    ds, dp, es, ep = default_ops().debug_packed_math(num, den, x)
    assert np.array_equal(bits(ds), bits(dp)), f"{(bits(ds) != bits(dp)).sum()} packed quotients differ from the scalar ones"
    with np.errstate(all="ignore"):
        want = (num / den).astype(np.float32)                      # IEEE correctly rounded, denormals kept
    same = (bits(dp) == bits(want)) | (np.isnan(dp) & np.isnan(want))
    assert same.all(), f"{(~same).sum()} quotients are not the correctly rounded ones"
    den_q = np.abs(want) < np.float32(1.1754944e-38)
    assert (den_q & (want != 0)).sum() > 1000                      # denormal quotients were really exercised


@pytest.mark.parametrize("seed", [3, 4])
def test_packed_exp_is_the_scalar_exp(hip, oracle, seed):
    from l4dc_mpc_ocd_amd.engine import default_ops
    rng = np.random.default_rng(seed)
    x = np.concatenate([rng.uniform(-100.0, 1.0, N // 2), -1.0 / rng.uniform(1e-3, 50.0, N // 4) + 1.0,
                        -1.0 / rng.uniform(1e-3, 50.0, N // 4)]).astype(np.float32)
    x[:6] = [1.0, 0.0, -87.0, -87.00001, -86.99999, -1e30]
    one = np.ones(N, dtype=np.float32)
    _, _, es, ep = default_ops().debug_packed_math(one, one, x)
    assert np.array_equal(bits(es), bits(ep)), f"{(bits(es) != bits(ep)).sum()} packed exponentials differ from the scalar ones"
    ref = oracle.exp_array(x[:4096])                               # and both are the contract's exp (oracle/ocd_refmath.h)
    assert np.array_equal(bits(ep[:4096]), bits(ref))
    assert ep[2] > 0 and ep[3] == 0.0 and ep[5] == 0.0            # flush below exp(-87)


# ---- the shortened divisions (csrc/ocd_devmath.h: recip_pair_guarded, quot2_by_recip) against correctly rounded division
def pow2_times_mantissa(rng, exps):
    return (rng.uniform(1.0, 2.0, exps.size) * np.exp2(exps.astype(np.float64))).astype(np.float32)


@pytest.mark.parametrize("seed", [5, 6])
def test_guarded_reciprocal_pair_is_ieee_over_its_whole_range(hip, seed):
    """m = -1/u and k = (-m)/u without v_div_scale / v_div_fixup, from one reciprocal refinement: correctly rounded for
    EVERY u in [2^-46, 2^62] (the callers guarantee [2^-32, 2^41]): random mantissas at every exponent of that range,
    the range's ends, powers of two, values next to them."""
    from l4dc_mpc_ocd_amd.engine import default_ops
    rng = np.random.default_rng(seed)
    u = pow2_times_mantissa(rng, rng.integers(-46, 62, N))
    k = N // 16
    u[:k] = np.exp2(rng.integers(-46, 63, k).astype(np.float64)).astype(np.float32)              # exact powers of two
    u[k:2 * k] = np.nextafter(u[:k], np.float32(np.inf))
    u[2 * k:3 * k] = np.nextafter(u[:k], np.float32(0))[:k]
    u[3 * k:4 * k] = (1.0 - rng.uniform(0, 1, k) ** 2).astype(np.float32).clip(2.0 ** -24, 1.0)     # 1 - xc^2 of a collision unit
    u[4 * k:5 * k] = (100.0 * rng.uniform(7.5e-9, 0.05, k)).astype(np.float32)                   # shape * xd of a fence unit
    u[-2:] = [np.float32(2.0 ** -46), np.float32(2.0 ** 62)]
    one = np.ones(N, dtype=np.float32)
    m, kk, _ = default_ops().debug_guarded_division(u, one, one)
    want_m = (np.float32(-1.0) / u).astype(np.float32)
    want_k = ((-want_m) / u).astype(np.float32)
    assert np.array_equal(bits(m), bits(want_m)), f"{(bits(m) != bits(want_m)).sum()} of {N} reciprocals are not the IEEE quotient"
    assert np.array_equal(bits(kk), bits(want_k)), f"{(bits(kk) != bits(want_k)).sum()} of {N} second quotients are not the IEEE quotient"
    assert np.isfinite(want_k).all() and (np.abs(want_k) >= np.float32(1.1754944e-38)).all()    # the range keeps them normal


@pytest.mark.parametrize("seed", [7, 8])
def test_quotient_by_refined_reciprocal_is_ieee_inside_its_guard(hip, seed):
    """n / w through w's refined reciprocal: correctly rounded for w in [2^-20, 2^20], |n| in [2^-100, 2^75] with the
    exponent difference below 96 (either sign of n and w); outside on the large side the square stays >= 1 (or NaN),
    which is all the caller asks of it."""
    from l4dc_mpc_ocd_amd.engine import default_ops
    rng = np.random.default_rng(seed)
    ew = rng.integers(-20, 21, N)
    en = np.minimum(rng.integers(-100, 76, N), ew + 94)
    w = pow2_times_mantissa(rng, ew) * rng.choice([-1.0, 1.0], N).astype(np.float32)
    n = pow2_times_mantissa(rng, en) * rng.choice([-1.0, 1.0], N).astype(np.float32)
    k = N // 8
    w[:k] = rng.choice([0.08, 0.15], k).astype(np.float32)                                        # the scenarios' half-widths
    n[:k] = rng.uniform(-0.2, 0.2, k).astype(np.float32)
    w[k:2 * k] = w[:k]
    n[k:2 * k] = (w[:k] * (1.0 + rng.uniform(-3e-7, 3e-7, k))).astype(np.float32)                 # quotients next to +-1
    u = np.ones(N, dtype=np.float32)
    _, _, q = default_ops().debug_guarded_division(u, n, w)
    want = (n / w).astype(np.float32)
    assert np.array_equal(bits(q), bits(want)), f"{(bits(q) != bits(want)).sum()} of {N} quotients are not the IEEE quotient"
    # beyond the guard on the large side: whatever comes out, its square is not below 1
    big = pow2_times_mantissa(rng, rng.integers(76, 128, 4096)) * rng.choice([-1.0, 1.0], 4096).astype(np.float32)
    big[:3] = [np.inf, -np.inf, np.nan]
    wb = pow2_times_mantissa(rng, rng.integers(-20, 21, 4096))
    _, _, qb = default_ops().debug_guarded_division(np.ones(4096, dtype=np.float32), big, wb)
    with np.errstate(all="ignore"):
        assert not ((qb * qb) < 1.0).any()
