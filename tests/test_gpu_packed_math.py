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
