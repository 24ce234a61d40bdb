"""The floating-point fact the work-item passes of the chunked kernel rest on (csrc/ocd_device.h, "reward, active features
as work items"): a (lane, step) pair WITHOUT an active fence / collision feature is handed (+0, +0) for
the terms it did not evaluate, where the full evaluation (merging.py:44-83 through reward_state) would have added terms that
are exactly +0 or -0.  That changes no bit because
  (1) x + (+-0) == x, bit for bit, for every float32 x except x = -0 (where x + (+0) = +0), and
  (2) a left-to-right float32 sum that STARTS from +0 (reward_state: `qx = 0.0f; qx = qx + term ...`) is never -0 under
      round-to-nearest: (+0) + (-0) = +0, exact cancellation of non-zero terms gives +0, and a non-zero partial sum stays
      non-zero or cancels to +0.
Checked here on the CPU with numpy's IEEE float32 arithmetic (no GPU, no product code)."""
import numpy as np


def _bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def test_adding_a_signed_zero_changes_no_bit_unless_the_sum_is_minus_zero():
    rng = np.random.default_rng(11)
    x = np.concatenate([
        rng.standard_normal(4096).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 4096).astype(np.float32),
        np.array([0.0, 1e-45, -1e-45, 1.17549435e-38, -1.17549435e-38, 3.4028235e38, -3.4028235e38, np.inf, -np.inf],
                 dtype=np.float32)])
    for z in (np.float32(0.0), np.float32(-0.0)):
        assert np.array_equal(_bits(x + z), _bits(x))
    assert np.isnan(np.float32(np.nan) + np.float32(-0.0))
    # the one exception, which (2) rules out for the sums in question
    m0 = np.float32(-0.0)
    assert _bits(m0 + np.float32(0.0)) == _bits(np.float32(0.0)) and _bits(m0 + m0) == _bits(m0)


def test_a_sum_that_starts_from_plus_zero_is_never_minus_zero():
    rng = np.random.default_rng(12)
    specials = np.array([0.0, -0.0, 1e-45, -1e-45, 1.0, -1.0, 1.17549435e-38, -1.17549435e-38, 3.0e38, -3.0e38],
                        dtype=np.float32)
    n_zero = 0
    for trial in range(20000):
        n = int(rng.integers(1, 7))
        terms = rng.choice(specials, n).astype(np.float32)
        if rng.random() < 0.5:                                      # pairs that cancel exactly, in random positions
            t = np.float32(rng.standard_normal())
            terms = np.concatenate([terms, [t, -t]]).astype(np.float32)
            rng.shuffle(terms)
        acc = np.float32(0.0)                                        # `float qx = 0.0f;`
        for t in terms:
            with np.errstate(over="ignore", invalid="ignore"):      # (3e38 + 3e38 = inf, inf - inf = NaN: not zeros)
                acc = np.float32(acc + t)
            if acc == 0.0:
                n_zero += 1
                assert not np.signbit(acc), (terms, acc)
    assert n_zero > 5000                                             # (the zero partial sums were exercised)
