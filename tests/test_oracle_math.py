"""Accuracy of the arithmetic contract's exp / sin / cos (oracle/ocd_refmath.h; the device code in
csrc/ocd_devmath.h is bit-identical to it, tests/test_gpu_parity.py::test_device_math_is_bitwise_the_oracles)
against float64 math.exp / sin / cos: the ulp bounds the headers state.

    exp      <= 1 ulp on [-87, 1]       (every exponent on the planner path is -1/u (+1) with u > 0)
    sin, cos <= 1.5 ulp on [-100, 100]  (headings stay within a few radians of pi/2)

ulp = spacing of binary32 at the correctly rounded result.  Results below FLT_MIN flush to +0
(TensorFlow's CPU kernels run flush-to-zero): exp(x) for x < -87.3365 is compared with 0."""
import numpy as np


def ulp_error(got, want64):
    """|got - want| in units of the binary32 spacing at the correctly rounded value of `want64`."""
    ref32 = np.float32(want64)
    spacing = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / spacing


def test_exp_within_one_ulp(oracle):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-87.0, 1.0, 200000), rng.uniform(-2.0, 1.0, 50000), np.linspace(-87.0, 1.0, 20001),
                        [0.0, -0.0, 1.0, -1.0, -87.0, np.log(2.0) / 2, -np.log(2.0) / 2]]).astype(np.float32)
    got = np.array([oracle.expf(v) for v in x], dtype=np.float32)
    want = np.exp(x.astype(np.float64))
    err = ulp_error(got, want)
    assert err.max() <= 1.0, (err.max(), x[np.argmax(err)])
    assert np.mean(err <= 0.5) > 0.85                      # mostly correctly rounded
    assert oracle.expf(0.0) == 1.0
    # flush-to-zero below FLT_MIN, no denormal results
    for v in (-87.4, -88.0, -100.0, -1e4):
        assert oracle.expf(v) == 0.0
    tiny = np.array([oracle.expf(v) for v in np.linspace(-87.3, -87.0, 301)], dtype=np.float32)
    assert np.all((tiny == 0.0) | (tiny >= np.finfo(np.float32).tiny))


def test_sin_cos_within_one_and_a_half_ulp(oracle):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-100.0, 100.0, 200000), rng.uniform(-4.0, 4.0, 100000),
                        np.float32(np.pi / 2) + rng.uniform(-1.0, 1.0, 50000),
                        [0.0, np.pi / 2, -np.pi / 2, np.pi, np.pi / 4, 3 * np.pi / 4, 1e-8, -1e-8]]).astype(np.float32)
    s = np.array([oracle.sinf(v) for v in x], dtype=np.float32)
    c = np.array([oracle.cosf(v) for v in x], dtype=np.float32)
    xs = x.astype(np.float64)
    es, ec = ulp_error(s, np.sin(xs)), ulp_error(c, np.cos(xs))
    assert es.max() <= 1.5, (es.max(), x[np.argmax(es)])
    assert ec.max() <= 1.5, (ec.max(), x[np.argmax(ec)])
    assert oracle.sinf(0.0) == 0.0 and oracle.cosf(0.0) == 1.0
