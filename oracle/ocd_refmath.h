/*
 * oracle/ocd_refmath.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * fp32 exp / sin / cos restated as explicit sequences of IEEE-754 binary32
 * operations (add, mul, fma, round-to-int, integer exponent insert), so that
 * the CPU oracle and the HIP device code (l4dc-mpc-ocd_amd/csrc/ocd_devmath.h,
 * a separately written copy of the same published algorithms) produce the
 * SAME bits on any IEEE machine.
 *
 * Why not libm: the reference computes tf.exp / tf.sin / tf.cos through
 * TensorFlow 2.1.0's Eigen kernels (interact_drive/math_utils.py:30,177,
 * interact_drive/simulation_utils.py:16-17, experiments/merging.py:52).  Their
 * ulp-level behaviour is not under /root/reference (third-party, pinned in
 * requirements.txt:69-71) and differs from glibc and from AMD's ocml, so no
 * choice reproduces TensorFlow bit-for-bit.  What CAN be guaranteed is
 * oracle == kernel, and that needs one algorithm written down operation by
 * operation.  `make libm` builds the same oracle on glibc's expf/sinf/cosf to
 * measure how much that choice moves episode returns (tests/test_sensitivity).
 *
 * Algorithms (published, textbook):
 *   exp : Cody-Waite reduction x = n*ln2 + r, |r| <= ln2/2, degree-5 Cephes
 *         polynomial for (exp(r)-1-r)/r^2 (S. Moshier, Cephes expf.c), scale
 *         by 2^n through the exponent field.  Results below FLT_MIN are
 *         flushed to +0 (TensorFlow's CPU kernels run with flush-to-zero).
 *   sin/cos: Cody-Waite reduction by pi/2 with a three-term split of pi/2
 *         evaluated with fma, minimax polynomials on [-pi/4, pi/4], quadrant
 *         fix-up.  Intended domain |x| <= 1e4 rad (headings here stay within a
 *         few radians of pi/2).
 * Measured accuracy (tests/test_oracle_math.py): exp <= 1 ulp on [-87, 1],
 * sin/cos <= 1.5 ulp on [-100, 100].
 */
#ifndef OCD_REFMATH_H
#define OCD_REFMATH_H

#include <stdint.h>
#include <string.h>

#ifdef OCD_USE_LIBM
#include <math.h>
static inline float ocd_ref_expf(float x) { return expf(x); }
static inline void ocd_ref_sincosf(float x, float *s, float *c) { *s = sinf(x); *c = cosf(x); }
#else

static inline float ocd_ref_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

static inline float ocd_ref_bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t ocd_ref_f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* round-to-nearest-even for |v| < 2^22 via the 1.5*2^23 magic constant */
static inline float ocd_ref_rint_small(float v)
{
    const float magic = 12582912.0f;
    volatile float t = v + magic; /* volatile: forbid the compiler from folding (v+m)-m */
    return t - magic;
}

static inline float ocd_ref_expf(float x)
{
    if (!(x >= -87.0f)) {           /* also catches NaN -> handled below */
        if (x != x) return x;
        return 0.0f;                /* exp(-87) = 1.6e-38 is the last normal result kept */
    }
    if (x > 88.0f) return ocd_ref_bits2f(0x7f800000u);
    const float n = ocd_ref_rint_small(x * 1.44269502162933349609375f);
    float r = ocd_ref_fma(n, -0.693359375f, x);
    r = ocd_ref_fma(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = ocd_ref_fma(p, r, 1.3981999507e-3f);
    p = ocd_ref_fma(p, r, 8.3334519073e-3f);
    p = ocd_ref_fma(p, r, 4.1665795894e-2f);
    p = ocd_ref_fma(p, r, 1.6666665459e-1f);
    p = ocd_ref_fma(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float e = ocd_ref_fma(p, r2, r);
    e = e + 1.0f;
    const int32_t ni = (int32_t)n;                 /* exact: n is integral, |n| <= 128 */
    const float scale = ocd_ref_bits2f((uint32_t)(ni + 127) << 23);
    return e * scale;
}

static inline void ocd_ref_sincosf(float x, float *s_out, float *c_out)
{
    const float n = ocd_ref_rint_small(x * 0.636619746685028076171875f);
    float r = ocd_ref_fma(n, -1.57079637050628662109375f, x);
    r = ocd_ref_fma(n, 4.37113882867379306e-8f, r);
    r = ocd_ref_fma(n, 1.71512451000588188e-15f, r);
    const float z = r * r;
    /* sin(r) = r + r*z*(S1 + z*(S2 + z*(S3 + z*S4))) */
    float ps = 2.86567956e-6f;
    ps = ocd_ref_fma(ps, z, -1.98559923e-4f);
    ps = ocd_ref_fma(ps, z, 8.33338592e-3f);
    ps = ocd_ref_fma(ps, z, -1.66666672e-1f);
    const float rz = r * z;
    const float sr = ocd_ref_fma(ps, rz, r);
    /* cos(r) = 1 + z*(C1 + z*(C2 + z*(C3 + z*C4))) */
    float pc = 2.44677067e-5f;
    pc = ocd_ref_fma(pc, z, -1.38877297e-3f);
    pc = ocd_ref_fma(pc, z, 4.16666567e-2f);
    pc = ocd_ref_fma(pc, z, -5.00000000e-1f);
    const float cr = ocd_ref_fma(pc, z, 1.0f);
    const int32_t q = (int32_t)n;
    const float sv = (q & 1) ? cr : sr;
    const float cv = (q & 1) ? sr : cr;
    *s_out = (q & 2) ? -sv : sv;
    *c_out = ((q + 1) & 2) ? -cv : cv;
}
#endif /* OCD_USE_LIBM */

#endif /* OCD_REFMATH_H */
