// ocd_devmath.h -- fp32 exp / sincos for the gfx950 kernels.
//
// The planner's arithmetic contract fixes exp, sin and
// cos as explicit sequences of IEEE binary32 operations so that results do not
// depend on a vendor math library: Cody-Waite argument reduction evaluated
// with v_fma_f32, a short polynomial, and an exponent-field insert.  Compiled
// with -ffp-contract=off, so the only fused operations are the __builtin_fmaf
// calls written here.
//
//   exp : x = n*ln2 + r, |r| <= ln2/2; Cephes degree-5 polynomial for
//         (exp(r)-1-r)/r^2; 2^n through the exponent field; results below
//         FLT_MIN flush to +0 (the reference's TensorFlow CPU kernels run with
//         flush-to-zero).  <= 1 ulp on [-87, 1].
//   sincos: x = n*pi/2 + r with a three-term split of pi/2, minimax
//         polynomials on [-pi/4, pi/4], quadrant fix-up.  <= 1.5 ulp for
//         |x| <= 1e4.
//
// No hardware transcendental (v_exp_f32 / v_sin_f32) is used: their results
// are not specified bit-for-bit and would make plans irreproducible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ocd {

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// round to nearest-even integer for |v| < 2^22 (1.5 * 2^23 trick; the asm
// barrier keeps the add and the subtract from being folded)
__device__ __forceinline__ float rint_small(float v)
{
    float t = v + 12582912.0f;
    asm volatile("" : "+v"(t));
    return t - 12582912.0f;
}

__device__ __forceinline__ float exp_(float x)
{
    const float xs = (x >= -87.0f) ? x : -87.0f;           // keep the reduction finite
    const float n = rint_small(xs * 1.44269502162933349609375f);
    float r = fma_(n, -0.693359375f, xs);
    r = fma_(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fma_(p, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float e = fma_(p, r2, r);
    e = e + 1.0f;
    const int32_t ni = (int32_t)n;
    const float scale = __int_as_float((ni + 127) << 23);
    float res = e * scale;
    res = (x >= -87.0f) ? res : 0.0f;                      // flush (NaN also lands here; see below)
    res = (x > 88.0f) ? __int_as_float(0x7f800000) : res;
    res = (x != x) ? x : res;
    return res;
}

// exp_ restricted to finite x <= 1 (every exponent on the planner path is -1/u (+1) with u > 0):
// same operations, without the overflow and NaN selects that cannot trigger there.
__device__ __forceinline__ float exp_le1(float x)
{
    const float xs = (x >= -87.0f) ? x : -87.0f;
    const float n = rint_small(xs * 1.44269502162933349609375f);
    float r = fma_(n, -0.693359375f, xs);
    r = fma_(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fma_(p, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float e = fma_(p, r2, r);
    e = e + 1.0f;
    const int32_t ni = (int32_t)n;
    const float scale = __int_as_float((ni + 127) << 23);
    const float res = e * scale;
    return (x >= -87.0f) ? res : 0.0f;
}

__device__ __forceinline__ void sincos_(float x, float &s_out, float &c_out)
{
    const float n = rint_small(x * 0.636619746685028076171875f);
    float r = fma_(n, -1.57079637050628662109375f, x);
    r = fma_(n, 4.37113882867379306e-8f, r);
    r = fma_(n, 1.71512451000588188e-15f, r);
    const float z = r * r;
    float ps = 2.86567956e-6f;
    ps = fma_(ps, z, -1.98559923e-4f);
    ps = fma_(ps, z, 8.33338592e-3f);
    ps = fma_(ps, z, -1.66666672e-1f);
    const float rz = r * z;
    const float sr = fma_(ps, rz, r);
    float pc = 2.44677067e-5f;
    pc = fma_(pc, z, -1.38877297e-3f);
    pc = fma_(pc, z, 4.16666567e-2f);
    pc = fma_(pc, z, -5.00000000e-1f);
    const float cr = fma_(pc, z, 1.0f);
    const int32_t q = (int32_t)n;
    const float sv = (q & 1) ? cr : sr;
    const float cv = (q & 1) ? sr : cr;
    s_out = (q & 2) ? -sv : sv;
    c_out = ((q + 1) & 2) ? -cv : cv;
}

} // namespace ocd
