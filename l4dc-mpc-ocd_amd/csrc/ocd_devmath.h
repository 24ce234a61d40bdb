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

// round to nearest-even integer for |v| < 2^22 (1.5 * 2^23 trick; the asm barrier keeps the add and the subtract
// from being folded)
__device__ __forceinline__ float rint_small(float v)
{
    float t = v + 12582912.0f;
    asm volatile("" : "+v"(t));
    return t - 12582912.0f;
}

// The same two additions without the barrier: hipcc may not reassociate (v + M) - M without fast-math flags, so the
// barrier is redundant -- and it costs an issue slot per call, because hipcc pads every register an asm statement
// "defines" with a hazard s_nop before the next vector instruction reads it.  Used by the latency builds' sincos
// (round 4).  Everything else keeps rint_small: in the register-bound throughput builds of the chunked kernel the
// barrier-free form moves live ranges and costs 12-32 more bytes of scratch per lane (config 5 whole +2.6 %, measured).
__device__ __forceinline__ float rint_small_nb(float v)
{
    const float t = v + 12582912.0f;
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

// ---- two-wide forms: the SAME operations on two independent values per lane through v_pk_* instructions.
// A lone wavefront issues a packed fp32 instruction at the cost of a scalar one (tools/microbench/
// valu_latency.hip).  The packed runs are inline asm: hipcc pads every dependent pair of packed
// instructions with an s_nop that the hardware does not need (tools/microbench/pk_hazard.hip: the
// unpadded chain is bit-identical to the scalar one on every lane), which would eat the gain.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f splat2(float v) { return v2f{v, v}; }

// IEEE-correct fp32 division of two independent quotients: hipcc's own expansion (v_div_scale, v_rcp, the
// Newton-Raphson refinement, v_div_fmas, v_div_fixup) with the six refinement operations packed.  The
// result of a correctly rounded division does not depend on the instruction sequence.
__device__ __forceinline__ v2f div2_(v2f n, v2f d)
{
    bool f0, f1, g0, g1;
    v2f ds, ns, r, r1, e, q;
    ds.x = __builtin_amdgcn_div_scalef(n.x, d.x, false, &g0);
    ds.y = __builtin_amdgcn_div_scalef(n.y, d.y, false, &g1);
    ns.x = __builtin_amdgcn_div_scalef(n.x, d.x, true, &f0);
    ns.y = __builtin_amdgcn_div_scalef(n.y, d.y, true, &f1);
    r.x = __builtin_amdgcn_rcpf(ds.x);
    r.y = __builtin_amdgcn_rcpf(ds.y);
    // s_nop 0: a v_rcp_f32 result needs one wait state before a non-transcendental instruction reads it
    asm("s_nop 0\n"
        "v_pk_fma_f32 %[e], %[ds], %[r], 1.0 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"   // 1 - ds*r
        "v_pk_fma_f32 %[r1], %[e], %[r], %[r]\n"                                                 // refined reciprocal
        "v_pk_mul_f32 %[q], %[ns], %[r1]\n"
        "v_pk_fma_f32 %[e], %[ds], %[q], %[ns] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
        "v_pk_fma_f32 %[q], %[e], %[r1], %[q]\n"
        "v_pk_fma_f32 %[e], %[ds], %[q], %[ns] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
        // (r1 is its own output: a tied in/out operand costs a v_mov_b64 of the two v_rcp results)
        : [e] "=&v"(e), [r1] "=&v"(r1), [q] "=&v"(q) : [ds] "v"(ds), [ns] "v"(ns), [r] "v"(r));
    v2f out;
    out.x = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e.x, r1.x, q.x, f0), d.x, n.x);
    out.y = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e.y, r1.y, q.y, f1), d.y, n.y);
    return out;
}

// m = -1/d and k = (-m)/d (= div2_(splat2(-1), d) and div2_(-m, d)) for denominators the CALLER has shown to lie well
// inside [2^-46, 2^62] (the callers' bounds: [2^-32, 2^41]): there v_div_scale_f32 returns both operands of both divisions
// unchanged with VCC = 0 -- the numerators are -1 and 1/d: neither exponent difference reaches 96 or -126, nothing is
// denormal, no numerator is below 2^-103 --, v_div_fmas_f32 is the plain fma and v_div_fixup_f32 passes the quotient
// through.  So the two divisions are the operations below, the SAME operations on the same values as div2_'s, with the
// scaling / fix-up instructions (identities) left out and ONE reciprocal refinement serving both (same denominator ->
// same v_rcp_f32, same two fma).  17 issue slots instead of 38.
__device__ __forceinline__ void recip_pair_guarded(v2f d, v2f &m, v2f &k)
{
    v2f r, r1, e, q;
    r.x = __builtin_amdgcn_rcpf(d.x);
    r.y = __builtin_amdgcn_rcpf(d.y);
    asm("s_nop 0\n"
        "v_pk_fma_f32 %[e], %[d], %[r], 1.0 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"     // 1 - d*r
        "v_pk_fma_f32 %[r1], %[e], %[r], %[r]\n"                                                   // refined reciprocal
        "v_pk_mul_f32 %[q], %[r1], -1.0 op_sel_hi:[1,0]\n"                                         // n*r1, n = -1
        "v_pk_fma_f32 %[e], %[d], %[q], -1.0 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"    // n - d*q
        "v_pk_fma_f32 %[q], %[e], %[r1], %[q]\n"
        "v_pk_fma_f32 %[e], %[d], %[q], -1.0 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
        "v_pk_fma_f32 %[m], %[e], %[r1], %[q]\n"                                                   // (v_div_fmas_f32, VCC = 0)
        "v_pk_mul_f32 %[q], %[m], %[r1] neg_lo:[1,0] neg_hi:[1,0]\n"                               // n*r1, n = -m
        "v_pk_fma_f32 %[e], %[d], %[q], %[m] neg_lo:[1,0,1] neg_hi:[1,0,1]\n"                      // n - d*q
        "v_pk_fma_f32 %[q], %[e], %[r1], %[q]\n"
        "v_pk_fma_f32 %[e], %[d], %[q], %[m] neg_lo:[1,0,1] neg_hi:[1,0,1]\n"
        "v_pk_fma_f32 %[k], %[e], %[r1], %[q]\n"
        : [e] "=&v"(e), [r1] "=&v"(r1), [q] "=&v"(q), [m] "=&v"(m), [k] "=&v"(k) : [d] "v"(d), [r] "v"(r));
}

// The refined reciprocal div2_ computes for a denominator it does not scale (v_rcp_f32 and one Newton step): for
// denominators that stay the same over many divisions (the bump half-widths of a control step), computed once.
__device__ __forceinline__ float refined_recip(float d)
{
    const float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

// n / d for two quotients, d in [2^-20, 2^20] with r1 = refined_recip(d), |n| in [2^-100, 2^76): there v_div_scale_f32
// changes neither operand, v_div_fmas_f32 is the plain fma and v_div_fixup_f32 passes the quotient through, so these five
// operations are div2_'s own (same operations, same values).  Beyond that range on the large side (or for a NaN / an
// infinite n) the result is some value of magnitude >= 2^56, an infinity or a NaN -- enough for the caller here, who
// only asks whether the square is below 1; zero and tiny numerators are the caller's to exclude.
__device__ __forceinline__ v2f quot2_by_recip(v2f n, v2f d, v2f r1)
{
    v2f e, q, out;
    asm("v_pk_mul_f32 %[q], %[n], %[r1]\n"
        "v_pk_fma_f32 %[e], %[d], %[q], %[n] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
        "v_pk_fma_f32 %[q], %[e], %[r1], %[q]\n"
        "v_pk_fma_f32 %[e], %[d], %[q], %[n] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
        "v_pk_fma_f32 %[out], %[e], %[r1], %[q]\n"
        : [e] "=&v"(e), [q] "=&v"(q), [out] "=&v"(out) : [n] "v"(n), [d] "v"(d), [r1] "v"(r1));
    return out;
}

// Constants of exp_le1_2, two per VGPR pair (a packed instruction picks the low or the high dword of a
// source for both of its lanes through op_sel / op_sel_hi).  Built once per kernel and pinned in registers.
struct PkConsts { v2f a, b, c, d, e; };

__device__ __forceinline__ PkConsts pk_consts()
{
    PkConsts k;
    k.a = v2f{1.44269502162933349609375f, 12582912.0f};      // log2(e)        | 1.5 * 2^23
    k.b = v2f{-12582912.0f, -0.693359375f};                   // -1.5 * 2^23    | -ln2 (high part)
    k.c = v2f{2.12194440e-4f, 1.9875691500e-4f};              // ln2 (low part) | c5
    k.d = v2f{1.3981999507e-3f, 8.3334519073e-3f};            // c4 | c3
    k.e = v2f{4.1665795894e-2f, 1.6666665459e-1f};            // c2 | c1
    // (c0 = 5.0000001201e-1f rounds to exactly 0.5f: an inline constant of the instruction, no register)
    static_assert(5.0000001201e-1f == 0.5f, "c0 is the inline constant 0.5");
    asm volatile("" : "+v"(k.a), "+v"(k.b), "+v"(k.c), "+v"(k.d), "+v"(k.e));   // keep, do not rematerialise
    return k;
}

// exp_le1 of two values: the reduction and the polynomial packed (same operations as exp_le1, element-wise)
__device__ __forceinline__ v2f exp_le1_2(v2f x, const PkConsts &k)
{
    v2f xs, n, r, p, e;
    xs.x = (x.x >= -87.0f) ? x.x : -87.0f;
    xs.y = (x.y >= -87.0f) ? x.y : -87.0f;
    asm("v_pk_mul_f32 %[n], %[xs], %[A] op_sel_hi:[1,0]\n"                        // xs * log2(e)
        "v_pk_add_f32 %[n], %[n], %[A] op_sel:[0,1] op_sel_hi:[1,1]\n"            // + 1.5*2^23  } round to nearest-even
        "v_pk_add_f32 %[n], %[n], %[B] op_sel_hi:[1,0]\n"                         // - 1.5*2^23  } integer
        "v_pk_fma_f32 %[r], %[n], %[B], %[xs] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n" // n * -ln2_hi + xs
        "v_pk_fma_f32 %[r], %[n], %[C], %[r] op_sel_hi:[1,0,1]\n"                 // n * ln2_lo + r
        "v_pk_fma_f32 %[p], %[C], %[r], %[D] op_sel:[1,0,0] op_sel_hi:[1,1,0]\n"   // c5 * r + c4
        "v_pk_fma_f32 %[p], %[p], %[r], %[D] op_sel:[0,0,1] op_sel_hi:[1,1,1]\n"   // p * r + c3
        "v_pk_fma_f32 %[p], %[p], %[r], %[E] op_sel_hi:[1,1,0]\n"                 // p * r + c2
        "v_pk_fma_f32 %[p], %[p], %[r], %[E] op_sel:[0,0,1] op_sel_hi:[1,1,1]\n"   // p * r + c1
        "v_pk_fma_f32 %[p], %[p], %[r], 0.5 op_sel_hi:[1,1,0]\n"                  // p * r + c0 (= 0.5 exactly)
        "v_pk_mul_f32 %[e], %[r], %[r]\n"
        "v_pk_fma_f32 %[e], %[p], %[e], %[r]\n"
        "v_pk_add_f32 %[e], %[e], 1.0 op_sel_hi:[1,0]\n"
        : [n] "=&v"(n), [r] "=&v"(r), [p] "=&v"(p), [e] "=&v"(e)
        : [xs] "v"(xs), [A] "v"(k.a), [B] "v"(k.b), [C] "v"(k.c), [D] "v"(k.d), [E] "v"(k.e));
    v2f scale;
    scale.x = __int_as_float(((int32_t)n.x + 127) << 23);
    scale.y = __int_as_float(((int32_t)n.y + 127) << 23);
    v2f out;
    out.x = (x.x >= -87.0f) ? (e.x * scale.x) : 0.0f;
    out.y = (x.y >= -87.0f) ? (e.y * scale.y) : 0.0f;
    return out;
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
    // quadrant signs as a sign-bit xor (exact negation): no compare, so no mask hazard slots
    s_out = __uint_as_float(__float_as_uint(sv) ^ (((uint32_t)q << 30) & 0x80000000u));
    c_out = __uint_as_float(__float_as_uint(cv) ^ (((uint32_t)(q + 1) << 30) & 0x80000000u));
}

// sincos_ with the two Horner chains (sin: ((s3 z + s2) z + s1) z + s0, cos: ((c3 z + c2) z + c1) z + c0) run two-wide
// and the quadrant swap done with bit selects instead of a compare and two v_cndmask (no mask hazard slot): the SAME
// operations on every element, 5 instructions fewer (round 4).  The coefficient pairs live in registers.
struct ScConsts { v2f k3, k2, k1, k0; };

__device__ __forceinline__ ScConsts sc_consts()
{
    ScConsts k;
    k.k3 = v2f{2.86567956e-6f, 2.44677067e-5f};
    k.k2 = v2f{-1.98559923e-4f, -1.38877297e-3f};
    k.k1 = v2f{8.33338592e-3f, 4.16666567e-2f};
    k.k0 = v2f{-1.66666672e-1f, -5.00000000e-1f};
    asm volatile("" : "+v"(k.k3), "+v"(k.k2), "+v"(k.k1), "+v"(k.k0));      // keep, do not rematerialise
    return k;
}

#define OCD_SC_WAVE_SHR " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// The reduction and the two Horner chains of sincos_pk: n (the quadrant count as a float), r, Z = (r^2, r^3), P = (sin
// chain, cos chain); the callers finish with sr = fma(P.x, Z.y, r), cr = fma(P.y, Z.x, 1) and the quadrant fix-up.
// PK = false: the two chains as scalar v_fma (round 4, measured: on a launch that fills every SIMD -- config 3 -- the
// chip holds a lower clock under the packed chains and the launch takes 0.8 % LONGER although each wavefront needs
// 2 % fewer cycles; launches that leave SIMDs idle -- V_ROW, small batches -- and the chunked latency builds gain the
// full 1.1-1.4 % from the packed form: tools/ab3.sh, profiles/r04_sincos_ab.txt).
template <bool PK = true>
__device__ __forceinline__ void sincos_pk_head(float x, const ScConsts &k, float &n, float &r, v2f &Z, v2f &P)
{
    n = rint_small_nb(x * 0.636619746685028076171875f);
    r = fma_(n, -1.57079637050628662109375f, x);
    r = fma_(n, 4.37113882867379306e-8f, r);
    r = fma_(n, 1.71512451000588188e-15f, r);
    Z.x = r * r;                                // both chains multiply by Z.x (op_sel_hi picks the low half)
    Z.y = r * Z.x;
    if constexpr (!PK) {
    float ps = 2.86567956e-6f, pc = 2.44677067e-5f;
    ps = fma_(ps, Z.x, -1.98559923e-4f); pc = fma_(pc, Z.x, -1.38877297e-3f);
    ps = fma_(ps, Z.x, 8.33338592e-3f);  pc = fma_(pc, Z.x, 4.16666567e-2f);
    ps = fma_(ps, Z.x, -1.66666672e-1f); pc = fma_(pc, Z.x, -5.00000000e-1f);
    P.x = ps; P.y = pc;
    } else
    asm("v_pk_fma_f32 %[p], %[k3], %[z], %[k2] op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %[p], %[p], %[z], %[k1] op_sel_hi:[1,0,1]\n"
        "v_pk_fma_f32 %[p], %[p], %[z], %[k0] op_sel_hi:[1,0,1]\n"
        : [p] "=&v"(P) : [z] "v"(Z), [k3] "v"(k.k3), [k2] "v"(k.k2), [k1] "v"(k.k1), [k0] "v"(k.k0));
}

// quadrant: q = (int)n; odd quadrants swap sin and cos (bit select on m = -(q & 1)); the sign bits are bit 1 of q (sin)
// and of q + 1 (cos), moved to bit 31: out = value ^ ((q << 30) & 0x80000000)   [bitop3 0x6c: B ^ (A & C)].
// The three integer instructions come first: they cover the packed chain's result latency.
#define OCD_SC_TAIL                                                                                                   \
        "v_cvt_i32_f32 %[q], %[n]\n"                                                                                  \
        "v_bfe_i32 %[m], %[q], 0, 1\n"                                                                                \
        "v_lshlrev_b32 %[q], 30, %[q]\n"                                                                              \
        "v_fma_f32 %[cr], %[py], %[zx], 1.0\n"                                                                        \
        "v_fmac_f32 %[sr], %[px], %[zy]\n"                                                                            \
        "v_bfi_b32 %[sn], %[m], %[cr], %[sr]\n"                                                                       \
        "v_bfi_b32 %[cn], %[m], %[sr], %[cr]\n"                                                                       \
        "v_bitop3_b32 %[sn], %[q], %[sn], %[k] bitop3:0x6c\n"                                                         \
        "v_add_u32 %[q], 2.0, %[q]\n"                                                                                 \
        "v_bitop3_b32 %[cn], %[q], %[cn], %[k] bitop3:0x6c\n"

__device__ __forceinline__ void sincos_pk(float x, float &s_out, float &c_out, const ScConsts &k)
{
    float n, r, cr, sn, cn;
    v2f Z, P;
    sincos_pk_head(x, k, n, r, Z, P);
    int32_t q, m;
    asm(OCD_SC_TAIL
        : [q] "=&v"(q), [m] "=&v"(m), [cr] "=&v"(cr), [sr] "+&v"(r), [sn] "=&v"(sn), [cn] "=&v"(cn)
        : [n] "v"(n), [px] "v"(P.x), [py] "v"(P.y), [zx] "v"(Z.x), [zy] "v"(Z.y), [k] "s"(0x80000000u));
    s_out = sn;
    c_out = cn;
}

// V_SEG: the same, then the previous lane's (sin, cos) -- the current heading's at a segment's first lane -- and the
// step's increments cd = cos_pre * dd, sd = sin_pre * dd in the SAME statement (select fused into the DPP move; each
// DPP read comes >= 2 instructions after its source's write).
__device__ __forceinline__ void sincos_pk_seg(float x, const ScConsts &k, float s0, float c0, float dd,
                                              unsigned long long first_mask, float &s_out, float &c_out,
                                              float &s_pre, float &c_pre, float &sd, float &cd)
{
    float n, r, cr, sn, cn, sp, cp, sdd, cdd;
    v2f Z, P;
    sincos_pk_head(x, k, n, r, Z, P);
    int32_t q, m;
    asm volatile("s_mov_b64 vcc, %[fm]\n"
        OCD_SC_TAIL
        "v_cndmask_b32_dpp %[sp], %[sn], %[s0], vcc" OCD_SC_WAVE_SHR
        "v_mul_f32 %[sdd], %[sp], %[dd]\n"
        "v_cndmask_b32_dpp %[cp], %[cn], %[c0], vcc" OCD_SC_WAVE_SHR
        "v_mul_f32 %[cdd], %[cp], %[dd]\n"
        : [q] "=&v"(q), [m] "=&v"(m), [cr] "=&v"(cr), [sr] "+&v"(r), [sn] "=&v"(sn), [cn] "=&v"(cn),
          [sp] "=&v"(sp), [cp] "=&v"(cp), [sdd] "=&v"(sdd), [cdd] "=&v"(cdd)
        : [n] "v"(n), [px] "v"(P.x), [py] "v"(P.y), [zx] "v"(Z.x), [zy] "v"(Z.y), [k] "s"(0x80000000u),
          [s0] "v"(s0), [c0] "v"(c0), [dd] "v"(dd), [fm] "s"(first_mask)
        : "vcc");
    s_out = sn; c_out = cn; s_pre = sp; c_pre = cp; sd = sdd; cd = cdd;
}

} // namespace ocd
