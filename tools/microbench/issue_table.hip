// Microbenchmark: AGGREGATE issue cost of every instruction class of the planner kernels, per SIMD, at
// 1 / 2 / 4 / 8 wavefronts per SIMD -- on one CU (one workgroup, placement fully controlled) and on the
// whole chip (n_cus x W workgroups of 4 wavefronts; the placement is verified from HW_ID).
//
//   cost(class, W) = cycles a SIMD needs per wavefront-instruction of that class when W wavefronts share it
//                  = (cycles one wavefront's loop took) / (W x instructions in the loop)
//
// A lone wavefront hands its SIMD one instruction per ~4.4 cycles whatever the class; what W > 1 gains depends
// on how long the class occupies the SIMD (plain fp32 2 cycles on the 32-lane SIMD, packed / transcendental /
// 64-bit / cross-lane classes longer).  The table answers which classes saturate the SIMD in the planner
// kernels once several wavefronts share it; tools/issue_model.py applies it to
// the kernels' instruction histograms.
//
// Each class is measured as ONE dependent chain per wavefront (D) and as FOUR independent chains (I): the
// planner's streams sit between the two.  Composite rows time the kernels' own device functions (IEEE
// division scalar / packed, exp scalar / packed, sincos) from csrc/ocd_devmath.h.
//
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize \
//                -I../../l4dc-mpc-ocd_amd/csrc issue_table.hip -o issue_table
//   run:   ./issue_table [--json out.json]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ocd_devmath.h"
#include "ocd_chains.h"

#define REP4(S) S S S S
#define REP8(S) REP4(S) REP4(S)
#define REP16(S) REP4(REP4(S))
#define REP64(S) REP4(REP16(S))

typedef float v2f_ __attribute__((ext_vector_type(2)));

#define DPP_ROW " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define DPP_WAVE " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"

// operand list shared by every asm statement
#define OPS                                                                                                   \
    : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2),  \
      [p3] "+v"(p3)                                                                                           \
    : [k] "v"(k), [m] "v"(m), [pk] "v"(pk), [pm] "v"(pm), [sk] "s"(sk), [sm] "s"(sm), [idx] "v"(idx)          \
    : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"

// D: 64 dependent instructions; I: 16 x 4 independent ones
#define CLASS_D(T0) asm volatile(REP64(T0) OPS)
#define CLASS_I(T0, T1, T2, T3) asm volatile(REP16(T0 T1 T2 T3) OPS)

enum {
    C_FMA, C_MUL, C_ADD, C_MAX, C_MOV, C_MUL_SGPR, C_ADD_LIT, C_PK_FMA, C_PK_MUL, C_PK_ADD, C_ADD_DPP_ROW, C_ADD_DPP_WAVE,
    C_MOV_DPP_ROW, C_CNDMASK_DPP_WAVE, C_CNDMASK_VCC, C_CNDMASK_SGPR, C_CMP_VCC, C_CMP_SGPR, C_RCP, C_DIV_SCALE, C_DIV_FMAS,
    C_DIV_FIXUP, C_CVT_I32, C_CVT_F32, C_LSHL, C_XOR, C_AND_OR, C_ADD_U32, C_BPERMUTE, C_S_NOP, C_S_AND, C_S_MOV_VCC_CNDMASK,
    C_MIX_CNDVCC_1IN2, C_MIX_CNDVCC_1IN4, C_MIX_CNDVCC_1IN8, C_MIX_CNDSGPR_1IN4, C_MIX_CMP_CNDVCC, C_FMA_HALF_EXEC, C_CNDSGPR_HALF_EXEC,
    C_DIV_SCALAR, C_DIV_PACKED, C_EXP_SCALAR, C_EXP_PACKED, C_SINCOS,
    C_SEG_FWD_VTH, C_SEG_FWD_XY, C_SEG_BWD_XY, C_SEG_BWD_VTH, C_ROW_FWD_VTH, C_ROW_FWD_XY, C_COUNT
};

struct ClassInfo { const char *name; int ops_d, ops_i; const char *note; };

// ops = wavefront-instructions (or calls, for the composite rows) per loop trip of the D / I form
static const ClassInfo kClasses[C_COUNT] = {
    {"v_fma_f32", 64, 64, ""},
    {"v_mul_f32", 64, 64, ""},
    {"v_add_f32", 64, 64, ""},
    {"v_max_f32", 64, 64, ""},
    {"v_mov_b32", 64, 64, ""},
    {"v_mul_f32 (SGPR source)", 64, 64, "the chains' fr / dt operands"},
    {"v_add_f32 (literal source)", 64, 64, ""},
    {"v_pk_fma_f32", 64, 64, "two floats per lane"},
    {"v_pk_mul_f32", 64, 64, "two floats per lane"},
    {"v_pk_add_f32", 64, 64, "two floats per lane"},
    {"v_add_f32_dpp row_shr", 64, 64, "D form: + s_nop 1 per instruction (DPP hazard), counted as one"},
    {"v_add_f32_dpp wave_shr", 64, 64, "D form: + s_nop 1"},
    {"v_mov_b32_dpp row_shr", 64, 64, "D form: + s_nop 1"},
    {"v_cndmask_b32_dpp wave_shr (VCC)", 64, 64, "V_SEG / V_CHUNK boundary select; D form: + s_nop 1"},
    {"v_cndmask_b32 (VCC)", 64, 64, ""},
    {"v_cndmask_b32 (SGPR pair)", 64, 64, ""},
    {"v_cmp_gt_f32 -> VCC", 64, 64, ""},
    {"v_cmp_gt_f32 -> SGPR pair", 64, 64, ""},
    {"v_rcp_f32", 64, 64, "transcendental pipe"},
    {"v_div_scale_f32", 64, 64, ""},
    {"v_div_fmas_f32", 64, 64, ""},
    {"v_div_fixup_f32", 64, 64, ""},
    {"v_cvt_i32_f32", 64, 64, ""},
    {"v_cvt_f32_i32", 64, 64, ""},
    {"v_lshlrev_b32", 64, 64, ""},
    {"v_xor_b32", 64, 64, ""},
    {"v_and_or_b32", 64, 64, ""},
    {"v_add_u32", 64, 64, ""},
    {"ds_bpermute_b32", 64, 64, "D form: s_waitcnt after each; I form: after four"},
    {"s_nop 0", 64, 64, ""},
    {"s_and_b64", 64, 64, "SALU"},
    {"s_mov_b64 vcc + v_cndmask_b32", 64, 64, "pairs; SALU write feeding a VALU read"},
    {"mix: 1 v_cndmask(VCC) + 1 v_mul", 64, 64, "instructions; is the VCC select slow when diluted?"},
    {"mix: 1 v_cndmask(VCC) + 3 v_mul", 64, 64, "instructions"},
    {"mix: 1 v_cndmask(VCC) + 7 v_mul", 64, 64, "instructions; ~ the planner kernels' density"},
    {"mix: 1 v_cndmask(SGPR) + 3 v_mul", 64, 64, "instructions"},
    {"mix: v_cmp->VCC, v_mul, v_mul, v_cndmask(VCC)", 64, 64, "instructions; the compiler's usual select"},
    {"v_fma_f32, EXEC = lanes 0-31 only", 64, 64, "does a half-empty wavefront cost half?"},
    {"v_cndmask_b32 (SGPR pair), EXEC = lanes 0-31", 64, 64, ""},
    {"IEEE division, scalar (hipcc expansion)", 16, 16, "calls; ~11 instructions each"},
    {"IEEE division, two packed (div2_)", 16, 16, "calls = TWO quotients each"},
    {"exp_le1, scalar", 16, 16, "calls"},
    {"exp_le1_2, two packed", 16, 16, "calls = TWO exponentials each"},
    {"sincos_", 16, 16, "calls"},
    {"seg_fwd_vth<10> (ocd_chains.h), per round", 36, 36, "8 instructions per round, 2 of them v_cndmask_b32_dpp on VCC"},
    {"seg_fwd_xy<10>, per round", 36, 36, "4 instructions per round, 2 v_cndmask_b32_dpp"},
    {"seg_bwd_xy<10>, per round", 36, 36, "4 instructions per round, 2 v_cndmask_b32_dpp"},
    {"seg_bwd_vth<10>, per round", 36, 36, "12 instructions per round, 2 v_cndmask_b32_dpp"},
    {"row_fwd_vth<10>, per round", 36, 36, "8 instructions per round, 2 v_mov_b32_dpp"},
    {"row_fwd_xy<10>, per round", 36, 36, "2 v_add_f32_dpp + s_nop per round"},
};

template <int CLS, bool IND>
__global__ void __launch_bounds__(1024) bench(float *out, long long *cyc, unsigned *hwid, int iters, float seed)
{
    float a0 = threadIdx.x * 1e-3f + seed, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
    const float k = 0.999f, m = 1e-4f;
    v2f_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0};
    const v2f_ pk = {k, k}, pm = {m, m};
    float sk = __builtin_amdgcn_readfirstlane(k), sm = __builtin_amdgcn_readfirstlane(m);
    const int idx = ((threadIdx.x + 1) & 63) << 2;
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (CLS == C_FMA) {
            if (IND) CLASS_I("v_fma_f32 %[a0], %[a0], %[k], %[m]\n", "v_fma_f32 %[a1], %[a1], %[k], %[m]\n", "v_fma_f32 %[a2], %[a2], %[k], %[m]\n", "v_fma_f32 %[a3], %[a3], %[k], %[m]\n");
            else CLASS_D("v_fma_f32 %[a0], %[a0], %[k], %[m]\n");
        } else if constexpr (CLS == C_MUL) {
            if (IND) CLASS_I("v_mul_f32 %[a0], %[a0], %[k]\n", "v_mul_f32 %[a1], %[a1], %[k]\n", "v_mul_f32 %[a2], %[a2], %[k]\n", "v_mul_f32 %[a3], %[a3], %[k]\n");
            else CLASS_D("v_mul_f32 %[a0], %[a0], %[k]\n");
        } else if constexpr (CLS == C_ADD) {
            if (IND) CLASS_I("v_add_f32 %[a0], %[a0], %[m]\n", "v_add_f32 %[a1], %[a1], %[m]\n", "v_add_f32 %[a2], %[a2], %[m]\n", "v_add_f32 %[a3], %[a3], %[m]\n");
            else CLASS_D("v_add_f32 %[a0], %[a0], %[m]\n");
        } else if constexpr (CLS == C_MAX) {
            if (IND) CLASS_I("v_max_f32 %[a0], %[a0], %[m]\n", "v_max_f32 %[a1], %[a1], %[m]\n", "v_max_f32 %[a2], %[a2], %[m]\n", "v_max_f32 %[a3], %[a3], %[m]\n");
            else CLASS_D("v_max_f32 %[a0], %[a0], %[m]\n");
        } else if constexpr (CLS == C_MOV) {
            if (IND) CLASS_I("v_mov_b32 %[a0], %[a1]\n", "v_mov_b32 %[a1], %[a2]\n", "v_mov_b32 %[a2], %[a3]\n", "v_mov_b32 %[a3], %[k]\n");
            else CLASS_D("v_mov_b32 %[a0], %[a0]\n");
        } else if constexpr (CLS == C_MUL_SGPR) {
            if (IND) CLASS_I("v_mul_f32 %[a0], %[sk], %[a0]\n", "v_mul_f32 %[a1], %[sk], %[a1]\n", "v_mul_f32 %[a2], %[sk], %[a2]\n", "v_mul_f32 %[a3], %[sk], %[a3]\n");
            else CLASS_D("v_mul_f32 %[a0], %[sk], %[a0]\n");
        } else if constexpr (CLS == C_ADD_LIT) {
            if (IND) CLASS_I("v_add_f32 %[a0], 0x3a83126f, %[a0]\n", "v_add_f32 %[a1], 0x3a83126f, %[a1]\n", "v_add_f32 %[a2], 0x3a83126f, %[a2]\n", "v_add_f32 %[a3], 0x3a83126f, %[a3]\n");
            else CLASS_D("v_add_f32 %[a0], 0x3a83126f, %[a0]\n");
        } else if constexpr (CLS == C_PK_FMA) {
            if (IND) CLASS_I("v_pk_fma_f32 %[p0], %[p0], %[pk], %[pm]\n", "v_pk_fma_f32 %[p1], %[p1], %[pk], %[pm]\n", "v_pk_fma_f32 %[p2], %[p2], %[pk], %[pm]\n", "v_pk_fma_f32 %[p3], %[p3], %[pk], %[pm]\n");
            else CLASS_D("v_pk_fma_f32 %[p0], %[p0], %[pk], %[pm]\n");
        } else if constexpr (CLS == C_PK_MUL) {
            if (IND) CLASS_I("v_pk_mul_f32 %[p0], %[p0], %[pk]\n", "v_pk_mul_f32 %[p1], %[p1], %[pk]\n", "v_pk_mul_f32 %[p2], %[p2], %[pk]\n", "v_pk_mul_f32 %[p3], %[p3], %[pk]\n");
            else CLASS_D("v_pk_mul_f32 %[p0], %[p0], %[pk]\n");
        } else if constexpr (CLS == C_PK_ADD) {
            if (IND) CLASS_I("v_pk_add_f32 %[p0], %[p0], %[pm]\n", "v_pk_add_f32 %[p1], %[p1], %[pm]\n", "v_pk_add_f32 %[p2], %[p2], %[pm]\n", "v_pk_add_f32 %[p3], %[p3], %[pm]\n");
            else CLASS_D("v_pk_add_f32 %[p0], %[p0], %[pm]\n");
        } else if constexpr (CLS == C_ADD_DPP_ROW) {
            if (IND) CLASS_I("v_add_f32_dpp %[a0], %[a0], %[m]" DPP_ROW, "v_add_f32_dpp %[a1], %[a1], %[m]" DPP_ROW, "v_add_f32_dpp %[a2], %[a2], %[m]" DPP_ROW, "v_add_f32_dpp %[a3], %[a3], %[m]" DPP_ROW);
            else CLASS_D("v_add_f32_dpp %[a0], %[a0], %[m]" DPP_ROW "s_nop 1\n");
        } else if constexpr (CLS == C_ADD_DPP_WAVE) {
            if (IND) CLASS_I("v_add_f32_dpp %[a0], %[a0], %[m]" DPP_WAVE, "v_add_f32_dpp %[a1], %[a1], %[m]" DPP_WAVE, "v_add_f32_dpp %[a2], %[a2], %[m]" DPP_WAVE, "v_add_f32_dpp %[a3], %[a3], %[m]" DPP_WAVE);
            else CLASS_D("v_add_f32_dpp %[a0], %[a0], %[m]" DPP_WAVE "s_nop 1\n");
        } else if constexpr (CLS == C_MOV_DPP_ROW) {
            if (IND) CLASS_I("v_mov_b32_dpp %[a0], %[a0]" DPP_ROW, "v_mov_b32_dpp %[a1], %[a1]" DPP_ROW, "v_mov_b32_dpp %[a2], %[a2]" DPP_ROW, "v_mov_b32_dpp %[a3], %[a3]" DPP_ROW);
            else CLASS_D("v_mov_b32_dpp %[a0], %[a0]" DPP_ROW "s_nop 1\n");
        } else if constexpr (CLS == C_CNDMASK_DPP_WAVE) {
            if (IND) CLASS_I("v_cndmask_b32_dpp %[a0], %[a0], %[m], vcc" DPP_WAVE, "v_cndmask_b32_dpp %[a1], %[a1], %[m], vcc" DPP_WAVE, "v_cndmask_b32_dpp %[a2], %[a2], %[m], vcc" DPP_WAVE, "v_cndmask_b32_dpp %[a3], %[a3], %[m], vcc" DPP_WAVE);
            else CLASS_D("v_cndmask_b32_dpp %[a0], %[a0], %[m], vcc" DPP_WAVE "s_nop 1\n");
        } else if constexpr (CLS == C_CNDMASK_VCC) {
            if (IND) CLASS_I("v_cndmask_b32 %[a0], %[a0], %[m], vcc\n", "v_cndmask_b32 %[a1], %[a1], %[m], vcc\n", "v_cndmask_b32 %[a2], %[a2], %[m], vcc\n", "v_cndmask_b32 %[a3], %[a3], %[m], vcc\n");
            else CLASS_D("v_cndmask_b32 %[a0], %[a0], %[m], vcc\n");
        } else if constexpr (CLS == C_CNDMASK_SGPR) {
            if (IND) CLASS_I("v_cndmask_b32 %[a0], %[a0], %[m], s[20:21]\n", "v_cndmask_b32 %[a1], %[a1], %[m], s[20:21]\n", "v_cndmask_b32 %[a2], %[a2], %[m], s[22:23]\n", "v_cndmask_b32 %[a3], %[a3], %[m], s[22:23]\n");
            else CLASS_D("v_cndmask_b32 %[a0], %[a0], %[m], s[20:21]\n");
        } else if constexpr (CLS == C_CMP_VCC) {
            if (IND) CLASS_I("v_cmp_gt_f32 vcc, %[a0], %[m]\n", "v_cmp_gt_f32 vcc, %[a1], %[m]\n", "v_cmp_gt_f32 vcc, %[a2], %[m]\n", "v_cmp_gt_f32 vcc, %[a3], %[m]\n");
            else CLASS_D("v_cmp_gt_f32 vcc, %[a0], %[m]\n");
        } else if constexpr (CLS == C_CMP_SGPR) {
            if (IND) CLASS_I("v_cmp_gt_f32 s[20:21], %[a0], %[m]\n", "v_cmp_gt_f32 s[22:23], %[a1], %[m]\n", "v_cmp_gt_f32 s[24:25], %[a2], %[m]\n", "v_cmp_gt_f32 s[26:27], %[a3], %[m]\n");
            else CLASS_D("v_cmp_gt_f32 s[20:21], %[a0], %[m]\n");
        } else if constexpr (CLS == C_RCP) {
            if (IND) CLASS_I("v_rcp_f32 %[a0], %[a0]\n", "v_rcp_f32 %[a1], %[a1]\n", "v_rcp_f32 %[a2], %[a2]\n", "v_rcp_f32 %[a3], %[a3]\n");
            else CLASS_D("v_rcp_f32 %[a0], %[a0]\n");
        } else if constexpr (CLS == C_DIV_SCALE) {
            if (IND) CLASS_I("v_div_scale_f32 %[a0], vcc, %[a0], %[k], %[a0]\n", "v_div_scale_f32 %[a1], vcc, %[a1], %[k], %[a1]\n", "v_div_scale_f32 %[a2], vcc, %[a2], %[k], %[a2]\n", "v_div_scale_f32 %[a3], vcc, %[a3], %[k], %[a3]\n");
            else CLASS_D("v_div_scale_f32 %[a0], vcc, %[a0], %[k], %[a0]\n");
        } else if constexpr (CLS == C_DIV_FMAS) {
            if (IND) CLASS_I("v_div_fmas_f32 %[a0], %[a0], %[k], %[m]\n", "v_div_fmas_f32 %[a1], %[a1], %[k], %[m]\n", "v_div_fmas_f32 %[a2], %[a2], %[k], %[m]\n", "v_div_fmas_f32 %[a3], %[a3], %[k], %[m]\n");
            else CLASS_D("v_div_fmas_f32 %[a0], %[a0], %[k], %[m]\n");
        } else if constexpr (CLS == C_DIV_FIXUP) {
            if (IND) CLASS_I("v_div_fixup_f32 %[a0], %[a0], %[k], %[m]\n", "v_div_fixup_f32 %[a1], %[a1], %[k], %[m]\n", "v_div_fixup_f32 %[a2], %[a2], %[k], %[m]\n", "v_div_fixup_f32 %[a3], %[a3], %[k], %[m]\n");
            else CLASS_D("v_div_fixup_f32 %[a0], %[a0], %[k], %[m]\n");
        } else if constexpr (CLS == C_CVT_I32) {
            if (IND) CLASS_I("v_cvt_i32_f32 %[a0], %[a0]\n", "v_cvt_i32_f32 %[a1], %[a1]\n", "v_cvt_i32_f32 %[a2], %[a2]\n", "v_cvt_i32_f32 %[a3], %[a3]\n");
            else CLASS_D("v_cvt_i32_f32 %[a0], %[a0]\n");
        } else if constexpr (CLS == C_CVT_F32) {
            if (IND) CLASS_I("v_cvt_f32_i32 %[a0], %[a0]\n", "v_cvt_f32_i32 %[a1], %[a1]\n", "v_cvt_f32_i32 %[a2], %[a2]\n", "v_cvt_f32_i32 %[a3], %[a3]\n");
            else CLASS_D("v_cvt_f32_i32 %[a0], %[a0]\n");
        } else if constexpr (CLS == C_LSHL) {
            if (IND) CLASS_I("v_lshlrev_b32 %[a0], 1, %[a0]\n", "v_lshlrev_b32 %[a1], 1, %[a1]\n", "v_lshlrev_b32 %[a2], 1, %[a2]\n", "v_lshlrev_b32 %[a3], 1, %[a3]\n");
            else CLASS_D("v_lshlrev_b32 %[a0], 1, %[a0]\n");
        } else if constexpr (CLS == C_XOR) {
            if (IND) CLASS_I("v_xor_b32 %[a0], %[a0], %[m]\n", "v_xor_b32 %[a1], %[a1], %[m]\n", "v_xor_b32 %[a2], %[a2], %[m]\n", "v_xor_b32 %[a3], %[a3], %[m]\n");
            else CLASS_D("v_xor_b32 %[a0], %[a0], %[m]\n");
        } else if constexpr (CLS == C_AND_OR) {
            if (IND) CLASS_I("v_and_or_b32 %[a0], %[a0], %[k], %[m]\n", "v_and_or_b32 %[a1], %[a1], %[k], %[m]\n", "v_and_or_b32 %[a2], %[a2], %[k], %[m]\n", "v_and_or_b32 %[a3], %[a3], %[k], %[m]\n");
            else CLASS_D("v_and_or_b32 %[a0], %[a0], %[k], %[m]\n");
        } else if constexpr (CLS == C_ADD_U32) {
            if (IND) CLASS_I("v_add_u32 %[a0], %[a0], %[m]\n", "v_add_u32 %[a1], %[a1], %[m]\n", "v_add_u32 %[a2], %[a2], %[m]\n", "v_add_u32 %[a3], %[a3], %[m]\n");
            else CLASS_D("v_add_u32 %[a0], %[a0], %[m]\n");
        } else if constexpr (CLS == C_BPERMUTE) {
            if (IND) CLASS_I("ds_bpermute_b32 %[a0], %[idx], %[a0]\n", "ds_bpermute_b32 %[a1], %[idx], %[a1]\n", "ds_bpermute_b32 %[a2], %[idx], %[a2]\n", "ds_bpermute_b32 %[a3], %[idx], %[a3]\ns_waitcnt lgkmcnt(0)\n");
            else CLASS_D("ds_bpermute_b32 %[a0], %[idx], %[a0]\ns_waitcnt lgkmcnt(0)\n");
        } else if constexpr (CLS == C_S_NOP) {
            CLASS_D("s_nop 0\n");
        } else if constexpr (CLS == C_S_AND) {
            if (IND) CLASS_I("s_and_b64 s[20:21], s[20:21], exec\n", "s_and_b64 s[22:23], s[22:23], exec\n", "s_and_b64 s[24:25], s[24:25], exec\n", "s_and_b64 s[26:27], s[26:27], exec\n");
            else CLASS_D("s_and_b64 s[20:21], s[20:21], exec\n");
        } else if constexpr (CLS == C_S_MOV_VCC_CNDMASK) {
            asm volatile(REP16("s_mov_b64 vcc, s[20:21]\n v_cndmask_b32 %[a0], %[a0], %[m], vcc\n"
                               "s_mov_b64 vcc, s[22:23]\n v_cndmask_b32 %[a1], %[a1], %[m], vcc\n") OPS);
        } else if constexpr (CLS == C_MIX_CNDVCC_1IN2) {
            asm volatile(REP16("v_cndmask_b32 %[a0], %[a0], %[m], vcc\n v_mul_f32 %[a1], %[a1], %[k]\n v_cndmask_b32 %[a2], %[a2], %[m], vcc\n v_mul_f32 %[a3], %[a3], %[k]\n") OPS);
        } else if constexpr (CLS == C_MIX_CNDVCC_1IN4) {
            asm volatile(REP16("v_cndmask_b32 %[a0], %[a0], %[m], vcc\n v_mul_f32 %[a1], %[a1], %[k]\n v_mul_f32 %[a2], %[a2], %[k]\n v_mul_f32 %[a3], %[a3], %[k]\n") OPS);
        } else if constexpr (CLS == C_MIX_CNDVCC_1IN8) {
            asm volatile(REP8("v_cndmask_b32 %[a0], %[a0], %[m], vcc\n v_mul_f32 %[a1], %[a1], %[k]\n v_mul_f32 %[a2], %[a2], %[k]\n v_mul_f32 %[a3], %[a3], %[k]\n"
                                   "v_mul_f32 %[a1], %[a1], %[k]\n v_mul_f32 %[a2], %[a2], %[k]\n v_mul_f32 %[a3], %[a3], %[k]\n v_mul_f32 %[a1], %[a1], %[k]\n") OPS);
        } else if constexpr (CLS == C_MIX_CNDSGPR_1IN4) {
            asm volatile(REP16("v_cndmask_b32 %[a0], %[a0], %[m], s[20:21]\n v_mul_f32 %[a1], %[a1], %[k]\n v_mul_f32 %[a2], %[a2], %[k]\n v_mul_f32 %[a3], %[a3], %[k]\n") OPS);
        } else if constexpr (CLS == C_MIX_CMP_CNDVCC) {
            asm volatile(REP16("v_cmp_gt_f32 vcc, %[a1], %[m]\n v_mul_f32 %[a1], %[a1], %[k]\n v_mul_f32 %[a2], %[a2], %[k]\n v_cndmask_b32 %[a0], %[a0], %[m], vcc\n") OPS);
        } else if constexpr (CLS == C_FMA_HALF_EXEC) {
            asm volatile("s_mov_b64 s[24:25], exec\n s_mov_b64 exec, 0xffffffff\n s_nop 4\n" REP64("v_fma_f32 %[a0], %[a0], %[k], %[m]\n") "s_mov_b64 exec, s[24:25]\n s_nop 4\n" OPS);
        } else if constexpr (CLS == C_CNDSGPR_HALF_EXEC) {
            asm volatile("s_mov_b64 s[24:25], exec\n s_mov_b64 exec, 0xffffffff\n s_nop 4\n" REP64("v_cndmask_b32 %[a0], %[a0], %[m], s[20:21]\n") "s_mov_b64 exec, s[24:25]\n s_nop 4\n" OPS);
        } else if constexpr (CLS == C_SEG_FWD_VTH) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ocd::seg_fwd_vth<10>(a0, a1, a2, a3, k, m, sk, sm, 0x0040100401004010ull);
        } else if constexpr (CLS == C_SEG_FWD_XY) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ocd::seg_fwd_xy<10>(a0, a1, a2, a3, k, m, 0x0040100401004010ull);
        } else if constexpr (CLS == C_SEG_BWD_XY) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ocd::seg_bwd_xy<10>(a0, a1, a2, a3, 0x0040100401004010ull);
        } else if constexpr (CLS == C_SEG_BWD_VTH) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ocd::seg_bwd_vth<10>(a0, a1, a2, a3, k, m, p0.x, p0.y, sk, sm, 0x0040100401004010ull);
        } else if constexpr (CLS == C_ROW_FWD_VTH) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ocd::row_fwd_vth<10>(a0, a1, k, m, sk, sm);
        } else if constexpr (CLS == C_ROW_FWD_XY) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ocd::row_fwd_xy<10>(a0, a1, k, m);
        } else if constexpr (CLS == C_DIV_SCALAR) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (IND && (j & 1)) a1 = k / a1;
                else a0 = a1 / a0;
            }
        } else if constexpr (CLS == C_DIV_PACKED) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (IND) { const ocd::v2f q = ocd::div2_(ocd::v2f{k, k}, ocd::v2f{(j & 1) ? p1.x : p0.x, (j & 1) ? p1.y : p0.y}); if (j & 1) { p1.x = q.x; p1.y = q.y; } else { p0.x = q.x; p0.y = q.y; } }
                else { const ocd::v2f q = ocd::div2_(ocd::v2f{p1.x, p1.y}, ocd::v2f{p0.x, p0.y}); p0.x = q.x; p0.y = q.y; }
            }
        } else if constexpr (CLS == C_EXP_SCALAR) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (IND && (j & 1)) a1 = ocd::exp_le1(a1 - 2.0f);
                else a0 = ocd::exp_le1(a0 - 2.0f);
            }
        } else if constexpr (CLS == C_EXP_PACKED) {
            const ocd::PkConsts pkc = ocd::pk_consts();
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (IND && (j & 1)) { const ocd::v2f e = ocd::exp_le1_2(ocd::v2f{p1.x - 2.0f, p1.y - 2.0f}, pkc); p1.x = e.x; p1.y = e.y; }
                else { const ocd::v2f e = ocd::exp_le1_2(ocd::v2f{p0.x - 2.0f, p0.y - 2.0f}, pkc); p0.x = e.x; p0.y = e.y; }
            }
        } else if constexpr (CLS == C_SINCOS) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float s_, c_;
                if (IND && (j & 1)) { ocd::sincos_(a1, s_, c_); a1 = s_ + c_; }
                else { ocd::sincos_(a0, s_, c_); a0 = s_ + c_; }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if ((threadIdx.x & 63) == 0) {
        const size_t wv = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cyc[2 * wv] = t1 - t0;
        cyc[2 * wv + 1] = r1 - r0;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        hwid[2 * wv] = hw;
        hwid[2 * wv + 1] = xcc;
    }
}

struct Result { double cyc_per_op_simd, ns_per_op_simd, wave_cycles_per_op, clock_ghz, resident_frac; int min_w, max_w; };

static int g_cus = 256;

// blocks x threads; W = wavefronts per SIMD this shape is meant to give
template <int CLS, bool IND>
static Result run(int blocks, int threads, int W)
{
    const int waves = blocks * (threads / 64);
    float *out; long long *cyc; unsigned *hw;
    hipMalloc(&out, (size_t)blocks * threads * sizeof(float));
    hipMalloc(&cyc, 2 * waves * sizeof(long long));
    hipMalloc(&hw, 2 * waves * sizeof(unsigned));
    const int ops = IND ? kClasses[CLS].ops_i : kClasses[CLS].ops_d;
    const int iters = (CLS >= C_DIV_SCALAR) ? 1500 : ((CLS == C_BPERMUTE) ? 1500 : 6000);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    bench<CLS, IND><<<blocks, threads>>>(out, cyc, hw, iters / 4, 1.0f);       // warm-up (clocks, code)
    hipEventRecord(e0);
    bench<CLS, IND><<<blocks, threads>>>(out, cyc, hw, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * waves); hipMemcpy(h.data(), cyc, 2 * waves * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<unsigned> id(2 * waves); hipMemcpy(id.data(), hw, 2 * waves * sizeof(unsigned), hipMemcpyDeviceToHost);
    double mean = 0, real = 0; for (int i = 0; i < waves; ++i) { mean += (double)h[2 * i]; real += (double)h[2 * i + 1]; } mean /= waves; real /= waves;
    // placement: wavefronts per (xcc, se, sh, cu, simd)
    std::map<unsigned, int> per_simd;
    for (int i = 0; i < waves; ++i) {
        const unsigned v = id[2 * i], x = id[2 * i + 1] & 0xf;
        const unsigned simd = (v >> 4) & 3, cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 7;
        per_simd[(x << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd] += 1;
    }
    int mn = 1 << 30, mx = 0;
    for (auto &kv : per_simd) { mn = kv.second < mn ? kv.second : mn; mx = kv.second > mx ? kv.second : mx; }
    const double total_ops = (double)iters * ops;
    Result r;
    r.wave_cycles_per_op = mean / total_ops;
    r.cyc_per_op_simd = mean / (total_ops * W);
    r.ns_per_op_simd = (double)ms * 1e6 / (total_ops * W);
    r.clock_ghz = real > 0 ? mean / real * 0.1 : 0.0;       // s_memrealtime counts at 100 MHz
    r.resident_frac = real * 10.0 / ((double)ms * 1e6);     // a wavefront's lifetime / the launch's
    r.min_w = mn; r.max_w = mx;
    hipFree(out); hipFree(cyc); hipFree(hw);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return r;
}

static FILE *g_json = nullptr;
static bool g_first = true;

template <int CLS>
static void row()
{
    // one CU: a workgroup of 1 / 4 / 8 / 16 wavefronts (1 wavefront; 1, 2, 4 per SIMD); whole chip: 1, 2, 4, 8 per SIMD
    struct Shape { const char *tag; int blocks, threads, W; };
    const Shape shapes[] = {
        {"cu_1wave", 1, 64, 1}, {"cu_1", 1, 256, 1}, {"cu_2", 1, 512, 2}, {"cu_4", 1, 1024, 4},
        {"chip_1", g_cus, 256, 1}, {"chip_2", g_cus * 2, 256, 2}, {"chip_4", g_cus * 4, 256, 4}, {"chip_8", g_cus * 8, 256, 8},
    };
    for (int ind = 0; ind < 2; ++ind) {
        if (CLS == C_S_NOP && ind) continue;
        if (CLS == C_S_MOV_VCC_CNDMASK && ind) continue;
        if (((CLS >= C_MIX_CNDVCC_1IN2 && CLS <= C_CNDSGPR_HALF_EXEC) || CLS >= C_SEG_FWD_VTH) && ind) continue;
        printf("%-42s %s |", kClasses[CLS].name, ind ? "I" : "D");
        if (g_json) fprintf(g_json, "%s\n  {\"class\": \"%s\", \"form\": \"%s\", \"note\": \"%s\"", g_first ? "" : ",", kClasses[CLS].name, ind ? "I" : "D", kClasses[CLS].note);
        g_first = false;
        for (const Shape &s : shapes) {
            const Result r = ind ? run<CLS, true>(s.blocks, s.threads, s.W) : run<CLS, false>(s.blocks, s.threads, s.W);
            printf(" %s %6.2f/%5.2fns@%.2f", s.tag, r.cyc_per_op_simd, r.ns_per_op_simd, r.clock_ghz);
            if (r.min_w != s.W || r.max_w != s.W) printf("(%d-%d/simd)", r.min_w, r.max_w);
            if (g_json) fprintf(g_json, ", \"%s\": {\"cycles_per_op_per_simd\": %.3f, \"ns_per_op_per_simd\": %.4f, \"wave_cycles_per_op\": %.3f, \"clock_ghz\": %.3f, \"wave_lifetime_frac\": %.3f, \"waves_per_simd_min\": %d, \"waves_per_simd_max\": %d}",
                                s.tag, r.cyc_per_op_simd, r.ns_per_op_simd, r.wave_cycles_per_op, r.clock_ghz, r.resident_frac, r.min_w, r.max_w);
        }
        printf("\n");
        fflush(stdout);
        if (g_json) { fprintf(g_json, "}"); fflush(g_json); }
    }
}

template <int CLS>
static void rows()
{
    if constexpr (CLS < C_COUNT) { row<CLS>(); rows<CLS + 1>(); }
}

int main(int argc, char **argv)
{
    for (int i = 1; i + 1 < argc; ++i)
        if (!strcmp(argv[i], "--json")) g_json = fopen(argv[i + 1], "w");
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    g_cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.gcnArchName, g_cus, prop.clockRate / 1e3);
    printf("cycles a SIMD spends per wavefront-instruction (or call) of the class, by shape: cu_N = one workgroup of 4N wavefronts on one CU "
           "(cu_1wave: a single wavefront), chip_N = n_cus x N workgroups of 4 wavefronts; D = one dependent chain per wavefront, I = four independent chains\n");
    if (g_json) fprintf(g_json, "{\"device\": \"%s\", \"cus\": %d, \"unit\": \"s_memtime cycles per wavefront-instruction per SIMD\", \"rows\": [", prop.gcnArchName, g_cus);
    rows<0>();
    if (g_json) { fprintf(g_json, "\n]}\n"); fclose(g_json); }
    return 0;
}
