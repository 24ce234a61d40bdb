// ocd_chains.h -- hand-scheduled horizon recurrences of the DPP variants (V_ROW, V_SEG), gfx950.
//
// A lone wavefront issues one instruction per ~4.6 cycles whatever its dependencies (measured:
// tools/microbench/valu_latency.hip), so a recurrence round costs its instruction count, s_nop
// included.  hipcc neither fuses the add into the DPP move when the boundary lane has to keep its value
// (v_add_f32_dpp with the destination tied as `old`), nor interleaves two chains so that each covers
// the other's DPP hazard (a VGPR written by a VALU instruction needs 2 wait states before a DPP
// instruction reads it).  Each recurrence below is ONE asm statement -- hipcc pads nothing inside it --
// scheduled so that every DPP read comes >= 2 instructions after the write; the arithmetic is the
// kernel's C++ formulation instruction for instruction (same operations, same operands, same order).
//
//   V_ROW: a trajectory owns a 16-lane row; row_shr:1 / row_shl:1 have no source in the row's first /
//          last lane, which therefore keeps its value (bound_ctrl off): the boundary costs nothing.
//   V_SEG: segments of H lanes anywhere in the wavefront; wave_shr:1 / wave_shl:1 cross segment
//          boundaries, so the move is a v_cndmask_b32_dpp that selects the boundary value in the
//          segment's first / last lane (VCC = boundary mask, loaded once per statement).
//
// The number of rounds is H-1: OCD_CHAIN_ROUNDS(HT, STMT) instantiates STMT with the repeat macro of the
// specialised horizon.
#pragma once
#include <hip/hip_runtime.h>

#include "ocd_devmath.h"

#define OCD_REP1(S) S
#define OCD_REP2(S) S S
#define OCD_REP4(S) S S S S
#define OCD_REP5(S) S S S S S
#define OCD_REP9(S) OCD_REP4(S) OCD_REP5(S)
#define OCD_REP14(S) OCD_REP9(S) OCD_REP5(S)
#define OCD_REP0(S)
#define OCD_REP3(S) S S S
#define OCD_REP8(S) OCD_REP4(S) OCD_REP4(S)
#define OCD_REP13(S) OCD_REP9(S) OCD_REP4(S)

// STMT(REP) must expand to a statement using REP("...") for the round body
#define OCD_CHAIN_ROUNDS(HT, STMT)                                   \
    do {                                                             \
        if constexpr ((HT) == 2) { STMT(OCD_REP1) }                  \
        else if constexpr ((HT) == 3) { STMT(OCD_REP2) }             \
        else if constexpr ((HT) == 5) { STMT(OCD_REP4) }             \
        else if constexpr ((HT) == 6) { STMT(OCD_REP5) }             \
        else if constexpr ((HT) == 10) { STMT(OCD_REP9) }            \
        else if constexpr ((HT) == 15) { STMT(OCD_REP14) }           \
        else static_assert((HT) == 2, "no repeat macro for this horizon"); \
    } while (0)

namespace ocd {

template <int HT> struct chain_supported { static constexpr bool value = HT == 2 || HT == 3 || HT == 5 || HT == 6 || HT == 10 || HT == 15; };

#define OCD_ROW_SHR " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OCD_ROW_SHL " row_shl:1 row_mask:0xf bank_mask:0xf\n"
#define OCD_WAVE_SHR " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define OCD_WAVE_SHL " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"

// the same with H-2 repeats: the V_SEG chains peel their first round, in which every lane still holds the boundary value
// -- the round reads the boundary registers directly and the recurrence registers are outputs only (no copies in)
#define OCD_CHAIN_ROUNDS_M1(HT, STMT)                                \
    do {                                                             \
        if constexpr ((HT) == 2) { STMT(OCD_REP0) }                  \
        else if constexpr ((HT) == 3) { STMT(OCD_REP1) }             \
        else if constexpr ((HT) == 5) { STMT(OCD_REP3) }             \
        else if constexpr ((HT) == 6) { STMT(OCD_REP4) }             \
        else if constexpr ((HT) == 10) { STMT(OCD_REP8) }            \
        else if constexpr ((HT) == 15) { STMT(OCD_REP13) }           \
        else static_assert((HT) == 2, "no repeat macro for this horizon"); \
    } while (0)

// ---- forward speed / heading recurrence ----
//   v_next = v + (a_c - fr * (v * v)) * dt ; th_next = th + wdt ; (v, th) <- (v_next, th_next) of the lane below
// The heading chain runs half a round ahead of the speed chain (thn is the NEXT round's th_next), so its
// two instructions fill the two wait states between the speed chain's add and its DPP move.
template <int HT>
__device__ __forceinline__ void row_fwd_vth(float &v, float &th, float a_c, float wdt, float fr, float dt)
{
    float thn = th + wdt, tmp;
#define OCD_STMT(REP)                                                                                     \
    asm volatile(REP("v_mul_f32 %[t], %[v], %[v]\n"                                                       \
                     "v_mul_f32 %[t], %[fr], %[t]\n"                                                      \
                     "v_sub_f32 %[t], %[ac], %[t]\n"                                                      \
                     "v_mul_f32 %[t], %[dt], %[t]\n"                                                      \
                     "v_add_f32 %[t], %[v], %[t]\n"                                                       \
                     "v_mov_b32_dpp %[th], %[thn]" OCD_ROW_SHR                                            \
                     "v_add_f32 %[thn], %[th], %[wdt]\n"                                                  \
                     "v_mov_b32_dpp %[v], %[t]" OCD_ROW_SHR)                                              \
                 : [v] "+&v"(v), [th] "+&v"(th), [thn] "+&v"(thn), [t] "=&v"(tmp)                           \
                 : [ac] "v"(a_c), [wdt] "v"(wdt), [fr] "s"(fr), [dt] "s"(dt));
    OCD_CHAIN_ROUNDS(HT, OCD_STMT);
#undef OCD_STMT
}

// (v, th) are outputs: every lane starts the recurrence at (ev, eth)
template <int HT>
__device__ __forceinline__ void seg_fwd_vth(float &v, float &th, float ev, float eth, float a_c, float wdt, float fr,
                                            float dt, unsigned long long first_mask)
{
    float thn = eth + wdt, tmp;
#define OCD_STMT(REP)                                                                                     \
    asm volatile("s_mov_b64 vcc, %[m]\n"                                                                  \
                 "v_mul_f32 %[t], %[ev], %[ev]\n"                                                         \
                 "v_mul_f32 %[t], %[fr], %[t]\n"                                                          \
                 "v_sub_f32 %[t], %[ac], %[t]\n"                                                          \
                 "v_mul_f32 %[t], %[dt], %[t]\n"                                                          \
                 "v_add_f32 %[t], %[ev], %[t]\n"                                                          \
                 "v_cndmask_b32_dpp %[th], %[thn], %[eth], vcc" OCD_WAVE_SHR                              \
                 "v_add_f32 %[thn], %[th], %[wdt]\n"                                                      \
                 "v_cndmask_b32_dpp %[v], %[t], %[ev], vcc" OCD_WAVE_SHR                                  \
                 REP("v_mul_f32 %[t], %[v], %[v]\n"                                                       \
                     "v_mul_f32 %[t], %[fr], %[t]\n"                                                      \
                     "v_sub_f32 %[t], %[ac], %[t]\n"                                                      \
                     "v_mul_f32 %[t], %[dt], %[t]\n"                                                      \
                     "v_add_f32 %[t], %[v], %[t]\n"                                                       \
                     "v_cndmask_b32_dpp %[th], %[thn], %[eth], vcc" OCD_WAVE_SHR                          \
                     "v_add_f32 %[thn], %[th], %[wdt]\n"                                                  \
                     "v_cndmask_b32_dpp %[v], %[t], %[ev], vcc" OCD_WAVE_SHR)                             \
                 : [v] "=&v"(v), [th] "=&v"(th), [thn] "+&v"(thn), [t] "=&v"(tmp)                        \
                 : [ac] "v"(a_c), [wdt] "v"(wdt), [ev] "v"(ev), [eth] "v"(eth), [fr] "s"(fr), [dt] "s"(dt), \
                   [m] "s"(first_mask)                                                                    \
                 : "vcc");
    OCD_CHAIN_ROUNDS_M1(HT, OCD_STMT);
#undef OCD_STMT
}

// ---- forward position recurrence:  x <- (x + cd) of the lane below, y <- (y + sd) of the lane below ----
// V_ROW: fused into the DPP add with the increment of the lane below (cdb, sdb) brought up once:
//        x[t] = x[t-1] + cd[t-1]; the row's lane 0 is not written and keeps ex.
template <int HT>
__device__ __forceinline__ void row_fwd_xy(float &x, float &y, float cdb, float sdb)
{
#define OCD_STMT(REP)                                                                                     \
    asm volatile("s_nop 1\n"                                                                              \
                 REP("v_add_f32_dpp %[x], %[x], %[cdb]" OCD_ROW_SHR                                       \
                     "v_add_f32_dpp %[y], %[y], %[sdb]" OCD_ROW_SHR                                       \
                     "s_nop 0\n")                                                                         \
                 : [x] "+&v"(x), [y] "+&v"(y) : [cdb] "v"(cdb), [sdb] "v"(sdb));
    OCD_CHAIN_ROUNDS(HT, OCD_STMT);
#undef OCD_STMT
}

// H = 10 with the target-speed feature's adjoint in the nine hazard slots (it needs the post-step speed and
// heading only, which are complete before the position recurrence starts):
//   vel = vn*sn ; dv = vel - tgt ; sq = dv*dv ; g = (sq <= bound ? w0 : 0)*2 ; g_dv = g*dv ;
//   qv = g_dv*sn ; qth = (g_dv*vn)*cn                      (reward_one's own sequence, ocd_device.h)
__device__ __forceinline__ void row_fwd_xy_phi0_h10(float &x, float &y, float cdb, float sdb, float vn, float sn,
                                                    float cn, float tgt, float bound, float w0, float &qv, float &qth)
{
    float t, g;
#define OCD_XY(FILL) "v_add_f32_dpp %[x], %[x], %[cdb]" OCD_ROW_SHR "v_add_f32_dpp %[y], %[y], %[sdb]" OCD_ROW_SHR FILL
    asm volatile("s_nop 1\n"
                 OCD_XY("v_mul_f32 %[t], %[vn], %[sn]\n")
                 OCD_XY("v_sub_f32 %[t], %[t], %[tgt]\n")
                 OCD_XY("v_mul_f32 %[g], %[t], %[t]\n")
                 OCD_XY("v_cmp_le_f32 vcc, %[g], %[bound]\n")
                 OCD_XY("v_cndmask_b32 %[g], 0, %[w0], vcc\n")
                 OCD_XY("v_add_f32 %[g], %[g], %[g]\n")
                 OCD_XY("v_mul_f32 %[g], %[g], %[t]\n")
                 OCD_XY("v_mul_f32 %[qv], %[g], %[sn]\n")
                 OCD_XY("v_mul_f32 %[g], %[g], %[vn]\n")
                 "v_mul_f32 %[qth], %[g], %[cn]\n"
                 : [x] "+&v"(x), [y] "+&v"(y), [t] "=&v"(t), [g] "=&v"(g), [qv] "=&v"(qv), [qth] "=&v"(qth)
                 : [cdb] "v"(cdb), [sdb] "v"(sdb), [vn] "v"(vn), [sn] "v"(sn), [cn] "v"(cn), [tgt] "v"(tgt),
                   [bound] "v"(bound), [w0] "v"(w0)
                 : "vcc");
#undef OCD_XY
}

// V_SEG: the y chain runs half a round behind the x chain; each chain's add and select cover the other's hazard.
// (x, y) are outputs: every lane starts at (ex, ey).
template <int HT>
__device__ __forceinline__ void seg_fwd_xy(float &x, float &y, float ex, float ey, float cd, float sd,
                                           unsigned long long first_mask)
{
    float sx, sy;                                                    // (sy = ey + sd opens the statement: no compiler-placed
#define OCD_STMT(REP)                                                 /*  instruction between it and the caller's asm) */ \
    asm volatile("v_add_f32 %[sy], %[ey], %[sd]\n"                                                        \
                 "s_mov_b64 vcc, %[m]\n"                                                                  \
                 "v_add_f32 %[sx], %[ex], %[cd]\n"                                                        \
                 "v_cndmask_b32_dpp %[y], %[sy], %[ey], vcc" OCD_WAVE_SHR                                 \
                 "v_add_f32 %[sy], %[y], %[sd]\n"                                                         \
                 "v_cndmask_b32_dpp %[x], %[sx], %[ex], vcc" OCD_WAVE_SHR                                 \
                 REP("v_add_f32 %[sx], %[x], %[cd]\n"                                                     \
                     "v_cndmask_b32_dpp %[y], %[sy], %[ey], vcc" OCD_WAVE_SHR                             \
                     "v_add_f32 %[sy], %[y], %[sd]\n"                                                     \
                     "v_cndmask_b32_dpp %[x], %[sx], %[ex], vcc" OCD_WAVE_SHR)                            \
                 : [x] "=&v"(x), [y] "=&v"(y), [sx] "=&v"(sx), [sy] "=&v"(sy)                            \
                 : [cd] "v"(cd), [sd] "v"(sd), [ex] "v"(ex), [ey] "v"(ey), [m] "s"(first_mask)           \
                 : "vcc");
    OCD_CHAIN_ROUNDS_M1(HT, OCD_STMT);
#undef OCD_STMT
}

// V_SEG, round 4: the tail of the own step's sincos (quadrant fix-up), the previous lane's (sin, cos) with the current
// heading's at a segment's first lane, the step's increments cd / sd AND the position recurrence in ONE statement
// (ocd_devmath.h: OCD_SC_TAIL).  hipcc pads every register an asm statement defines with a hazard s_nop before the
// next vector instruction reads it -- also between two adjacent statements -- so fewer, longer statements are shorter
// streams.  Same instructions as sincos_pk_seg followed by seg_fwd_xy.
template <int HT>
__device__ __forceinline__ void seg_sincos_fwd_xy(float thn, const ScConsts &k, float s0, float c0, float dd,
                                                  float ex, float ey, unsigned long long first_mask,
                                                  float &s_out, float &c_out, float &s_pre, float &c_pre,
                                                  float &sd_out, float &cd_out, float &x, float &y)
{
    float n, r, cr, sn, cn, sp, cp, sd, cd, sx, sy;
    v2f Z, P;
    sincos_pk_head<false>(thn, k, n, r, Z, P);    // scalar chains: this statement serves launches that fill the chip
    int32_t q, m;
#define OCD_STMT(REP)                                                                                     \
    asm volatile("s_mov_b64 vcc, %[fm]\n"                                                                 \
                 OCD_SC_TAIL                                                                              \
                 "v_cndmask_b32_dpp %[sp], %[sn], %[s0], vcc" OCD_WAVE_SHR                                \
                 "v_mul_f32 %[sd], %[sp], %[dd]\n"                                                        \
                 "v_cndmask_b32_dpp %[cp], %[cn], %[c0], vcc" OCD_WAVE_SHR                                \
                 "v_add_f32 %[sy], %[ey], %[sd]\n"                                                        \
                 "v_mul_f32 %[cd], %[cp], %[dd]\n"                                                        \
                 "v_add_f32 %[sx], %[ex], %[cd]\n"                                                        \
                 "v_cndmask_b32_dpp %[y], %[sy], %[ey], vcc" OCD_WAVE_SHR                                 \
                 "v_add_f32 %[sy], %[y], %[sd]\n"                                                         \
                 "v_cndmask_b32_dpp %[x], %[sx], %[ex], vcc" OCD_WAVE_SHR                                 \
                 REP("v_add_f32 %[sx], %[x], %[cd]\n"                                                     \
                     "v_cndmask_b32_dpp %[y], %[sy], %[ey], vcc" OCD_WAVE_SHR                             \
                     "v_add_f32 %[sy], %[y], %[sd]\n"                                                     \
                     "v_cndmask_b32_dpp %[x], %[sx], %[ex], vcc" OCD_WAVE_SHR)                            \
                 : [q] "=&v"(q), [m] "=&v"(m), [cr] "=&v"(cr), [sr] "+&v"(r), [sn] "=&v"(sn), [cn] "=&v"(cn),     \
                   [sp] "=&v"(sp), [cp] "=&v"(cp), [sd] "=&v"(sd), [cd] "=&v"(cd), [x] "=&v"(x), [y] "=&v"(y),    \
                   [sx] "=&v"(sx), [sy] "=&v"(sy)                                                         \
                 : [n] "v"(n), [px] "v"(P.x), [py] "v"(P.y), [zx] "v"(Z.x), [zy] "v"(Z.y), [k] "s"(0x80000000u),   \
                   [s0] "v"(s0), [c0] "v"(c0), [dd] "v"(dd), [ex] "v"(ex), [ey] "v"(ey), [fm] "s"(first_mask)     \
                 : "vcc");
    OCD_CHAIN_ROUNDS_M1(HT, OCD_STMT);
#undef OCD_STMT
    s_out = sn; c_out = cn; s_pre = sp; c_pre = cp; sd_out = sd; cd_out = cd;
}

// (Tried for V_SEG: the position recurrences without boundary selects -- a DPP operand cannot read a lane EXEC
//  disables, the reader keeps its value, so H-2 rounds with the segments' last lanes disabled plus one closing
//  step with the first lanes disabled need only one fused DPP add per coordinate and round.  8 instructions
//  fewer per pass, bit-identical, and 0.9 % SLOWER: the two EXEC writes with their 5 wait states per recurrence
//  cost more than the selects.  tools/microbench/dpp_exec.hip keeps the probe.)

// ---- adjoint position recurrence (V_SEG):  Lx <- (qx + Lx) of the lane above, 0 in the segment's top lane ----
// (Lx, Ly) are outputs: every lane starts at 0 (the first round adds the inline constant).
template <int HT>
__device__ __forceinline__ void seg_bwd_xy(float &Lx, float &Ly, float qx, float qy, unsigned long long last_mask)
{
    float ax, ay = qy + 0.0f;
    const float zero = 0.0f;
#define OCD_STMT(REP)                                                                                     \
    asm volatile("s_mov_b64 vcc, %[m]\n"                                                                  \
                 "v_add_f32 %[ax], 0, %[qx]\n"                                                            \
                 "v_cndmask_b32_dpp %[Ly], %[ay], %[z], vcc" OCD_WAVE_SHL                                 \
                 "v_add_f32 %[ay], %[qy], %[Ly]\n"                                                        \
                 "v_cndmask_b32_dpp %[Lx], %[ax], %[z], vcc" OCD_WAVE_SHL                                 \
                 REP("v_add_f32 %[ax], %[qx], %[Lx]\n"                                                    \
                     "v_cndmask_b32_dpp %[Ly], %[ay], %[z], vcc" OCD_WAVE_SHL                             \
                     "v_add_f32 %[ay], %[qy], %[Ly]\n"                                                    \
                     "v_cndmask_b32_dpp %[Lx], %[ax], %[z], vcc" OCD_WAVE_SHL)                            \
                 : [Lx] "=&v"(Lx), [Ly] "=&v"(Ly), [ax] "=&v"(ax), [ay] "+&v"(ay)                        \
                 : [qx] "v"(qx), [qy] "v"(qy), [z] "v"(zero), [m] "s"(last_mask)                         \
                 : "vcc");
    OCD_CHAIN_ROUNDS_M1(HT, OCD_STMT);
#undef OCD_STMT
}

// ---- adjoint speed / heading recurrence (V_SEG) ----
//   Av = qv + Lv ; gA = gA1 + Av*dt ; gv2 = (-gA)*fr ; gv3 = (gv2*2)*v ; Lv <- ((gv1 + Av) + gv3) of the lane above
//   Lth <- ((qth + Lth) + tau) of the lane above ; both 0 in the segment's top lane.
// The heading chain runs half a round ahead (ltd is the NEXT round's (qth + Lth) + tau).
// (Lv, Lth) are outputs: every lane starts at 0 (the first round adds the inline constant).
#define OCD_SEG_LV(LV) "v_add_f32 %[av], " LV ", %[qv]\n"        \
                       "v_add_f32 %[s], %[gv1], %[av]\n"         \
                       "v_mul_f32 %[g], %[dt], %[av]\n"          \
                       "v_add_f32 %[g], %[gA1], %[g]\n"          \
                       "v_mul_f32_e64 %[g], -%[g], %[fr]\n"      \
                       "v_add_f32 %[g], %[g], %[g]\n"            \
                       "v_mul_f32 %[g], %[g], %[v]\n"            \
                       "v_add_f32 %[s], %[s], %[g]\n"            \
                       "v_cndmask_b32_dpp %[Lth], %[ltd], %[z], vcc" OCD_WAVE_SHL \
                       "v_add_f32 %[ltd], %[qth], %[Lth]\n"      \
                       "v_add_f32 %[ltd], %[ltd], %[tau]\n"      \
                       "v_cndmask_b32_dpp %[Lv], %[s], %[z], vcc" OCD_WAVE_SHL
template <int HT>
__device__ __forceinline__ void seg_bwd_vth(float &Lv, float &Lth, float qv, float qth, float gA1, float gv1, float v,
                                            float tau, float fr, float dt, unsigned long long last_mask)
{
    float ltd = (qth + 0.0f) + tau, av, s, g;
    const float zero = 0.0f;
#define OCD_STMT(REP)                                                                                     \
    asm volatile("s_mov_b64 vcc, %[m]\n"                                                                  \
                 OCD_SEG_LV("0")                                                                          \
                 REP(OCD_SEG_LV("%[Lv]"))                                                                 \
                 : [Lv] "=&v"(Lv), [Lth] "=&v"(Lth), [ltd] "+&v"(ltd), [av] "=&v"(av), [s] "=&v"(s), [g] "=&v"(g) \
                 : [qv] "v"(qv), [qth] "v"(qth), [gA1] "v"(gA1), [gv1] "v"(gv1), [v] "v"(v), [tau] "v"(tau), \
                   [z] "v"(zero), [fr] "s"(fr), [dt] "s"(dt), [m] "s"(last_mask)                          \
                 : "vcc");
    OCD_CHAIN_ROUNDS_M1(HT, OCD_STMT);
#undef OCD_STMT
}

} // namespace ocd
