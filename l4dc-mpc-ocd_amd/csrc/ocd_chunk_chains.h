// ocd_chunk_chains.h -- hand-scheduled horizon recurrences of the chunked kernel (V_CHUNK), chunk sizes 2, 3 and 5.
//
// A recurrence round of the chunked mapping walks the lane's S steps and hands the chunk's end value to the
// neighbouring lane with the segment-boundary value selected in.  hipcc compiles the hand-over as v_mov_b32_dpp +
// v_cndmask_b32 (SGPR mask): two instructions per value and round where one v_cndmask_b32_dpp (VCC = boundary mask)
// does both -- 8 of a round's instructions, ~5 % of a pass at S = 2 / H = 15 (7 rounds).  As in ocd_chains.h each
// recurrence is ONE asm statement (hipcc pads nothing inside it), the arithmetic is the kernel's C++ formulation
// instruction for instruction, and every DPP read comes >= 2 instructions after the write of its source: the
// heading / x / Lth / Lx chains run one instruction ahead of their partners (the first add of the NEXT round sits
// between the partner's last write and its DPP read).  The first round is peeled: every lane still holds the
// boundary value there, so it reads the boundary registers (or the inline 0) directly and the recurrence registers
// are outputs only -- no copies in.
#pragma once
#include <hip/hip_runtime.h>

#include "ocd_chains.h"

#define OCD_REP6(S) OCD_REP5(S) S
#define OCD_REP7(S) OCD_REP5(S) OCD_REP2(S)
#define OCD_REP11(S) OCD_REP9(S) OCD_REP2(S)

// NR = NC - 1 rounds: the peeled one + NR - 1 repeats
#define OCD_CHUNK_ROUNDS_M1(NR, STMT)                                \
    do {                                                             \
        if constexpr ((NR) == 1) { STMT(OCD_REP0) }                  \
        else if constexpr ((NR) == 2) { STMT(OCD_REP1) }             \
        else if constexpr ((NR) == 4) { STMT(OCD_REP3) }             \
        else if constexpr ((NR) == 7) { STMT(OCD_REP6) }             \
        else if constexpr ((NR) == 8) { STMT(OCD_REP7) }             \
        else if constexpr ((NR) == 12) { STMT(OCD_REP11) }           \
        else static_assert((NR) == 1, "no repeat macro for this number of rounds"); \
    } while (0)

namespace ocd {

// the forward chains and the adjoint position chain exist for S = 2, 3, 5; the adjoint speed / heading chain for S = 2, 3
// (at S = 5 it would need 40 asm operands; the limit is 30)
template <int S, int NR> struct chunk_chain_supported {
    static constexpr bool rounds = NR == 1 || NR == 2 || NR == 4 || NR == 7 || NR == 8 || NR == 12;
    static constexpr bool value = (S == 2 || S == 3 || S == 5) && rounds;
    static constexpr bool bwd_vth = (S == 2 || S == 3) && rounds;
};

// one speed step from SRC:  v <- SRC + (a_c - fr * (SRC * SRC)) * dt
#define OCD_VSTEP(AC, SRC) "v_mul_f32 %[t], " SRC ", " SRC "\n"  \
                           "v_mul_f32 %[t], %[fr], %[t]\n"       \
                           "v_sub_f32 %[t], %[" AC "], %[t]\n"   \
                           "v_mul_f32 %[t], %[dt], %[t]\n"       \
                           "v_add_f32 %[v], " SRC ", %[t]\n"
#define OCD_V "%[v]"
#define OCD_FWD_VTH_TAIL "v_cndmask_b32_dpp %[th], %[thn], %[eth], vcc" OCD_WAVE_SHR \
                         "v_add_f32 %[thn], %[th], %[wd0]\n"                         \
                         "v_cndmask_b32_dpp %[v], %[v], %[ev], vcc" OCD_WAVE_SHR
#define OCD_FWD_VTH_2(SRC) "v_add_f32 %[thn], %[thn], %[wd1]\n" \
                           OCD_VSTEP("ac0", SRC) OCD_VSTEP("ac1", OCD_V) OCD_FWD_VTH_TAIL
#define OCD_FWD_VTH_3(SRC) "v_add_f32 %[thn], %[thn], %[wd1]\n" "v_add_f32 %[thn], %[thn], %[wd2]\n" \
                           OCD_VSTEP("ac0", SRC) OCD_VSTEP("ac1", OCD_V) OCD_VSTEP("ac2", OCD_V) OCD_FWD_VTH_TAIL
#define OCD_FWD_VTH_5(SRC) "v_add_f32 %[thn], %[thn], %[wd1]\n" "v_add_f32 %[thn], %[thn], %[wd2]\n" \
                           "v_add_f32 %[thn], %[thn], %[wd3]\n" "v_add_f32 %[thn], %[thn], %[wd4]\n" \
                           OCD_VSTEP("ac0", SRC) OCD_VSTEP("ac1", OCD_V) OCD_VSTEP("ac2", OCD_V)     \
                           OCD_VSTEP("ac3", OCD_V) OCD_VSTEP("ac4", OCD_V) OCD_FWD_VTH_TAIL

// ---- forward speed / heading: on return (v, th) are the values at the START of the lane's chunk ----
// VR: fr and dt arrive in vector registers (the builds that share a SIMD: a scalar source halves the instruction's rate)
template <int S, int NR, bool VR = false>
__device__ __forceinline__ void chunk_fwd_vth(float &v, float &th, float ev, float eth, const float (&ac)[S],
                                              const float (&wd)[S], float fr, float dt, unsigned long long first_mask)
{
    float thn = eth + wd[0], tmp;
    if constexpr (S == 2) {
#define OCD_STMTC(REP, C)                                                                                     \
        asm volatile("s_mov_b64 vcc, %[m]\n" OCD_FWD_VTH_2("%[ev]") REP(OCD_FWD_VTH_2(OCD_V))             \
                     : [v] "=&v"(v), [th] "=&v"(th), [thn] "+&v"(thn), [t] "=&v"(tmp)                    \
                     : [ac0] "v"(ac[0]), [ac1] "v"(ac[1]), [wd0] "v"(wd[0]), [wd1] "v"(wd[1]), [ev] "v"(ev), \
                       [eth] "v"(eth), [fr] C(fr), [dt] C(dt), [m] "s"(first_mask)                    \
                     : "vcc");
#define OCD_STMT(REP) OCD_STMTC(REP, "v")
        if constexpr (VR) { OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT); }
#undef OCD_STMT
#define OCD_STMT(REP) OCD_STMTC(REP, "s")
        if constexpr (!VR) { OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT); }
#undef OCD_STMT
#undef OCD_STMTC
    } else if constexpr (S == 3) {
#define OCD_STMTC(REP, C)                                                                                     \
        asm volatile("s_mov_b64 vcc, %[m]\n" OCD_FWD_VTH_3("%[ev]") REP(OCD_FWD_VTH_3(OCD_V))             \
                     : [v] "=&v"(v), [th] "=&v"(th), [thn] "+&v"(thn), [t] "=&v"(tmp)                    \
                     : [ac0] "v"(ac[0]), [ac1] "v"(ac[1]), [ac2] "v"(ac[2]), [wd0] "v"(wd[0]), [wd1] "v"(wd[1]), \
                       [wd2] "v"(wd[2]), [ev] "v"(ev), [eth] "v"(eth), [fr] C(fr), [dt] C(dt), [m] "s"(first_mask) \
                     : "vcc");
#define OCD_STMT(REP) OCD_STMTC(REP, "v")
        if constexpr (VR) { OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT); }
#undef OCD_STMT
#define OCD_STMT(REP) OCD_STMTC(REP, "s")
        if constexpr (!VR) { OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT); }
#undef OCD_STMT
#undef OCD_STMTC
    } else {
        static_assert(S == 5, "chunk sizes 2, 3, 5");
#define OCD_STMTC(REP, C)                                                                                     \
        asm volatile("s_mov_b64 vcc, %[m]\n" OCD_FWD_VTH_5("%[ev]") REP(OCD_FWD_VTH_5(OCD_V))             \
                     : [v] "=&v"(v), [th] "=&v"(th), [thn] "+&v"(thn), [t] "=&v"(tmp)                    \
                     : [ac0] "v"(ac[0]), [ac1] "v"(ac[1]), [ac2] "v"(ac[2]), [ac3] "v"(ac[3]), [ac4] "v"(ac[4]), \
                       [wd0] "v"(wd[0]), [wd1] "v"(wd[1]), [wd2] "v"(wd[2]), [wd3] "v"(wd[3]), [wd4] "v"(wd[4]), \
                       [ev] "v"(ev), [eth] "v"(eth), [fr] C(fr), [dt] C(dt), [m] "s"(first_mask)      \
                     : "vcc");
#define OCD_STMT(REP) OCD_STMTC(REP, "v")
        if constexpr (VR) { OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT); }
#undef OCD_STMT
#define OCD_STMT(REP) OCD_STMTC(REP, "s")
        if constexpr (!VR) { OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT); }
#undef OCD_STMT
#undef OCD_STMTC
    }
}

// ---- forward position: on return (x, y) are the position at the start of the lane's chunk ----
//   a round: xn (one add ahead) takes the rest of the x walk, y its whole walk from YSRC, then the two hand-overs
#define OCD_FWD_XY_TAIL "v_cndmask_b32_dpp %[x], %[xn], %[ex], vcc" OCD_WAVE_SHR \
                        "v_add_f32 %[xn], %[x], %[cd0]\n"                        \
                        "v_cndmask_b32_dpp %[y], %[y], %[ey], vcc" OCD_WAVE_SHR
#define OCD_FWD_XY_2(YSRC) "v_add_f32 %[xn], %[xn], %[cd1]\n"            \
                           "v_add_f32 %[y], " YSRC ", %[sd0]\n" "v_add_f32 %[y], %[y], %[sd1]\n" OCD_FWD_XY_TAIL
#define OCD_FWD_XY_3(YSRC) "v_add_f32 %[xn], %[xn], %[cd1]\n" "v_add_f32 %[xn], %[xn], %[cd2]\n" \
                           "v_add_f32 %[y], " YSRC ", %[sd0]\n" "v_add_f32 %[y], %[y], %[sd1]\n" \
                           "v_add_f32 %[y], %[y], %[sd2]\n" OCD_FWD_XY_TAIL
#define OCD_FWD_XY_5(YSRC) "v_add_f32 %[xn], %[xn], %[cd1]\n" "v_add_f32 %[xn], %[xn], %[cd2]\n" \
                           "v_add_f32 %[xn], %[xn], %[cd3]\n" "v_add_f32 %[xn], %[xn], %[cd4]\n" \
                           "v_add_f32 %[y], " YSRC ", %[sd0]\n" "v_add_f32 %[y], %[y], %[sd1]\n" \
                           "v_add_f32 %[y], %[y], %[sd2]\n" "v_add_f32 %[y], %[y], %[sd3]\n"     \
                           "v_add_f32 %[y], %[y], %[sd4]\n" OCD_FWD_XY_TAIL
template <int S, int NR>
__device__ __forceinline__ void chunk_fwd_xy(float &x, float &y, float ex, float ey, const float (&cd)[S],
                                             const float (&sd)[S], unsigned long long first_mask)
{
    float xn = ex + cd[0];
    if constexpr (S == 2) {
#define OCD_STMT(REP)                                                                                     \
        asm volatile("s_mov_b64 vcc, %[m]\n" OCD_FWD_XY_2("%[ey]") REP(OCD_FWD_XY_2("%[y]"))              \
                     : [x] "=&v"(x), [y] "=&v"(y), [xn] "+&v"(xn)                                        \
                     : [cd0] "v"(cd[0]), [cd1] "v"(cd[1]), [sd0] "v"(sd[0]), [sd1] "v"(sd[1]), [ex] "v"(ex), \
                       [ey] "v"(ey), [m] "s"(first_mask)                                                  \
                     : "vcc");
        OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT);
#undef OCD_STMT
    } else if constexpr (S == 3) {
#define OCD_STMT(REP)                                                                                     \
        asm volatile("s_mov_b64 vcc, %[m]\n" OCD_FWD_XY_3("%[ey]") REP(OCD_FWD_XY_3("%[y]"))              \
                     : [x] "=&v"(x), [y] "=&v"(y), [xn] "+&v"(xn)                                        \
                     : [cd0] "v"(cd[0]), [cd1] "v"(cd[1]), [cd2] "v"(cd[2]), [sd0] "v"(sd[0]), [sd1] "v"(sd[1]), \
                       [sd2] "v"(sd[2]), [ex] "v"(ex), [ey] "v"(ey), [m] "s"(first_mask)                  \
                     : "vcc");
        OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT);
#undef OCD_STMT
    } else {
        static_assert(S == 5, "chunk sizes 2, 3, 5");
#define OCD_STMT(REP)                                                                                     \
        asm volatile("s_mov_b64 vcc, %[m]\n" OCD_FWD_XY_5("%[ey]") REP(OCD_FWD_XY_5("%[y]"))              \
                     : [x] "=&v"(x), [y] "=&v"(y), [xn] "+&v"(xn)                                        \
                     : [cd0] "v"(cd[0]), [cd1] "v"(cd[1]), [cd2] "v"(cd[2]), [cd3] "v"(cd[3]), [cd4] "v"(cd[4]), \
                       [sd0] "v"(sd[0]), [sd1] "v"(sd[1]), [sd2] "v"(sd[2]), [sd3] "v"(sd[3]), [sd4] "v"(sd[4]), \
                       [ex] "v"(ex), [ey] "v"(ey), [m] "s"(first_mask)                                    \
                     : "vcc");
        OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT);
#undef OCD_STMT
    }
}

// ---- adjoint position: on return (Lx, Ly) are the adjoint arriving at the END of the lane's chunk ----
//   per round  L <- q[S-1] + L, ..., L <- q[0] + L  (the chunk walked backwards), handed to the lane below; 0 in the top
//   lane.  YSRC = the value Ly starts the round with ("0" in the peeled round).
#define OCD_BWD_XY_TAIL(QXT) "v_cndmask_b32_dpp %[Lx], %[xn], %[z], vcc" OCD_WAVE_SHL \
                             "v_add_f32 %[xn], %[" QXT "], %[Lx]\n"                  \
                             "v_cndmask_b32_dpp %[Ly], %[Ly], %[z], vcc" OCD_WAVE_SHL
#define OCD_BWD_XY_2(YSRC) "v_add_f32 %[xn], %[qx0], %[xn]\n"                     \
                           "v_add_f32 %[Ly], " YSRC ", %[qy1]\n" "v_add_f32 %[Ly], %[qy0], %[Ly]\n" OCD_BWD_XY_TAIL("qx1")
#define OCD_BWD_XY_3(YSRC) "v_add_f32 %[xn], %[qx1], %[xn]\n" "v_add_f32 %[xn], %[qx0], %[xn]\n" \
                           "v_add_f32 %[Ly], " YSRC ", %[qy2]\n" "v_add_f32 %[Ly], %[qy1], %[Ly]\n" \
                           "v_add_f32 %[Ly], %[qy0], %[Ly]\n" OCD_BWD_XY_TAIL("qx2")
#define OCD_BWD_XY_5(YSRC) "v_add_f32 %[xn], %[qx3], %[xn]\n" "v_add_f32 %[xn], %[qx2], %[xn]\n" \
                           "v_add_f32 %[xn], %[qx1], %[xn]\n" "v_add_f32 %[xn], %[qx0], %[xn]\n" \
                           "v_add_f32 %[Ly], " YSRC ", %[qy4]\n" "v_add_f32 %[Ly], %[qy3], %[Ly]\n" \
                           "v_add_f32 %[Ly], %[qy2], %[Ly]\n" "v_add_f32 %[Ly], %[qy1], %[Ly]\n" \
                           "v_add_f32 %[Ly], %[qy0], %[Ly]\n" OCD_BWD_XY_TAIL("qx4")
template <int S, int NR>
__device__ __forceinline__ void chunk_bwd_xy(float &Lx, float &Ly, const float (&qx)[S], const float (&qy)[S],
                                             unsigned long long last_mask)
{
    float xn = qx[S - 1] + 0.0f;
    const float zero = 0.0f;
    if constexpr (S == 2) {
#define OCD_STMT(REP)                                                                                     \
        asm volatile("s_mov_b64 vcc, %[m]\n" OCD_BWD_XY_2("0") REP(OCD_BWD_XY_2("%[Ly]"))                 \
                     : [Lx] "=&v"(Lx), [Ly] "=&v"(Ly), [xn] "+&v"(xn)                                    \
                     : [qx0] "v"(qx[0]), [qx1] "v"(qx[1]), [qy0] "v"(qy[0]), [qy1] "v"(qy[1]), [z] "v"(zero), \
                       [m] "s"(last_mask)                                                                 \
                     : "vcc");
        OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT);
#undef OCD_STMT
    } else if constexpr (S == 3) {
#define OCD_STMT(REP)                                                                                     \
        asm volatile("s_mov_b64 vcc, %[m]\n" OCD_BWD_XY_3("0") REP(OCD_BWD_XY_3("%[Ly]"))                 \
                     : [Lx] "=&v"(Lx), [Ly] "=&v"(Ly), [xn] "+&v"(xn)                                    \
                     : [qx0] "v"(qx[0]), [qx1] "v"(qx[1]), [qx2] "v"(qx[2]), [qy0] "v"(qy[0]), [qy1] "v"(qy[1]), \
                       [qy2] "v"(qy[2]), [z] "v"(zero), [m] "s"(last_mask)                                \
                     : "vcc");
        OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT);
#undef OCD_STMT
    } else {
        static_assert(S == 5, "chunk sizes 2, 3, 5");
#define OCD_STMT(REP)                                                                                     \
        asm volatile("s_mov_b64 vcc, %[m]\n" OCD_BWD_XY_5("0") REP(OCD_BWD_XY_5("%[Ly]"))                 \
                     : [Lx] "=&v"(Lx), [Ly] "=&v"(Ly), [xn] "+&v"(xn)                                    \
                     : [qx0] "v"(qx[0]), [qx1] "v"(qx[1]), [qx2] "v"(qx[2]), [qx3] "v"(qx[3]), [qx4] "v"(qx[4]), \
                       [qy0] "v"(qy[0]), [qy1] "v"(qy[1]), [qy2] "v"(qy[2]), [qy3] "v"(qy[3]), [qy4] "v"(qy[4]), \
                       [z] "v"(zero), [m] "s"(last_mask)                                                  \
                     : "vcc");
        OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT);
#undef OCD_STMT
    }
}

// one adjoint speed step (step index I) from LV:  Av = qv + LV ; gA = gA1 + Av*dt ; gv2 = (-gA)*fr ;
//                                                 gv3 = (gv2*2)*vpre ; Lv = (gv1 + Av) + gv3
#define OCD_LVSTEP(I, LV) "v_add_f32 %[av], " LV ", %[qv" I "]\n"   \
                          "v_add_f32 %[s], %[gv1" I "], %[av]\n"    \
                          "v_mul_f32 %[g], %[dt], %[av]\n"           \
                          "v_add_f32 %[g], %[gA1" I "], %[g]\n"     \
                          "v_mul_f32_e64 %[g], -%[g], %[fr]\n"       \
                          "v_add_f32 %[g], %[g], %[g]\n"             \
                          "v_mul_f32 %[g], %[g], %[vp" I "]\n"      \
                          "v_add_f32 %[Lv], %[s], %[g]\n"
#define OCD_LV "%[Lv]"
// one adjoint heading step:  Lth = (qth + Lth) + tau   (on the running value %[ltd])
#define OCD_LTHSTEP(I) "v_add_f32 %[ltd], %[qth" I "], %[ltd]\n" \
                       "v_add_f32 %[ltd], %[ltd], %[tau" I "]\n"
#define OCD_BWD_VTH_TAIL(T) "v_cndmask_b32_dpp %[Lth], %[ltd], %[z], vcc" OCD_WAVE_SHL \
                            "v_add_f32 %[ltd], %[qth" T "], %[Lth]\n"                  \
                            "v_add_f32 %[ltd], %[ltd], %[tau" T "]\n"                  \
                            "v_cndmask_b32_dpp %[Lv], %[Lv], %[z], vcc" OCD_WAVE_SHL
#define OCD_BWD_VTH_2(LV) OCD_LTHSTEP("0") OCD_LVSTEP("1", LV) OCD_LVSTEP("0", OCD_LV) OCD_BWD_VTH_TAIL("1")
#define OCD_BWD_VTH_3(LV) OCD_LTHSTEP("1") OCD_LTHSTEP("0") \
                          OCD_LVSTEP("2", LV) OCD_LVSTEP("1", OCD_LV) OCD_LVSTEP("0", OCD_LV) OCD_BWD_VTH_TAIL("2")

// ---- adjoint speed / heading: on return (Lv, Lth) are the adjoint arriving at the END of the lane's chunk ----
template <int S, int NR, bool VR = false>
__device__ __forceinline__ void chunk_bwd_vth(float &Lv, float &Lth, const float (&qv)[S], const float (&qth)[S],
                                              const float (&gA1)[S], const float (&gv1)[S], const float (&vp)[S],
                                              const float (&tau)[S], float fr, float dt, unsigned long long last_mask)
{
    float ltd = (qth[S - 1] + 0.0f) + tau[S - 1], av, s, g;
    const float zero = 0.0f;
    if constexpr (S == 2) {
#define OCD_STMTC(REP, C)                                                                                     \
        asm volatile("s_mov_b64 vcc, %[m]\n" OCD_BWD_VTH_2("0") REP(OCD_BWD_VTH_2(OCD_LV))                \
                     : [Lv] "=&v"(Lv), [Lth] "=&v"(Lth), [ltd] "+&v"(ltd), [av] "=&v"(av), [s] "=&v"(s), [g] "=&v"(g) \
                     : [qv0] "v"(qv[0]), [qv1] "v"(qv[1]), [qth0] "v"(qth[0]), [qth1] "v"(qth[1]),        \
                       [gA10] "v"(gA1[0]), [gA11] "v"(gA1[1]), [gv10] "v"(gv1[0]), [gv11] "v"(gv1[1]),    \
                       [vp0] "v"(vp[0]), [vp1] "v"(vp[1]), [tau0] "v"(tau[0]), [tau1] "v"(tau[1]),        \
                       [z] "v"(zero), [fr] C(fr), [dt] C(dt), [m] "s"(last_mask)                      \
                     : "vcc");
#define OCD_STMT(REP) OCD_STMTC(REP, "v")
        if constexpr (VR) { OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT); }
#undef OCD_STMT
#define OCD_STMT(REP) OCD_STMTC(REP, "s")
        if constexpr (!VR) { OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT); }
#undef OCD_STMT
#undef OCD_STMTC
    } else {
        static_assert(S == 3, "adjoint speed / heading chain: chunk sizes 2 and 3");
#define OCD_STMTC(REP, C)                                                                                     \
        asm volatile("s_mov_b64 vcc, %[m]\n" OCD_BWD_VTH_3("0") REP(OCD_BWD_VTH_3(OCD_LV))                \
                     : [Lv] "=&v"(Lv), [Lth] "=&v"(Lth), [ltd] "+&v"(ltd), [av] "=&v"(av), [s] "=&v"(s), [g] "=&v"(g) \
                     : [qv0] "v"(qv[0]), [qv1] "v"(qv[1]), [qv2] "v"(qv[2]), [qth0] "v"(qth[0]), [qth1] "v"(qth[1]), \
                       [qth2] "v"(qth[2]), [gA10] "v"(gA1[0]), [gA11] "v"(gA1[1]), [gA12] "v"(gA1[2]),    \
                       [gv10] "v"(gv1[0]), [gv11] "v"(gv1[1]), [gv12] "v"(gv1[2]), [vp0] "v"(vp[0]), [vp1] "v"(vp[1]), \
                       [vp2] "v"(vp[2]), [tau0] "v"(tau[0]), [tau1] "v"(tau[1]), [tau2] "v"(tau[2]),      \
                       [z] "v"(zero), [fr] C(fr), [dt] C(dt), [m] "s"(last_mask)                      \
                     : "vcc");
#define OCD_STMT(REP) OCD_STMTC(REP, "v")
        if constexpr (VR) { OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT); }
#undef OCD_STMT
#define OCD_STMT(REP) OCD_STMTC(REP, "s")
        if constexpr (!VR) { OCD_CHUNK_ROUNDS_M1(NR, OCD_STMT); }
#undef OCD_STMT
#undef OCD_STMTC
    }
}

} // namespace ocd
