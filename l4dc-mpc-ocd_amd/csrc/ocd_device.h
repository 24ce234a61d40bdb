// ocd_device.h -- device primitives shared by the planner kernels: the reference's math helpers, the
// car dynamics step, the reward features with their reverse-mode adjoint, the terminal-value lookup.
//
// Reference (file:line relative to the reference tree):
//   _f / smooth_threshold / smooth_bump       interact_drive/math_utils.py:7-31,59-97,135-180
//   car_dynamics_step                         interact_drive/simulation_utils.py:9-21
//   ThreeLaneTestCar.features                 experiments/merging.py:32-83
//   LinearRewardCar.reward_fn                 interact_drive/car/linear_reward_car.py:49-55
//   ValueFeature.interpolate_value            interact_drive/reward_design/value_interpolation.py:28-61
//
// Numerics: IEEE binary32, one rounding per TensorFlow op of the reference, -ffp-contract=off; the
// operation order is the arithmetic contract of DESIGN.md section 3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocd.h"
#include "ocd_devmath.h"

namespace ocd {

// ---------------------------------------------------------------- primitives
__device__ __forceinline__ float min_tf(float a, float b) { return (a <= b) ? a : b; }
__device__ __forceinline__ float max_tf(float a, float b) { return (a >= b) ? a : b; }

// _f (math_utils.py:28-31)
struct FTape { bool pos; float m, e, u; };

__device__ __forceinline__ float f_fwd(float t, float shape, FTape &tp)
{
    const bool pos = t > 0.0f;
    const float tc = pos ? t : (0.0f + 0.01f);
    const float u = shape * tc;
    const float m = -1.0f / u;
    const float e = exp_le1(m);
    tp.pos = pos; tp.m = m; tp.e = e; tp.u = u;
    return pos ? e : 0.0f;
}

__device__ __forceinline__ float f_bwd(float g, float shape, const FTape &tp, float k /* (-m)/u */)
{
    const float g_e = tp.pos ? g : 0.0f;
    const float g_m = g_e * tp.e;
    const float g_u = g_m * k;
    const float g_tc = g_u * shape;
    return tp.pos ? g_tc : 0.0f;
}

// f_bwd without its first select: where pos is false the result is 0 through the LAST select whatever the
// product chain computed, where it is true both selects pass the value -- same result, one select fewer
__device__ __forceinline__ float f_bwd_gated(float g, float shape, const FTape &tp, float k)
{
    const float g_m = g * tp.e;
    const float g_u = g_m * k;
    const float g_tc = g_u * shape;
    return tp.pos ? g_tc : 0.0f;
}

// smooth_threshold (math_utils.py:87-95)
struct ThrTape { FTape t1, t2; float den, S; };

__device__ __forceinline__ float thr_fwd(float z, float lo, float width, float shape, ThrTape &tp)
{
    const float xd = z - lo;
    const float F1 = f_fwd(xd, shape, tp.t1);
    const float xd2 = width - xd;
    const float F2 = f_fwd(xd2, shape, tp.t2);
    const float den = F1 + F2;
    const float S = F1 / den;
    tp.den = den; tp.S = S;
    return S;
}

__device__ __forceinline__ float thr_bwd(float g_S, float shape, const ThrTape &tp)
{
    const float g_F1a = g_S / tp.den;
    const float g_den = g_S * ((-tp.S) / tp.den);
    const float k1 = (-tp.t1.m) / tp.t1.u;      // shared by the two _f(x_diff) call sites
    const float k2 = (-tp.t2.m) / tp.t2.u;
    const float ga = f_bwd(g_F1a, shape, tp.t1, k1);
    const float gb = f_bwd(g_den, shape, tp.t1, k1);
    const float gc = f_bwd(g_den, shape, tp.t2, k2);
    return (ga + gb) + (-gc);
}

// smooth_bump (math_utils.py:166-178); center/width precomputed per control step
struct BumpTape { bool cond; float xc, q, m, e; };

__device__ __forceinline__ float bump_fwd(float z, float center, float width, BumpTape &tp)
{
    const float zn = (z - center) / width;
    const bool cond = (zn * zn) < 1.0f;
    const float xc = cond ? zn : 0.0f;
    const float q = 1.0f - xc * xc;
    const float m = -1.0f / q;
    const float arg = m + 1.0f;
    const float e = exp_le1(arg);
    tp.cond = cond; tp.xc = xc; tp.q = q; tp.m = m; tp.e = e;
    return cond ? e : 0.0f;
}

__device__ __forceinline__ float bump_bwd(float g, float width, const BumpTape &tp)
{
    const float g_e = tp.cond ? g : 0.0f;
    const float g_arg = g_e * tp.e;
    const float g_q = g_arg * ((-tp.m) / tp.q);
    const float g_xc2 = -g_q;
    const float g_xc = (g_xc2 * 2.0f) * tp.xc;
    const float g_zn = tp.cond ? g_xc : 0.0f;
    return g_zn / width;
}

// car_dynamics_step (simulation_utils.py:9-21) on explicit cos/sin of the heading
__device__ __forceinline__ void dyn_step(float x, float y, float v, float th, float c, float s,
                                         float a, float w, float dt, float dt2, float f,
                                         float &xn, float &yn, float &vn, float &thn)
{
    const float a_c = max_tf(min_tf(a, 4.0f), -8.0f);
    const float w_c = max_tf(min_tf(w, 4.0f), -4.0f);
    const float v2 = v * v;
    const float fv2 = f * v2;
    const float acc = a_c - fv2;
    const float vdt = v * dt;
    const float hA = 0.5f * acc;
    const float hAdt2 = hA * dt2;
    const float d = vdt + hAdt2;
    xn = x + c * d;
    yn = y + s * d;
    vn = v + acc * dt;
    thn = th + w_c * dt;
}

// bump centre / half-width of a scripted car at (ox, oy): smooth_bump(o - h, o + h)
// (merging.py:72-73, math_utils.py:167-168)
struct BumpGeom { float cx, wx, cy, wy; };

__device__ __forceinline__ BumpGeom bump_geom(float ox, float oy, float hx, float hy)
{
    BumpGeom g;
    const float sx = ox - hx, ex = ox + hx;
    g.wx = (ex - sx) / 2.0f;
    g.cx = (sx + ex) / 2.0f;
    const float sy = oy - hy, ey = oy + hy;
    g.wy = (ey - sy) / 2.0f;
    g.cy = (sy + ey) / 2.0f;
    return g;
}

// refined reciprocals of a BumpGeom's half-widths (ocd_devmath.h: refined_recip), once per control step: reward_one's
// FASTDIV form divides by them
struct BumpRecip { float rx, ry; };

// the half-widths a FASTDIV pass may divide by through quot2_by_recip: [2^-20, 2^20] (false for NaN)
__device__ __forceinline__ bool bump_widths_guarded(const BumpGeom &g)
{
    const float lo = 9.5367431640625e-7f, hi = 1048576.0f;
    return g.wx >= lo && g.wx <= hi && g.wy >= lo && g.wy <= hi;
}

// Skipping a car's collision feature on a lane outside its box rests on the skipped adjoint g_zn / width being exactly
// +-0 (g_zn = 0 there): true for a finite, non-zero width.  A half-width that rounds away against the car's position
// ((o + h) - (o - h) = 0) or a non-finite position makes it 0/0 or 0/NaN = NaN in the full evaluation: such a control step
// evaluates every feature of every lane.
__device__ __forceinline__ bool bump_widths_degenerate(const BumpGeom &g)
{
    return !(g.wx > 0.0f && g.wx <= 3.4028234663852886e38f && g.wy > 0.0f && g.wy <= 3.4028234663852886e38f);
}

struct Q4 { float qx, qy, qv, qth; };

// The kernels' template parameter L: > 0 = lane-feature reward with L lanes, 0 = target-speed test reward,
// -1 = linear target-speed reward.  Number of features / weights of a reward:
__host__ __device__ constexpr int feat_dim(int L) { return L > 0 ? L + 4 : (L < 0 ? 2 : 0); }

// 1.0f / (float)n for a tie count n in [1, 4]: the correctly rounded quotients as constants
__device__ __forceinline__ float inv_count(int n)
{
    float r = 1.0f;
    r = (n == 2) ? 0.5f : r;
    r = (n == 3) ? (1.0f / 3.0f) : r;
    r = (n == 4) ? 0.25f : r;
    return r;
}

// Conservative per-lane tests for the wave-uniform feature skips.
//  fence:  _f(x - lo) and _f(-x - lo) are both 0 (value AND gradient, math_utils.py:28-31) unless one
//          argument is > 0, i.e. unless |x| > lo (a - b > 0 <=> a > b in IEEE arithmetic with gradual
//          underflow); then S = 0/den = 0, the feature is 0*|x| and every adjoint term is +-0.
//  collision: bump_x*bump_y and its gradients are +-0 unless x_norm^2 < 1 AND y_norm^2 < 1
//          (math_utils.py:171-178); |z - c| < 1.001*w is a cheap superset of ((z-c)/w)^2 < 1.
__device__ __forceinline__ bool needs_fence(const ocd_scenario_desc &d, float x)
{
    return __builtin_fabsf(x) > d.fence_lo;
}

__device__ __forceinline__ bool needs_collision1(float x, float y, const BumpGeom &g)
{
    const float dx = x - g.cx, dy = y - g.cy;
    return (__builtin_fabsf(dx) < g.wx * 1.001f) && (__builtin_fabsf(dy) < g.wy * 1.001f);
}

// ---------------------------------------------------------------- reward, every feature evaluated
// reward of one world state and (GRAD) its gradient w.r.t. the ego state
// (merging.py:44-83, linear_reward_car.py:49-55, targetSpeedRewardMaximizerCar.py:50-56).
// do_col / do_fence are WAVE-UNIFORM: false only when the caller has proved that, for every live lane,
// the collision bumps / the fence thresholds are identically zero together with their gradients (see
// needs_collision1 / needs_fence), so skipping them changes no bit of any result.
// SCORED (value only): car.reward_fn(past_state, ...) as an episode is scored (mpc_ord.py:99) and as ocd_reward_batch returns
// it -- the lane offset carries StraightLane.dist2median's y-term (y - p[1]) * n[1], n[1] = 0 (world.py:216-217): +-0 for a
// finite y, NaN beyond; the planner's objective keeps (x - p[0]) * -1 (include/ocd.h ABI 3).
template <int NO, int L, bool GRAD, bool SCORED = false>
__device__ __forceinline__ float reward_state(const ocd_scenario_desc &d, const float (&w)[OCD_MAX_FEATURES],
                                              float x, float y, float v, float sn, float cn,
                                              const BumpGeom (&bg)[NO > 0 ? NO : 1], Q4 &q,
                                              float *feats /* nullptr or [D] global */,
                                              const bool do_col = true, const bool do_fence = true,
                                              const bool two_sided = false)
{
    if (L == 0) {                                  // OCD_REWARD_TARGET_SPEED (the planner KAT car)
        const float dv = v - d.target_speed;
        const float sq = dv * dv;
        if (GRAD) { q.qx = 0.0f; q.qy = 0.0f; q.qth = 0.0f; q.qv = (-1.0f * 2.0f) * dv; }
        return 0.0f - sq;
    }
    if (L < 0) {                                   // OCD_REWARD_LINEAR_TARGET_SPEED: w . [v, (v - target)^2]
        const float dv = v - d.target_speed;       // (linearTargetSpeedPlannerCar.py:36-44)
        const float sq = dv * dv;
        float r = w[0] * v;
        r = r + w[1] * sq;
        if (feats) { feats[0] = v; feats[1] = sq; }
        if (GRAD) { q.qx = 0.0f; q.qy = 0.0f; q.qth = 0.0f; q.qv = w[0] + (w[1] * 2.0f) * dv; }
        return r;
    }
    constexpr int NOA = NO > 0 ? NO : 1;
    const float tgt = d.target_speed;
    const float bound = 4.0f * (tgt * tgt);
    const float vel = v * sn;
    const float dv = vel - tgt;
    const float sq = dv * dv;
    const bool pass0 = sq <= bound;
    const float phi0 = min_tf(sq, bound);

    constexpr int LA = L > 0 ? L : 1;
    float rl[LA], pl[LA];
    float pmin = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float diff = x - d.lane_center[l];
        rl[l] = diff * -1.0f;
        if (SCORED) rl[l] = rl[l] + (y - d.lane_origin_y) * d.lane_normal_y;
        const float d2 = rl[l] * rl[l];
        pl[l] = d2 * 10.0f;
        pmin = (l == 0) ? pl[0] : min_tf(pmin, pl[l]);
    }
    static_assert(!(SCORED && GRAD), "the scored form has no gradient");
    int ntie_min = 0;
#pragma unroll
    for (int l = 0; l < L; ++l) ntie_min += (pl[l] == pmin) ? 1 : 0;

    BumpTape bx[NOA], by[NOA];
    float bxv[NOA], byv[NOA], col[NOA];
    float pcol = 0.0f;
    int ntie_col = NO;
    ThrTape tp_f, tp_m;
    const bool side_p = x > d.fence_lo;
    float Ssum = 0.0f, ax = 0.0f, pf = 0.0f;
    if (do_col) {
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            bxv[j] = bump_fwd(x, bg[j].cx, bg[j].wx, bx[j]);
            byv[j] = bump_fwd(y, bg[j].cy, bg[j].wy, by[j]);
            col[j] = bxv[j] * byv[j];
            pcol = (j == 0) ? col[0] : max_tf(pcol, col[j]);
        }
        ntie_col = 0;
#pragma unroll
        for (int j = 0; j < NO; ++j) ntie_col += (col[j] == pcol) ? 1 : 0;
    }
    // fences = (S(x) + S(-x)) * |x| (merging.py:80-81).  With threshold - width = fence_lo >= 0 the two
    // arguments x - lo and -x - lo cannot both be positive, and a side whose argument is <= 0 has
    // F1 = 0 exactly: S = 0/den = 0 and every adjoint term of that side is +-0 (see needs_fence).  So
    // one smooth_threshold evaluation on the possibly-active side gives S(x) + S(-x) and its gradient
    // bit for bit (x + 0 = x), at half the divisions and exponentials.
    // two_sided (wave-uniform; the generic kernels only, for descriptors with fence_shape * fence_width < 1/80): the
    // argument above needs F2 = exp(-1/(shape * (width - x_diff))) of the inactive side to be non-zero, i.e.
    // shape * width >= 1/87.  Below that smooth_threshold is 0/0 = NaN on a band of the ROAD in the reference itself
    // (merging.py:80-81 evaluates both sides), and both sides are evaluated here as the reference writes them.
    if (do_fence) {
        if (two_sided) {
            const float Sp = thr_fwd(x, d.fence_lo, d.fence_width, d.fence_shape, tp_f);
            const float Sm = thr_fwd(-x, d.fence_lo, d.fence_width, d.fence_shape, tp_m);
            Ssum = Sp + Sm;
        } else {
            Ssum = thr_fwd(side_p ? x : -x, d.fence_lo, d.fence_width, d.fence_shape, tp_f);
        }
        ax = (x < 0.0f) ? -x : x;
        pf = Ssum * ax;
    }

    // reduce_sum(weights * feats), left to right over [phi0, lanes..., min, collision, fences]
    float r = w[0] * phi0;
#pragma unroll
    for (int l = 0; l < L; ++l) r = r + w[1 + l] * pl[l];
    const float w_min = w[L + 1], w_col = w[L + 2], w_f = w[L + 3];
    r = r + w_min * pmin;
    if (do_col) r = r + w_col * pcol;              // skipped terms are exactly +-0
    if (do_fence) r = r + w_f * pf;
    if (feats) {
        feats[0] = phi0;
#pragma unroll
        for (int l = 0; l < L; ++l) feats[1 + l] = pl[l];
        feats[L + 1] = pmin; feats[L + 2] = pcol; feats[L + 3] = pf;
    }
    if (!GRAD) return r;

    const float g_sq = pass0 ? w[0] : 0.0f;
    const float g_dv = (g_sq * 2.0f) * dv;
    q.qv = g_dv * sn;
    const float g_sn = g_dv * v;
    q.qth = g_sn * cn;

    float qx = 0.0f, qy = 0.0f;
    const float min_share = inv_count(ntie_min) * w_min;          // (indicator / num_ties) * grad
#pragma unroll
    for (int l = 0; l < L; ++l) {
        float g = w[1 + l];
        g = (pl[l] == pmin) ? (g + min_share) : g;
        const float g_d2 = g * 10.0f;
        const float g_r = (g_d2 * 2.0f) * rl[l];
        qx = qx + g_r * -1.0f;
    }
    if (NO > 0 && do_col) {
        const float col_share = inv_count(ntie_col) * w_col;
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            const float share = (col[j] == pcol) ? col_share : 0.0f;
            const float g_bx = share * byv[j];
            const float g_by = share * bxv[j];
            qx = qx + bump_bwd(g_bx, bg[j].wx, bx[j]);
            qy = qy + bump_bwd(g_by, bg[j].wy, by[j]);
        }
    }
    if (do_fence) {
        const float g_Ssum = w_f * ax;
        const float g_ax = w_f * Ssum;
        if (two_sided) {
            qx = qx + thr_bwd(g_Ssum, d.fence_shape, tp_f);
            qx = qx + (-thr_bwd(g_Ssum, d.fence_shape, tp_m));
        } else {
            const float g_z = thr_bwd(g_Ssum, d.fence_shape, tp_f);
            qx = qx + (side_p ? g_z : -g_z);
        }
        const float sgn = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
        qx = qx + g_ax * sgn;
    }
    q.qx = qx; q.qy = qy;
    return r;
}

// Lane-feature gradient factors of one trajectory, precomputed once per kernel (lane_grad_const below): per lane l the
// chain  g = w_l (+ w_min when lane l is the sole minimum) ; g * 10 ; * 2  of the backward pass, and x_hi, the guard of the
// shortened reciprocals.
template <int L>
struct LaneGradConst { float g0[L > 0 ? L : 1], g1[L > 0 ? L : 1]; float x_hi; };

// dr/dx through the lane features (merging.py:55-66: distances to the lane medians and their reduce_min) for the
// multi-feature evaluations, in the form reward_one uses: the per-lane factors come precomputed (LaneGradConst), ties of the
// minimum are detected on the lane masks of the comparisons (scalar and / or, free beside the vector stream), and only a
// pass in which some live lane HAS a tie recomputes the chain with the tie count -- bit for bit the plain chain
//   gl = w_l (+ (1 / ntie) w_min on the minimum) ; (gl * 10) * 2 * r_l ; qx += . * -1.
template <int L>
__device__ __forceinline__ float lane_grad_qx(const float (&w)[OCD_MAX_FEATURES], const float (&pl)[L], const float (&rl)[L],
                                              float pmin, const LaneGradConst<L> &lgc, unsigned long long live_mask)
{
    bool tie[L];
    unsigned long long tie_two = 0ull, tie_seen = 0ull;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        tie[l] = pl[l] == pmin;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(tie[l]);
        tie_two |= tie_seen & m;
        tie_seen |= m;
    }
    float qx = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float g_r = (tie[l] ? lgc.g1[l] : lgc.g0[l]) * rl[l];
        qx = qx + g_r * -1.0f;
    }
    if (__builtin_expect((tie_two & live_mask) != 0ull, 0)) {
        float pm = pmin;
        asm volatile("" : "+v"(pm));               // (the count stays in this branch)
        int ntie_min = 0;
#pragma unroll
        for (int l = 0; l < L; ++l) ntie_min += (pl[l] == pm) ? 1 : 0;
        const float min_share = inv_count(ntie_min) * w[L + 1];
        qx = 0.0f;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            float gl = w[1 + l];
            gl = tie[l] ? (gl + min_share) : gl;
            const float g_d2 = gl * 10.0f;
            const float g_r = (g_d2 * 2.0f) * rl[l];
            qx = qx + g_r * -1.0f;
        }
    }
    return qx;
}

// ---------------------------------------------------------------- reward, fence + at most ONE scripted car per lane
// Precondition (WAVE-UNIFORM, proved by the caller with needs_collision1): no live lane is inside the collision box
// of MORE THAN ONE scripted car; nc[j] marks the lanes inside car j's box.  The fence is evaluated for every lane
// (it is exactly 0 with +-0 adjoints where needs_fence fails), the collision term for the lane's one car only: the
// cars a lane is not near have col == 0 exactly with +-0 adjoints, so the evaluated car is the reduce_max, tied with
// all others iff its own product is 0 (as in reward_one).  Same operations on the same values as reward_state, sums
// that only skip +-0 terms: the full evaluation's result bit for bit, at 15 divisions and 4 exponentials per
// lane-step instead of 17 + 6 per scripted car.  This is the common multi-feature case of the scenarios whose
// fence region overlaps a car's collision box (replanning: x in (0.05, 0.08); merging: x in (0.1, 0.18)).
// FASTDIV (GRAD only): precondition as reward_one's -- every live fence lane has |x| < LaneGradConst::x_hi (the other
// lanes' fence units have u = shape * 0.01 or shape * (width + [0, 2 fence_lo]), inside the guard by x_hi's conditions).
template <int NO, int L, bool GRAD, bool FASTDIV = false>
__device__ __forceinline__ float reward_fc(const ocd_scenario_desc &d, const float (&w)[OCD_MAX_FEATURES],
                                           float x, float y, float v, float sn, float cn,
                                           const BumpGeom (&bg)[NO > 0 ? NO : 1], const bool (&nc)[NO > 0 ? NO : 1], Q4 &q,
                                           const PkConsts &pkc, const LaneGradConst<L> *lgc = nullptr,
                                           const unsigned long long live_mask = ~0ull)
{
    // lgc (straight-line builds): the lane-feature gradient through the precomputed per-lane factors (lane_grad_qx)
    static_assert(L > 0 && NO > 0, "lane-feature reward only");
    const float tgt = d.target_speed;
    const float bound = 4.0f * (tgt * tgt);
    const float vel = v * sn;
    const float dv = vel - tgt;
    const float sq = dv * dv;
    const bool pass0 = sq <= bound;
    const float phi0 = min_tf(sq, bound);

    float rl[L], pl[L];
    float pmin = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float diff = x - d.lane_center[l];
        rl[l] = diff * -1.0f;
        const float d2 = rl[l] * rl[l];
        pl[l] = d2 * 10.0f;
        pmin = (l == 0) ? pl[0] : min_tf(pmin, pl[l]);
    }
    int ntie_min = 0;
#pragma unroll
    for (int l = 0; l < L; ++l) ntie_min += (pl[l] == pmin) ? 1 : 0;

    // the one scripted car this lane may be colliding with
    BumpGeom g = bg[0];
#pragma unroll
    for (int j = 1; j < NO; ++j) {
        g.cx = nc[j] ? bg[j].cx : g.cx; g.wx = nc[j] ? bg[j].wx : g.wx;
        g.cy = nc[j] ? bg[j].cy : g.cy; g.wy = nc[j] ? bg[j].wy : g.wy;
    }
    // ---- the two bump units (x, y) two-wide: bump_fwd's operations (ocd_devmath.h: div2_, exp_le1_2) ----
#ifdef OCD_NO_PACKED
    const float znx = (x - g.cx) / g.wx, zny = (y - g.cy) / g.wy;
#else
    const v2f ZN = div2_(v2f{x - g.cx, y - g.cy}, v2f{g.wx, g.wy});
    const float znx = ZN.x, zny = ZN.y;
#endif
    const bool condx = (znx * znx) < 1.0f, condy = (zny * zny) < 1.0f;
    const float xcx = condx ? znx : 0.0f, xcy = condy ? zny : 0.0f;
    const float ub1 = 1.0f - xcx * xcx, ub2 = 1.0f - xcy * xcy;
    // ---- the two fence units two-wide: thr_fwd's operations on the possibly-active side ----
    const bool side_p = x > d.fence_lo;
    const float z = side_p ? x : -x;
    const float xd = z - d.fence_lo;
    const bool pos1 = xd > 0.0f;
    const float uf1 = d.fence_shape * (pos1 ? xd : (0.0f + 0.01f));
    const float xd2 = d.fence_width - xd;
    const bool pos2 = xd2 > 0.0f;
    const float uf2 = d.fence_shape * (pos2 ? xd2 : (0.0f + 0.01f));
#ifdef OCD_NO_PACKED
    const float mb1 = -1.0f / ub1, mb2 = -1.0f / ub2, mf1 = -1.0f / uf1, mf2 = -1.0f / uf2;
    const float eb1 = exp_le1(mb1 + 1.0f), eb2 = exp_le1(mb2 + 1.0f), ef1 = exp_le1(mf1), ef2 = exp_le1(mf2);
#else
    const v2f UB = {ub1, ub2}, UF = {uf1, uf2};
    v2f MB, MF, KB, KF;
    if constexpr (FASTDIV && GRAD) {
        recip_pair_guarded(UB, MB, KB);
        recip_pair_guarded(UF, MF, KF);
    } else {
        MB = div2_(splat2(-1.0f), UB);
        MF = div2_(splat2(-1.0f), UF);
    }
    const v2f EB = exp_le1_2(MB + splat2(1.0f), pkc), EF = exp_le1_2(MF, pkc);
    const float mb1 = MB.x, mb2 = MB.y, mf1 = MF.x, mf2 = MF.y, eb1 = EB.x, eb2 = EB.y, ef1 = EF.x, ef2 = EF.y;
    (void)mb1; (void)mb2;
#endif
    const float bxv = condx ? eb1 : 0.0f;
    const float byv = condy ? eb2 : 0.0f;
    const float pcol = bxv * byv;                  // the reduce_max: every other car's product is exactly 0
    const float F1 = pos1 ? ef1 : 0.0f, F2 = pos2 ? ef2 : 0.0f;
    const float den = F1 + F2;
    const float Ssum = F1 / den;
    const float ax = (x < 0.0f) ? -x : x;
    const float pf = Ssum * ax;

    float r = w[0] * phi0;
#pragma unroll
    for (int l = 0; l < L; ++l) r = r + w[1 + l] * pl[l];
    const float w_min = w[L + 1], w_col = w[L + 2], w_f = w[L + 3];
    r = r + w_min * pmin;
    r = r + w_col * pcol;
    r = r + w_f * pf;
    if (!GRAD) return r;

    const float g_sq = pass0 ? w[0] : 0.0f;
    const float g_dv = (g_sq * 2.0f) * dv;
    q.qv = g_dv * sn;
    const float g_sn = g_dv * v;
    q.qth = g_sn * cn;

    float qx = 0.0f, qy = 0.0f;
    if (lgc != nullptr) {
        qx = lane_grad_qx<L>(w, pl, rl, pmin, *lgc, live_mask);
    } else {
        const float min_share = inv_count(ntie_min) * w_min;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            float gl = w[1 + l];
            gl = (pl[l] == pmin) ? (gl + min_share) : gl;
            const float g_d2 = gl * 10.0f;
            const float g_r = (g_d2 * 2.0f) * rl[l];
            qx = qx + g_r * -1.0f;
        }
    }
    // the evaluated car is among the maxima in both cases: alone (product > 0) or tied with all NO cars (product 0)
    const float col_share = ((NO == 1) ? 1.0f : ((pcol == 0.0f) ? inv_count(NO) : 1.0f)) * w_col;
    const float g_Ssum = w_f * ax;
    const float g_ax = w_f * Ssum;
#ifdef OCD_NO_PACKED
    const float kb1 = (-mb1) / ub1, kb2 = (-mb2) / ub2, kf1 = (-mf1) / uf1, kf2 = (-mf2) / uf2;
    // bump_bwd (without its head select: where cond is false the last select yields 0 whatever the chain computed)
    const float gx_q = ((col_share * byv) * eb1) * kb1;
    const float gx_xc = ((-gx_q) * 2.0f) * xcx;
    const float gy_q = ((col_share * bxv) * eb2) * kb2;
    const float gy_xc = ((-gy_q) * 2.0f) * xcy;
    const float g_znx = condx ? gx_xc : 0.0f, g_zny = condy ? gy_xc : 0.0f;
    const float qbx = g_znx / g.wx, qby = g_zny / g.wy;
    const float q1 = g_Ssum / den, q2 = (-Ssum) / den;
#else
    if constexpr (!(FASTDIV && GRAD)) { KB = div2_(-MB, UB); KF = div2_(-MF, UF); }
    const float kf1 = KF.x, kf2 = KF.y;
    v2f GXY;
    {
        const v2f B = {bxv, byv}, CS = splat2(col_share), XC = {xcx, xcy};
        asm("v_pk_mul_f32 %[g], %[b], %[cs] op_sel:[1,0] op_sel_hi:[0,1]\n"     // (byv, bxv) * col_share
            "v_pk_mul_f32 %[g], %[g], %[e]\n"
            "v_pk_mul_f32 %[g], %[g], %[k]\n"
            "v_pk_mul_f32 %[g], %[g], 2.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
            "v_pk_mul_f32 %[g], %[g], %[xc]\n"
            : [g] "=&v"(GXY) : [b] "v"(B), [cs] "v"(CS), [e] "v"(EB), [k] "v"(KB), [xc] "v"(XC));
    }
    const float g_znx = condx ? GXY.x : 0.0f, g_zny = condy ? GXY.y : 0.0f;
    const v2f QB = div2_(v2f{g_znx, g_zny}, v2f{g.wx, g.wy});
    const v2f QF = div2_(v2f{g_Ssum, -Ssum}, v2f{den, den});
    const float qbx = QB.x, qby = QB.y, q1 = QF.x, q2 = QF.y;
#endif
    qx = qx + qbx;
    qy = qy + qby;
    // thr_bwd
    const float g_den = g_Ssum * q2;
    FTape t1, t2;
    t1.pos = pos1; t1.m = mf1; t1.e = ef1; t1.u = uf1;
    t2.pos = pos2; t2.m = mf2; t2.e = ef2; t2.u = uf2;
    const float ga = f_bwd_gated(q1, d.fence_shape, t1, kf1);
    const float gb = f_bwd_gated(g_den, d.fence_shape, t1, kf1);
    const float gc = f_bwd_gated(g_den, d.fence_shape, t2, kf2);
    const float g_z = (ga + gb) + (-gc);
    qx = qx + (side_p ? g_z : -g_z);
    const float sgn = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
    qx = qx + g_ax * sgn;
    q.qx = qx; q.qy = qy;
    return r;
}

// ---------------------------------------------------------------- reward, fence + BOTH scripted cars (NO == 2)
// reward_state<2, L, GRAD>(..., do_col = true, do_fence = true) with its divisions and exponentials two-wide: the x
// bump units of the two cars are one packed pair, their y units another, the fence's two _f units a third (as in
// reward_fc).  No precondition: every feature of every lane is evaluated, same operations on the same values in the
// same order as reward_state (the packed instructions are element-wise; bump_bwd loses its head select exactly as in
// reward_fc).  11 packed + 1 scalar divisions and 3 packed exponentials per lane-step instead of 23 + 6 scalar ones:
// this is the evaluation of a pass in which some lane sits inside BOTH cars' collision boxes -- most passes of the
// slowest wavefronts of the replanning scenario, whose two scripted cars start at the same place.
template <int L, bool GRAD, bool FASTDIV = false>
__device__ __forceinline__ float reward_fcc(const ocd_scenario_desc &d, const float (&w)[OCD_MAX_FEATURES],
                                            float x, float y, float v, float sn, float cn,
                                            const BumpGeom (&bg)[2], Q4 &q, const PkConsts &pkc,
                                            const LaneGradConst<L> *lgc = nullptr, const unsigned long long live_mask = ~0ull)
{
    static_assert(L > 0, "lane-feature reward only");
    const float tgt = d.target_speed;
    const float bound = 4.0f * (tgt * tgt);
    const float vel = v * sn;
    const float dv = vel - tgt;
    const float sq = dv * dv;
    const bool pass0 = sq <= bound;
    const float phi0 = min_tf(sq, bound);

    float rl[L], pl[L];
    float pmin = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float diff = x - d.lane_center[l];
        rl[l] = diff * -1.0f;
        const float d2 = rl[l] * rl[l];
        pl[l] = d2 * 10.0f;
        pmin = (l == 0) ? pl[0] : min_tf(pmin, pl[l]);
    }
    int ntie_min = 0;
#pragma unroll
    for (int l = 0; l < L; ++l) ntie_min += (pl[l] == pmin) ? 1 : 0;

    // ---- bump_fwd of the four units: (car 0, car 1) pairs for x and for y ----
    const v2f WX = {bg[0].wx, bg[1].wx}, WY = {bg[0].wy, bg[1].wy};
    const v2f ZNX = div2_(v2f{x - bg[0].cx, x - bg[1].cx}, WX);
    const v2f ZNY = div2_(v2f{y - bg[0].cy, y - bg[1].cy}, WY);
    const bool cx0 = (ZNX.x * ZNX.x) < 1.0f, cx1 = (ZNX.y * ZNX.y) < 1.0f;
    const bool cy0 = (ZNY.x * ZNY.x) < 1.0f, cy1 = (ZNY.y * ZNY.y) < 1.0f;
    const v2f XCX = {cx0 ? ZNX.x : 0.0f, cx1 ? ZNX.y : 0.0f}, XCY = {cy0 ? ZNY.x : 0.0f, cy1 ? ZNY.y : 0.0f};
    const v2f UBX = {1.0f - XCX.x * XCX.x, 1.0f - XCX.y * XCX.y}, UBY = {1.0f - XCY.x * XCY.x, 1.0f - XCY.y * XCY.y};
    // ---- thr_fwd on the possibly-active side (see reward_state) ----
    const bool side_p = x > d.fence_lo;
    const float z = side_p ? x : -x;
    const float xd = z - d.fence_lo;
    const bool pos1 = xd > 0.0f;
    const float uf1 = d.fence_shape * (pos1 ? xd : (0.0f + 0.01f));
    const float xd2 = d.fence_width - xd;
    const bool pos2 = xd2 > 0.0f;
    const float uf2 = d.fence_shape * (pos2 ? xd2 : (0.0f + 0.01f));
    const v2f UF = {uf1, uf2};
    v2f MBX, MBY, MF, KBX, KBY, KF;
    if constexpr (FASTDIV && GRAD) {               // (precondition: reward_fc's)
        recip_pair_guarded(UBX, MBX, KBX);
        recip_pair_guarded(UBY, MBY, KBY);
        recip_pair_guarded(UF, MF, KF);
    } else {
        MBX = div2_(splat2(-1.0f), UBX);
        MBY = div2_(splat2(-1.0f), UBY);
        MF = div2_(splat2(-1.0f), UF);
    }
    const v2f EBX = exp_le1_2(MBX + splat2(1.0f), pkc), EBY = exp_le1_2(MBY + splat2(1.0f), pkc), EF = exp_le1_2(MF, pkc);
    const float bxv0 = cx0 ? EBX.x : 0.0f, bxv1 = cx1 ? EBX.y : 0.0f;
    const float byv0 = cy0 ? EBY.x : 0.0f, byv1 = cy1 ? EBY.y : 0.0f;
    const float col0 = bxv0 * byv0, col1 = bxv1 * byv1;
    const float pcol = max_tf(col0, col1);
    const int ntie_col = ((col0 == pcol) ? 1 : 0) + ((col1 == pcol) ? 1 : 0);
    const float F1 = pos1 ? EF.x : 0.0f, F2 = pos2 ? EF.y : 0.0f;
    const float den = F1 + F2;
    const float Ssum = F1 / den;
    const float ax = (x < 0.0f) ? -x : x;
    const float pf = Ssum * ax;

    float r = w[0] * phi0;
#pragma unroll
    for (int l = 0; l < L; ++l) r = r + w[1 + l] * pl[l];
    const float w_min = w[L + 1], w_col = w[L + 2], w_f = w[L + 3];
    r = r + w_min * pmin;
    r = r + w_col * pcol;
    r = r + w_f * pf;
    if (!GRAD) return r;

    const float g_sq = pass0 ? w[0] : 0.0f;
    const float g_dv = (g_sq * 2.0f) * dv;
    q.qv = g_dv * sn;
    const float g_sn = g_dv * v;
    q.qth = g_sn * cn;

    float qx = 0.0f, qy = 0.0f;
    if (lgc != nullptr) {
        qx = lane_grad_qx<L>(w, pl, rl, pmin, *lgc, live_mask);
    } else {
        const float min_share = inv_count(ntie_min) * w_min;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            float gl = w[1 + l];
            gl = (pl[l] == pmin) ? (gl + min_share) : gl;
            const float g_d2 = gl * 10.0f;
            const float g_r = (g_d2 * 2.0f) * rl[l];
            qx = qx + g_r * -1.0f;
        }
    }
    const float col_share = inv_count(ntie_col) * w_col;
    const v2f SH = {(col0 == pcol) ? col_share : 0.0f, (col1 == pcol) ? col_share : 0.0f};
    const float g_Ssum = w_f * ax;
    const float g_ax = w_f * Ssum;
    if constexpr (!(FASTDIV && GRAD)) { KBX = div2_(-MBX, UBX); KBY = div2_(-MBY, UBY); KF = div2_(-MF, UF); }
    // bump_bwd of the four units: ((share * other axis' bump) * e) * k, (-.) * 2, * xc
    v2f GX, GY;
    {
        const v2f BX = {bxv0, bxv1}, BY = {byv0, byv1};
        asm("v_pk_mul_f32 %[g], %[b], %[sh]\n"
            "v_pk_mul_f32 %[g], %[g], %[e]\n"
            "v_pk_mul_f32 %[g], %[g], %[k]\n"
            "v_pk_mul_f32 %[g], %[g], 2.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
            "v_pk_mul_f32 %[g], %[g], %[xc]\n"
            : [g] "=&v"(GX) : [b] "v"(BY), [sh] "v"(SH), [e] "v"(EBX), [k] "v"(KBX), [xc] "v"(XCX));
        asm("v_pk_mul_f32 %[g], %[b], %[sh]\n"
            "v_pk_mul_f32 %[g], %[g], %[e]\n"
            "v_pk_mul_f32 %[g], %[g], %[k]\n"
            "v_pk_mul_f32 %[g], %[g], 2.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
            "v_pk_mul_f32 %[g], %[g], %[xc]\n"
            : [g] "=&v"(GY) : [b] "v"(BX), [sh] "v"(SH), [e] "v"(EBY), [k] "v"(KBY), [xc] "v"(XCY));
    }
    const v2f QBX = div2_(v2f{cx0 ? GX.x : 0.0f, cx1 ? GX.y : 0.0f}, WX);
    const v2f QBY = div2_(v2f{cy0 ? GY.x : 0.0f, cy1 ? GY.y : 0.0f}, WY);
    const v2f QF = div2_(v2f{g_Ssum, -Ssum}, v2f{den, den});
    qx = qx + QBX.x; qy = qy + QBY.x;
    qx = qx + QBX.y; qy = qy + QBY.y;
    // thr_bwd
    const float g_den = g_Ssum * QF.y;
    FTape t1, t2;
    t1.pos = pos1; t1.m = MF.x; t1.e = EF.x; t1.u = uf1;
    t2.pos = pos2; t2.m = MF.y; t2.e = EF.y; t2.u = uf2;
    const float ga = f_bwd_gated(QF.x, d.fence_shape, t1, KF.x);
    const float gb = f_bwd_gated(g_den, d.fence_shape, t1, KF.x);
    const float gc = f_bwd_gated(g_den, d.fence_shape, t2, KF.y);
    const float g_z = (ga + gb) + (-gc);
    qx = qx + (side_p ? g_z : -g_z);
    const float sgn = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
    qx = qx + g_ax * sgn;
    q.qx = qx; q.qy = qy;
    return r;
}

// ---------------------------------------------------------------- reward, a lane inside BOTH cars' boxes without the fence (NO == 2)
// Precondition (WAVE-UNIFORM, proved by the caller): no live lane has all three of {fence, car 0's box, car 1's box} active;
// nc[j] marks the lanes inside car j's box.  Each lane runs TWO pairs of "exp(-1/u + c)" units through one instruction
// stream (round 6):
//   pair A  the bumps of ONE car -- car 1 where the lane is inside car 1's box only, else car 0;
//   pair B  car 1's bumps where the lane is inside BOTH boxes, else the fence's two _f units
// -- reward_fc with its fence pair lent to the second car on the lanes that need it.  What a lane does not evaluate is
// exactly 0 with +-0 adjoints (the fence outside its region, a car outside its box: needs_fence / needs_collision1), so
// the sums only skip +-0 terms and the result is reward_state's bit for bit: on a lane inside both boxes
// qx = (lanes + car 0) + car 1, qy = (0 + car 0) + car 1 with reduce_max's gradient shared by comparing the two products
// (the larger takes w, equal ones w / 2 each, the smaller exactly 0); elsewhere reward_fc's sums -- the one car, tied with
// the other iff its own product is 0, then the fence.  Two packed unit pairs and one shared pair of backward divisions
// instead of reward_fcc's three + three.  Gradient passes only.  FASTDIV: precondition as reward_fc's.  FASTZN: the four
// quotients (x - cx_j) / wx_j, (y - cy_j) / wy_j by the control step's refined reciprocals br[j] (quot2_by_recip);
// precondition: both cars' widths bump_widths_guarded and |x - cx_j|, |y - cy_j| >= 2^-100 on every live lane, both cars.
template <int NO, int L, bool FASTDIV, bool FASTZN>
__device__ __forceinline__ void reward_two(const ocd_scenario_desc &d, const float (&w)[OCD_MAX_FEATURES],
                                           float x, float y, float v, float sn, float cn,
                                           const BumpGeom (&bg)[NO > 0 ? NO : 1], const BumpRecip (&br)[NO > 0 ? NO : 1],
                                           const bool (&nc)[NO > 0 ? NO : 1], Q4 &q,
                                           const PkConsts &pkc, const LaneGradConst<L> &lgc, const unsigned long long live_mask)
{
    static_assert(L > 0, "lane-feature reward only");
    constexpr int J1 = NO > 1 ? 1 : 0;             // (instantiated for NO == 2 only; the index keeps other NO well-formed)
    static_assert(!FASTZN || FASTDIV, "the reciprocal quotients come with the shortened reciprocals");
    const float tgt = d.target_speed;
    const float bound = 4.0f * (tgt * tgt);
    const float vel = v * sn;
    const float dv = vel - tgt;
    const float sq = dv * dv;
    const bool pass0 = sq <= bound;

    float rl[L], pl[L];
    float pmin = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float diff = x - d.lane_center[l];
        rl[l] = diff * -1.0f;
        const float d2 = rl[l] * rl[l];
        pl[l] = d2 * 10.0f;
        pmin = (l == 0) ? pl[0] : min_tf(pmin, pl[l]);
    }

    const bool both = nc[0] && nc[J1];
    const bool a1 = nc[J1] && !nc[0];              // pair A is car 1's
    BumpGeom g = bg[0];
    g.cx = a1 ? bg[J1].cx : g.cx; g.wx = a1 ? bg[J1].wx : g.wx;
    g.cy = a1 ? bg[J1].cy : g.cy; g.wy = a1 ? bg[J1].wy : g.wy;
    v2f ZA, ZB;
    if constexpr (FASTZN) {
        const v2f RA = {a1 ? br[J1].rx : br[0].rx, a1 ? br[J1].ry : br[0].ry};
        ZA = quot2_by_recip(v2f{x - g.cx, y - g.cy}, v2f{g.wx, g.wy}, RA);
        ZB = quot2_by_recip(v2f{x - bg[J1].cx, y - bg[J1].cy}, v2f{bg[J1].wx, bg[J1].wy}, v2f{br[J1].rx, br[J1].ry});
    } else {
        ZA = div2_(v2f{x - g.cx, y - g.cy}, v2f{g.wx, g.wy});
        ZB = div2_(v2f{x - bg[J1].cx, y - bg[J1].cy}, v2f{bg[J1].wx, bg[J1].wy});
    }
    const bool cax = (ZA.x * ZA.x) < 1.0f, cay = (ZA.y * ZA.y) < 1.0f;
    const bool cbx = (ZB.x * ZB.x) < 1.0f, cby = (ZB.y * ZB.y) < 1.0f;
    const v2f XCA = {cax ? ZA.x : 0.0f, cay ? ZA.y : 0.0f}, XCB = {cbx ? ZB.x : 0.0f, cby ? ZB.y : 0.0f};
    const v2f UA = {1.0f - XCA.x * XCA.x, 1.0f - XCA.y * XCA.y};
    const bool side_p = x > d.fence_lo;
    const float z = side_p ? x : -x;
    const float xd = z - d.fence_lo;
    const bool pos1 = xd > 0.0f;
    const float uf1 = d.fence_shape * (pos1 ? xd : (0.0f + 0.01f));
    const float xd2 = d.fence_width - xd;
    const bool pos2 = xd2 > 0.0f;
    const float uf2 = d.fence_shape * (pos2 ? xd2 : (0.0f + 0.01f));
    const v2f UB = {both ? (1.0f - XCB.x * XCB.x) : uf1, both ? (1.0f - XCB.y * XCB.y) : uf2};
    const float addc = both ? 1.0f : 0.0f;
    v2f MA, MB, KA, KB;
    if constexpr (FASTDIV) {
        recip_pair_guarded(UA, MA, KA);
        recip_pair_guarded(UB, MB, KB);
    } else {
        MA = div2_(splat2(-1.0f), UA);
        MB = div2_(splat2(-1.0f), UB);
    }
    const v2f EA = exp_le1_2(MA + splat2(1.0f), pkc), EB = exp_le1_2(MB + splat2(addc), pkc);
    if constexpr (!FASTDIV) { KA = div2_(-MA, UA); KB = div2_(-MB, UB); }
    const float bxa = cax ? EA.x : 0.0f, bya = cay ? EA.y : 0.0f;
    const float bxb = cbx ? EB.x : 0.0f, byb = cby ? EB.y : 0.0f;
    const float cola = bxa * bya;
    const float other = both ? (bxb * byb) : 0.0f;           // the other car's product: exactly 0 where it is not evaluated
    const float F1 = pos1 ? EB.x : 0.0f, F2 = pos2 ? EB.y : 0.0f;
    const float den = F1 + F2;
    const float S = F1 / den;
    const float ax = (x < 0.0f) ? -x : x;

    const float g_sq = pass0 ? w[0] : 0.0f;
    const float g_dv = (g_sq * 2.0f) * dv;
    q.qv = g_dv * sn;
    const float g_sn = g_dv * v;
    q.qth = g_sn * cn;
    const float qx = lane_grad_qx<L>(w, pl, rl, pmin, lgc, live_mask);

    const float w_col = w[L + 2], w_f = w[L + 3];
    const float share = ((cola == other) ? inv_count(2) : 1.0f) * w_col;
    const v2f SHA = splat2((cola >= other) ? share : 0.0f), SHB = splat2((other >= cola) ? share : 0.0f);
    v2f GA, GB;
    {
        const v2f BA = {bxa, bya}, BB = {bxb, byb};
        asm("v_pk_mul_f32 %[g], %[b], %[sh] op_sel:[1,0] op_sel_hi:[0,1]\n"
            "v_pk_mul_f32 %[g], %[g], %[e]\n"
            "v_pk_mul_f32 %[g], %[g], %[k]\n"
            "v_pk_mul_f32 %[g], %[g], 2.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
            "v_pk_mul_f32 %[g], %[g], %[xc]\n"
            : [g] "=&v"(GA) : [b] "v"(BA), [sh] "v"(SHA), [e] "v"(EA), [k] "v"(KA), [xc] "v"(XCA));
        asm("v_pk_mul_f32 %[g], %[b], %[sh] op_sel:[1,0] op_sel_hi:[0,1]\n"
            "v_pk_mul_f32 %[g], %[g], %[e]\n"
            "v_pk_mul_f32 %[g], %[g], %[k]\n"
            "v_pk_mul_f32 %[g], %[g], 2.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
            "v_pk_mul_f32 %[g], %[g], %[xc]\n"
            : [g] "=&v"(GB) : [b] "v"(BB), [sh] "v"(SHB), [e] "v"(EB), [k] "v"(KB), [xc] "v"(XCB));
    }
    const float g_Ssum = w_f * ax;
    const float g_ax = w_f * S;
    const v2f QA = div2_(v2f{cax ? GA.x : 0.0f, cay ? GA.y : 0.0f}, v2f{g.wx, g.wy});
    const v2f QB = div2_(v2f{both ? (cbx ? GB.x : 0.0f) : g_Ssum, both ? (cby ? GB.y : 0.0f) : -S},
                         v2f{both ? bg[J1].wx : den, both ? bg[J1].wy : den});
    const float qx1 = qx + QA.x, qy1 = 0.0f + QA.y;
    const float g_den = g_Ssum * QB.y;
    FTape t1, t2;
    t1.pos = pos1; t1.m = MB.x; t1.e = EB.x; t1.u = uf1;
    t2.pos = pos2; t2.m = MB.y; t2.e = EB.y; t2.u = uf2;
    const float ga = f_bwd_gated(QB.x, d.fence_shape, t1, KB.x);
    const float gb = f_bwd_gated(g_den, d.fence_shape, t1, KB.x);
    const float gc = f_bwd_gated(g_den, d.fence_shape, t2, KB.y);
    const float g_z = (ga + gb) + (-gc);
    const float sgn = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
    const float qx_f = (qx1 + (side_p ? g_z : -g_z)) + g_ax * sgn;
    q.qx = both ? (qx1 + QB.x) : qx_f;
    q.qy = both ? (qy1 + QB.y) : qy1;
}

// every feature of every lane: the packed evaluation where there is one (two scripted cars), else reward_state
template <int NO, int L, bool GRAD, bool FASTDIV = false>
__device__ __forceinline__ float reward_every(const ocd_scenario_desc &d, const float (&w)[OCD_MAX_FEATURES],
                                              float x, float y, float v, float sn, float cn,
                                              const BumpGeom (&bg)[NO > 0 ? NO : 1], Q4 &q, const PkConsts &pkc,
                                              const LaneGradConst<(L > 0 ? L : 1)> *lgc = nullptr,
                                              const unsigned long long live_mask = ~0ull)
{
#ifndef OCD_NO_PACKED
    if constexpr (NO == 2 && L > 0) return reward_fcc<L, GRAD, FASTDIV>(d, w, x, y, v, sn, cn, bg, q, pkc, lgc, live_mask);
    else
#endif
    return reward_state<NO, L, GRAD>(d, w, x, y, v, sn, cn, bg, q, nullptr, true, true);
}

// Lane-feature gradient factors of one trajectory (its weights are fixed for the whole episode): the chain
//   g = w_l (+ w_min when lane l is the sole minimum) ; g_d2 = g * 10 ; (g_d2 * 2)
// of the backward pass evaluated once, outside the SGD loop (same operations, same order, hoisted).
// x_hi: the guard of reward_one's FASTDIV form (recip_pair_guarded, ocd_devmath.h) -- a pass may use it if every live
// fence lane has |x| < x_hi (the callers test that; 0 = never).  The denominators of -1/u on a fence lane are
// u1 = shape * xd and u2 = shape * (xd2 > 0 ? xd2 : 0.01) with xd = |x| - fence_lo > 0 and xd2 = width - xd.  Two floats
// that differ differ by a unit in the last place at least: xd >= fence_lo * 2^-24, and xd2 >= width * 2^-25 where it is
// positive (and xd2 <= width).  With shape * fence_lo >= 2^-5, 2^-5 <= shape * width <= 2^39 and
// 2^-31 <= shape * 0.01 <= 2^39 (checked here, once per kernel) every such denominator is >= 2^-31 and u2 <= 2^39;
// u1 <= shape * |x| < 2^38 (1 + 2^-22) is the per-pass test.  On the other lanes the denominators are 1 - xc^2 with
// xc^2 < 1 in fp32, i.e. in [2^-24, 1]; where an evaluation runs the fence units on a lane outside the fence region
// (reward_fc, reward_fcc) theirs are shape * 0.01 and shape * (width + [0, 2 fence_lo]) <= 3 * 2^39 (a <= 2^39 too) --
// all far inside what recip_pair_guarded needs.

template <int L>
__device__ __forceinline__ LaneGradConst<L> lane_grad_const(const float (&w)[OCD_MAX_FEATURES], const ocd_scenario_desc &d)
{
    LaneGradConst<L> c;
    {
        const float a = d.fence_shape * d.fence_lo, b = d.fence_shape * d.fence_width, k = d.fence_shape * (0.0f + 0.01f);
        const float lo5 = 0.03125f, hi39 = 549755813888.0f, lo31 = 4.656612873077392578125e-10f;
        const bool ok = a >= lo5 && a <= hi39 && b >= lo5 && b <= hi39 && k >= lo31 && k <= hi39;      // (false for NaN)
        c.x_hi = ok ? 274877906944.0f / d.fence_shape : 0.0f;
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        c.g0[l] = (w[1 + l] * 10.0f) * 2.0f;
        c.g1[l] = ((w[1 + l] + inv_count(1) * w[L + 1]) * 10.0f) * 2.0f;
        asm volatile("" : "+v"(c.g0[l]), "+v"(c.g1[l]));     // keep the products: do not fold the select back in
    }
    return c;
}

// ---------------------------------------------------------------- reward, one active feature per lane
// Precondition (WAVE-UNIFORM, proved by the caller with needs_fence / needs_collision1): every live lane
// has AT MOST ONE active feature among {fence, collision with scripted car 0, ..., car NO-1}; `is_f`
// marks the lanes whose active feature is the fence, nc[j] the lanes inside car j's bump box.
//
// The fence is built from two "exp(-1/u + c)" units -- _f(x_diff) and _f(width - x_diff), c = 0 -- and so
// is a collision term -- the x bump and the y bump, c = 1.  Each lane feeds the two units of ITS feature
// through one shared instruction stream; the features it does not evaluate are exactly 0 with +-0
// adjoints (see needs_fence / needs_collision1), so the result is the full evaluation's bit for bit:
// same operations on the same values (m + 0.0f == m), sums that only skip +-0 terms.  Likewise the two
// divisions of a lane's backward pass (g/den and (-S)/den of the fence, g_zn/width of the two bumps)
// share two division slots.
//   collision ties (reduce_max over cars, merging.py:78): the cars a lane does not evaluate have
//   col == 0 exactly, so the evaluated car is the maximum, tied with all others iff its own col is 0.
// has_col / has_f (wave-uniform): some live lane is a collision / fence lane.
// SUB = false: has_col / has_f are taken as true, the evaluation is one straight line.  A lone wavefront
// pays ~2 issue slots for a branch it does not take and ~5-6 for one it takes (tools/microbench), and at
// one wavefront per SIMD the launch lasts as long as its slowest wavefront, which has both kinds of lane
// in nearly every pass: the skips only help where several wavefronts share a SIMD.
// PRE0 (GRAD only): the caller has filled q.qv / q.qth with the target-speed feature's adjoint already (the
// V_ROW build computes it in the hazard slots of the position recurrence, ocd_chains.h).
// FASTDIV (GRAD only): preconditions, tested by the caller for every live lane -- a fence lane has |x| < lgc.x_hi (see
// LaneGradConst); with ONE scripted car also: its half-widths are bump_widths_guarded, br[0] their refined reciprocals,
// and |x - cx|, |y - cy| >= 2^-100 (quot2_by_recip).  (With two cars the per-lane choice of the reciprocals and the four
// extra tests cost more than the shorter division returns -- most of their passes take reward_fc / reward_every anyway:
// measured +1.5...+2.2 % on the per-GPU shares of configs 4 / 5.)
template <int NO, int L, bool GRAD, bool SUB = true, bool PRE0 = false, bool FASTDIV = false, bool FASTZN = FASTDIV && NO == 1>
__device__ __forceinline__ float reward_one(const ocd_scenario_desc &d, const float (&w)[OCD_MAX_FEATURES],
                                            float x, float y, float v, float sn, float cn,
                                            const BumpGeom (&bg)[NO > 0 ? NO : 1], const BumpRecip (&br)[NO > 0 ? NO : 1],
                                            const bool (&nc)[NO > 0 ? NO : 1],
                                            const bool is_f, const bool has_col_, const bool has_f_, Q4 &q,
                                            const PkConsts &pkc, const LaneGradConst<L> &lgc,
                                            const unsigned long long live_mask)
{
    static_assert(L > 0 && NO > 0, "lane-feature reward only");
    static_assert(!PRE0 || GRAD, "PRE0 is a gradient-pass option");
    static_assert(!FASTZN || (FASTDIV && NO == 1), "the reciprocal form of (x - cx) / wx: one scripted car, with FASTDIV");
    const bool has_col = SUB ? has_col_ : true, has_f = SUB ? has_f_ : true;
    const float tgt = d.target_speed;
    const float bound = 4.0f * (tgt * tgt);
    const float vel = v * sn;
    const float dv = vel - tgt;
    const float sq = dv * dv;
    const bool pass0 = sq <= bound;
    const float phi0 = min_tf(sq, bound);

    float rl[L], pl[L];
    float pmin = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float diff = x - d.lane_center[l];
        rl[l] = diff * -1.0f;
        const float d2 = rl[l] * rl[l];
        pl[l] = d2 * 10.0f;
        pmin = (l == 0) ? pl[0] : min_tf(pmin, pl[l]);
    }

    // the one scripted car this lane may be colliding with
    BumpGeom g = bg[0];
    BumpRecip gr = br[0];
#pragma unroll
    for (int j = 1; j < NO; ++j) {
        g.cx = nc[j] ? bg[j].cx : g.cx; g.wx = nc[j] ? bg[j].wx : g.wx;
        g.cy = nc[j] ? bg[j].cy : g.cy; g.wy = nc[j] ? bg[j].wy : g.wy;
    }
    // inputs of the fence units
    const bool side_p = x > d.fence_lo;
    const float z = side_p ? x : -x;
    const float xd = z - d.fence_lo;
    const bool pos1 = xd > 0.0f;
    const float uf1 = d.fence_shape * (pos1 ? xd : (0.0f + 0.01f));
    const float xd2 = d.fence_width - xd;
    const bool pos2 = xd2 > 0.0f;
    const float uf2 = d.fence_shape * (pos2 ? xd2 : (0.0f + 0.01f));
    // inputs of the bump units
    bool condx = false, condy = false;
    float xcx = 0.0f, xcy = 0.0f;
    if (has_col) {
#ifdef OCD_NO_PACKED
        const float znx = (x - g.cx) / g.wx;
        const float zny = (y - g.cy) / g.wy;
#else
        v2f ZN;
        if constexpr (FASTZN && GRAD) ZN = quot2_by_recip(v2f{x - g.cx, y - g.cy}, v2f{g.wx, g.wy}, v2f{gr.rx, gr.ry});
        else ZN = div2_(v2f{x - g.cx, y - g.cy}, v2f{g.wx, g.wy});
        const float znx = ZN.x, zny = ZN.y;
#endif
        condx = (znx * znx) < 1.0f;
        xcx = condx ? znx : 0.0f;
        condy = (zny * zny) < 1.0f;
        xcy = condy ? zny : 0.0f;
    }
    // the two shared units
    const float u1 = is_f ? uf1 : (1.0f - xcx * xcx);
    const float u2 = is_f ? uf2 : (1.0f - xcy * xcy);
    const float addc = is_f ? 0.0f : 1.0f;
#ifdef OCD_NO_PACKED
    const float m1 = -1.0f / u1, m2 = -1.0f / u2;
    const float e1 = exp_le1(m1 + addc), e2 = exp_le1(m2 + addc);
#else
    const v2f U = {u1, u2};
    v2f M, Kk;
    if constexpr (FASTDIV && GRAD) {
        // both quotients of this pass by the denominators U at once, without the scaling / fix-up instructions: the
        // caller has checked the guard (LaneGradConst::x_hi) for every live fence lane
        recip_pair_guarded(U, M, Kk);
    } else {
        M = div2_(splat2(-1.0f), U);
    }
    const v2f E = exp_le1_2(M + splat2(addc), pkc);
    const float m1 = M.x, m2 = M.y, e1 = E.x, e2 = E.y;
#endif
    // fence outputs (meaningful on fence lanes)
    float den = 1.0f, S = 0.0f, Ssum = 0.0f, ax = 0.0f, pf = 0.0f;
    if (has_f) {
        const float F1 = pos1 ? e1 : 0.0f, F2 = pos2 ? e2 : 0.0f;
        den = F1 + F2;
        S = F1 / den;
        ax = (x < 0.0f) ? -x : x;
        Ssum = GRAD ? S : (is_f ? S : 0.0f);       // GRAD: only the fence lanes' adjoint reads it (selected at the end)
        pf = is_f ? (S * ax) : 0.0f;
    }
    // bump outputs (meaningful on the other lanes)
    const float bxv = condx ? e1 : 0.0f;
    const float byv = condy ? e2 : 0.0f;
    const float pcol = is_f ? 0.0f : (bxv * byv);

    const float w_min = w[L + 1], w_col = w[L + 2], w_f = w[L + 3];
    if (!GRAD) {
        float r = w[0] * phi0;
#pragma unroll
        for (int l = 0; l < L; ++l) r = r + w[1 + l] * pl[l];
        r = r + w_min * pmin;
        r = r + w_col * pcol;                      // a skipped feature's term is exactly +-0
        r = r + w_f * pf;
        return r;
    }

#ifdef OCD_NO_PACKED
    const float k1 = (-m1) / u1, k2 = (-m2) / u2;
#else
    if constexpr (!(FASTDIV && GRAD)) Kk = div2_(-M, U);
    const float k1 = Kk.x, k2 = Kk.y;
#endif
    if constexpr (!PRE0) {
        const float g_sq = pass0 ? w[0] : 0.0f;
        const float g_dv = (g_sq * 2.0f) * dv;
        q.qv = g_dv * sn;
        const float g_sn = g_dv * v;
        q.qth = g_sn * cn;
    }

    // reduce_min over the lanes: the gradient goes to the minimum, split equally among exact ties.  Ties are
    // rare: unless some live lane has one (wave-uniform test), the per-lane factors come precomputed.
    // "some live lane has two lanes at the minimum" from the lane masks of the comparisons the selects below need anyway
    // (scalar and / or: free beside the vector stream) instead of counting the ties per lane (two selects, an add, a
    // compare: four vector slots per pass for an event that almost never happens)
    bool tie[L];
    unsigned long long tie_m[L], tie_two = 0ull, tie_seen = 0ull;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        tie[l] = pl[l] == pmin;
        tie_m[l] = __builtin_amdgcn_ballot_w64(tie[l]);
        tie_two |= tie_seen & tie_m[l];
        tie_seen |= tie_m[l];
    }
    float qx = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float g_r = (tie[l] ? lgc.g1[l] : lgc.g0[l]) * rl[l];
        qx = qx + g_r * -1.0f;
    }
    if (__builtin_expect((tie_two & live_mask) != 0ull, 0)) {
        float pm = pmin;
        asm volatile("" : "+v"(pm));               // (the count stays HERE: not speculated into the pass by the optimiser)
        int ntie_min = 0;
#pragma unroll
        for (int l = 0; l < L; ++l) ntie_min += (pl[l] == pm) ? 1 : 0;
        const float min_share = inv_count(ntie_min) * w_min;
        qx = 0.0f;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            float gl = w[1 + l];
            gl = tie[l] ? (gl + min_share) : gl;
            const float g_d2 = gl * 10.0f;
            const float g_r = (g_d2 * 2.0f) * rl[l];
            qx = qx + g_r * -1.0f;
        }
    }
    // collision adjoint up to the division by the bump width (zero on fence lanes)
    const float col_share = ((NO == 1) ? 1.0f : ((pcol == 0.0f) ? inv_count(NO) : 1.0f)) * w_col;
    // (no select on is_f / cond at the head of these chains: on fence lanes the results are discarded by the
    //  selects below, and where cond is false the last select of a chain yields 0 whatever it computed)
#ifdef OCD_NO_PACKED
    const float g_bx = col_share * byv;
    const float g_by = col_share * bxv;
    const float gx_q = (g_bx * e1) * k1;
    const float gx_xc = ((-gx_q) * 2.0f) * xcx;
    const float gy_q = (g_by * e2) * k2;
    const float gy_xc = ((-gy_q) * 2.0f) * xcy;
#else
    // the x and y chains two-wide: (col_share * (byv, bxv)) * (e1, e2) * (k1, k2), (-.) * 2, * (xcx, xcy)
    v2f GXY;
    {
        const v2f B = {bxv, byv}, CS = splat2(col_share), XC = {xcx, xcy};
        asm("v_pk_mul_f32 %[g], %[b], %[cs] op_sel:[1,0] op_sel_hi:[0,1]\n"     // (byv, bxv) * col_share
            "v_pk_mul_f32 %[g], %[g], %[e]\n"
            "v_pk_mul_f32 %[g], %[g], %[k]\n"
            "v_pk_mul_f32 %[g], %[g], 2.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
            "v_pk_mul_f32 %[g], %[g], %[xc]\n"
            : [g] "=&v"(GXY) : [b] "v"(B), [cs] "v"(CS), [e] "v"(E), [k] "v"(Kk), [xc] "v"(XC));
    }
    const float gx_xc = GXY.x, gy_xc = GXY.y;
#endif
    const float g_znx = condx ? gx_xc : 0.0f;
    const float g_zny = condy ? gy_xc : 0.0f;
    // fence adjoint up to its two divisions by den
    const float g_Ssum = w_f * ax;
    const float g_ax = w_f * Ssum;
    // the two shared division slots
#ifdef OCD_NO_PACKED
    const float q1 = (is_f ? g_Ssum : g_znx) / (is_f ? den : g.wx);   // g_S/den           | g_zn_x / width_x
    const float q2 = (is_f ? -S : g_zny) / (is_f ? den : g.wy);       // (-S)/den          | g_zn_y / width_y
#else
    const v2f Q = div2_(v2f{is_f ? g_Ssum : g_znx, is_f ? -S : g_zny}, v2f{is_f ? den : g.wx, is_f ? den : g.wy});
    const float q1 = Q.x, q2 = Q.y;
#endif
    float qx_f = qx;
    if (has_f) {
        const float g_den = g_Ssum * q2;
        FTape t1, t2;
        t1.pos = pos1; t1.m = m1; t1.e = e1; t1.u = u1;
        t2.pos = pos2; t2.m = m2; t2.e = e2; t2.u = u2;
        const float ga = f_bwd_gated(q1, d.fence_shape, t1, k1);
        const float gb = f_bwd_gated(g_den, d.fence_shape, t1, k1);
        const float gc = f_bwd_gated(g_den, d.fence_shape, t2, k2);
        const float g_z = (ga + gb) + (-gc);
        const float sgn = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
        qx_f = (qx + (side_p ? g_z : -g_z)) + g_ax * sgn;
    }
    q.qx = is_f ? qx_f : (qx + q1);
    q.qy = is_f ? 0.0f : (0.0f + q2);
    return 0.0f;
}

// ---------------------------------------------------------------- reward, active features as work items (V_CHUNK, batches)
// The throughput builds of the chunked kernel hold S steps per lane and several wavefronts per SIMD; on BASELINE's
// configurations about 30 % of their (lane, step) pairs have an active fence or collision feature in a given pass
// (measured on the CPU oracle).  Instead of pushing all S x 64 pairs through reward_one's two
// "exp(-1/u + c)" units, such a pass
//   1. gives every pair the features every pair has (reward_base_grad: target speed, lanes, min over lanes),
//   2. appends one WORK ITEM per active (pair, feature) to a list in LDS (prefix count over the ballot),
//   3. evaluates the list 64 items at a time (feature_item_grad: reward_one's instruction stream on the item's operands),
//   4. hands each pair its two adjoint terms back: q.qx = ((qx + car.o1) + fence.o1) + fence.o2, q.qy = 0 + car.o2 --
//      the order of reward_state's sums; a pair without the feature reads (+0, +0), and x + (+-0) == x for every x
//      but -0, which a sum that starts from +0 never is (round to nearest: (+0) + (-0) = +0, exact cancellation = +0).
// A fence and ONE car on the same pair are two independent items (the car's share of reduce_max depends on its own
// value only: the cars a pair does not evaluate have col == 0 exactly, see reward_one); two cars on one pair are not,
// and such a step takes reward_state as before.

// (1.) the adjoint of the features every state has; q.qy = +0.  Same operations as the head of reward_one.
template <int L>
__device__ __forceinline__ void reward_base_grad(const ocd_scenario_desc &d, const float (&w)[OCD_MAX_FEATURES],
                                                 float x, float v, float sn, float cn, Q4 &q,
                                                 const LaneGradConst<L> &lgc, const unsigned long long live_mask)
{
    static_assert(L > 0, "lane-feature reward only");
    const float tgt = d.target_speed;
    const float bound = 4.0f * (tgt * tgt);
    const float vel = v * sn;
    const float dv = vel - tgt;
    const float sq = dv * dv;
    const bool pass0 = sq <= bound;
    const float g_sq = pass0 ? w[0] : 0.0f;
    const float g_dv = (g_sq * 2.0f) * dv;
    q.qv = g_dv * sn;
    const float g_sn = g_dv * v;
    q.qth = g_sn * cn;

    float rl[L], pl[L];
    float pmin = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float diff = x - d.lane_center[l];
        rl[l] = diff * -1.0f;
        const float d2 = rl[l] * rl[l];
        pl[l] = d2 * 10.0f;
        pmin = (l == 0) ? pl[0] : min_tf(pmin, pl[l]);
    }
    bool tie[L];
    unsigned long long tie_two = 0ull, tie_seen = 0ull;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        tie[l] = pl[l] == pmin;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(tie[l]);
        tie_two |= tie_seen & m;
        tie_seen |= m;
    }
    float qx = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float g_r = (tie[l] ? lgc.g1[l] : lgc.g0[l]) * rl[l];
        qx = qx + g_r * -1.0f;
    }
    if (__builtin_expect((tie_two & live_mask) != 0ull, 0)) {      // two lanes at the minimum: rare, see reward_one
        float pm = pmin;
        asm volatile("" : "+v"(pm));
        int ntie_min = 0;
#pragma unroll
        for (int l = 0; l < L; ++l) ntie_min += (pl[l] == pm) ? 1 : 0;
        const float min_share = inv_count(ntie_min) * w[L + 1];
        qx = 0.0f;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            float gl = w[1 + l];
            gl = tie[l] ? (gl + min_share) : gl;
            const float g_d2 = gl * 10.0f;
            const float g_r = (g_d2 * 2.0f) * rl[l];
            qx = qx + g_r * -1.0f;
        }
    }
    q.qx = qx;
    q.qy = 0.0f;
}

// (3.) one work item: the fence of a state (is_f: a = x, wsel = w_fence) or its collision term with ONE scripted car
// (a = x - cx, dy = y - cy, the car's half-widths wx / wy -- with FASTZN their refined reciprocals rx / ry --,
// wsel = w_collision).  Preconditions as reward_one<..., FASTDIV = true>: a fence item has |x| < LaneGradConst::x_hi;
// FASTZN: widths bump_widths_guarded, |a|, |dy| >= 2^-100.  Results: fence (o1, o2) = (+-g_z, g_|x| * sign(x)),
// car (o1, o2) = (x adjoint, y adjoint).
// is_pair (two scripted cars): the state is inside BOTH cars' boxes and its two collision items sit in neighbouring lanes
// (even / odd); each lane evaluates its own car, and the share of reduce_max's gradient (merging.py:78) comes from comparing
// the two products: the larger takes w_collision, equal ones half each, the smaller exactly 0 -- reward_state's
// (col == max ? w / ties : 0).  A single item's other car has col == 0 exactly (see reward_one).
template <int NO, bool FASTZN>
__device__ __forceinline__ void feature_item_grad(const ocd_scenario_desc &d, const bool is_f, const bool is_pair, const float a,
                                                  const float dy, const float wx, const float wy, const float rx, const float ry,
                                                  const float wsel, const PkConsts &pkc, float &o1, float &o2)
{
    static_assert(!FASTZN || NO == 1, "the reciprocal form of (x - cx) / wx: one scripted car");
    static_assert(NO <= 2, "reduce_max over at most two scripted cars");
    // inputs of the fence units
    const float x = a;
    const bool side_p = x > d.fence_lo;
    const float z = side_p ? x : -x;
    const float xd = z - d.fence_lo;
    const bool pos1 = xd > 0.0f;
    const float uf1 = d.fence_shape * (pos1 ? xd : (0.0f + 0.01f));
    const float xd2 = d.fence_width - xd;
    const bool pos2 = xd2 > 0.0f;
    const float uf2 = d.fence_shape * (pos2 ? xd2 : (0.0f + 0.01f));
    // inputs of the bump units
    v2f ZN;
    if constexpr (FASTZN) ZN = quot2_by_recip(v2f{a, dy}, v2f{wx, wy}, v2f{rx, ry});
    else ZN = div2_(v2f{a, dy}, v2f{wx, wy});
    const float znx = ZN.x, zny = ZN.y;
    const bool condx = (znx * znx) < 1.0f;
    const float xcx = condx ? znx : 0.0f;
    const bool condy = (zny * zny) < 1.0f;
    const float xcy = condy ? zny : 0.0f;
    // the two shared units
    const float u1 = is_f ? uf1 : (1.0f - xcx * xcx);
    const float u2 = is_f ? uf2 : (1.0f - xcy * xcy);
    const float addc = is_f ? 0.0f : 1.0f;
    const v2f U = {u1, u2};
    v2f M, Kk;
    recip_pair_guarded(U, M, Kk);
    const v2f E = exp_le1_2(M + splat2(addc), pkc);
    const float m1 = M.x, m2 = M.y, e1 = E.x, e2 = E.y;
    const float k1 = Kk.x, k2 = Kk.y;
    // fence
    const float F1 = pos1 ? e1 : 0.0f, F2 = pos2 ? e2 : 0.0f;
    const float den = F1 + F2;
    const float S = F1 / den;
    const float ax = (x < 0.0f) ? -x : x;
    // bumps
    const float bxv = condx ? e1 : 0.0f;
    const float byv = condy ? e2 : 0.0f;
    const float pcol = bxv * byv;
    float col_share;
    if constexpr (NO == 1) {
        col_share = 1.0f * wsel;
    } else {
        // the neighbouring lane's product (quad_perm [1, 0, 3, 2]); 0 for a single item
        const float nb = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(pcol), 0xB1, 0xf, 0xf, true));
        const float other = is_pair ? nb : 0.0f;
        const float part = (pcol == other) ? inv_count(2) : 1.0f;
        col_share = (pcol < other) ? 0.0f : (part * wsel);
    }
    v2f GXY;
    {
        const v2f B = {bxv, byv}, CS = splat2(col_share), XC = {xcx, xcy};
        asm("v_pk_mul_f32 %[g], %[b], %[cs] op_sel:[1,0] op_sel_hi:[0,1]\n"     // (byv, bxv) * col_share
            "v_pk_mul_f32 %[g], %[g], %[e]\n"
            "v_pk_mul_f32 %[g], %[g], %[k]\n"
            "v_pk_mul_f32 %[g], %[g], 2.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
            "v_pk_mul_f32 %[g], %[g], %[xc]\n"
            : [g] "=&v"(GXY) : [b] "v"(B), [cs] "v"(CS), [e] "v"(E), [k] "v"(Kk), [xc] "v"(XC));
    }
    const float g_znx = condx ? GXY.x : 0.0f;
    const float g_zny = condy ? GXY.y : 0.0f;
    const float g_Ssum = wsel * ax;
    const float g_ax = wsel * S;
    const v2f Q = div2_(v2f{is_f ? g_Ssum : g_znx, is_f ? -S : g_zny}, v2f{is_f ? den : wx, is_f ? den : wy});
    const float q1 = Q.x, q2 = Q.y;
    const float g_den = g_Ssum * q2;
    FTape t1, t2;
    t1.pos = pos1; t1.m = m1; t1.e = e1; t1.u = u1;
    t2.pos = pos2; t2.m = m2; t2.e = e2; t2.u = u2;
    const float ga = f_bwd_gated(q1, d.fence_shape, t1, k1);
    const float gb = f_bwd_gated(g_den, d.fence_shape, t1, k1);
    const float gc = f_bwd_gated(g_den, d.fence_shape, t2, k2);
    const float g_z = (ga + gb) + (-gc);
    const float sgn = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
    o1 = is_f ? (side_p ? g_z : -g_z) : q1;
    o2 = is_f ? (g_ax * sgn) : q2;
}

// ---------------------------------------------------------------- terminal value (leaf_evaluation)
// ValueFeature.interpolate_value (value_interpolation.py:28-61): trilinear interpolation of a value
// table over the coarse state proj(world_state) = (x, y, v) or (x, y, v*sin(heading)); outside the grid the
// value is NaN and its gradient zero.  Layout and semantics: include/ocd.h, ocd_scenario_set_leaf_value.
struct LeafTable {
    const float *grid;        // [n0 + n1 + n2] cell boundaries, ascending per dimension
    const float *values;      // [n0, n1, n2]
    int32_t n[3];
    int32_t proj_kind;        // 0: (x, y, v)   1: (x, y, v * sin(heading))
    float g0[3], scale[3];    // first boundary and (n - 1) / span per dimension: the corner search's starting guess
};

// fill g0 / scale from the grid (once per kernel; any guess is corrected by the walk, so this is not part of the contract)
__device__ __forceinline__ void leaf_guess_setup(LeafTable &lt)
{
    const float *g = lt.grid;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float span = g[lt.n[k] - 1] - g[0];
        lt.g0[k] = g[0];
        lt.scale[k] = (span > 0.0f) ? (float)(lt.n[k] - 1) / span : 0.0f;
        g += lt.n[k];
    }
}

// last index i in [0, n-2] with grid[i] <= x (x is inside [grid[0], grid[n-1]]), with that cell's lower / upper boundary.
// Start from the uniform-grid guess c and read the FOUR boundaries around it in one go (independent loads: one LDS
// latency instead of a dependent chain): the answer is c-1, c or c+1 for every uniform grid (np.linspace: the guess is
// off by at most one).  `settled` reports whether the result satisfies the definition; the walk (leaf_corner_walk,
// exact from any start, for arbitrary ascending grids) runs only for a wavefront with an unsettled lane.
__device__ __forceinline__ int leaf_corner(const float *gr, int n, float x, float g0, float scale, bool &settled,
                                           float &lower, float &upper)
{
    int c = (int)((x - g0) * scale);
    c = c < 0 ? 0 : (c > n - 2 ? n - 2 : c);
    const float bm1 = gr[c > 0 ? c - 1 : 0], b0 = gr[c], b1 = gr[c + 1], b2 = gr[c + 2 < n ? c + 2 : n - 1];
    const bool up = (c < n - 2) && (b1 <= x);
    const bool down = !up && (c > 0) && (b0 > x);
    const int cc = up ? c + 1 : (down ? c - 1 : c);
    lower = up ? b1 : (down ? bm1 : b0);
    upper = up ? b2 : (down ? b0 : b1);
    settled = (cc == 0 || lower <= x) && (cc == n - 2 || x < upper);
    return cc;
}

__device__ __forceinline__ int leaf_corner_walk(const float *gr, int n, float x, int c)
{
    while (c < n - 2 && gr[c + 1] <= x) ++c;
    while (c > 0 && gr[c] > x) --c;
    return c;
}

// The lookup in two halves, so that a kernel can start the eight table loads (HBM / L2 latency) before the reward
// features of the other lanes and consume them afterwards:
//   leaf_prepare  coarse state, inside test, cell corner, steps, the eight corner values (loads issued)
//   leaf_finish   value and (GRAD) gradient w.r.t. the ego state -- the arithmetic of the contract, in its order
struct LeafLoad { float val[8], a[3], st[3]; bool inside; };

__device__ __forceinline__ void leaf_prepare(const LeafTable &lt, float x, float y, float v, float sn, LeafLoad &ld)
{
    const float xc[3] = {x, y, (lt.proj_kind == 1) ? (v * sn) : v};
    const float *gr[3] = {lt.grid, lt.grid + lt.n[0], lt.grid + lt.n[0] + lt.n[1]};
    bool inside = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) inside = inside && (xc[k] >= gr[k][0]) && (xc[k] <= gr[k][lt.n[k] - 1]);
    ld.inside = inside;
    int c[3];
    float lo[3], hi[3];
    bool settled = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        bool ok;
        c[k] = leaf_corner(gr[k], lt.n[k], xc[k], lt.g0[k], lt.scale[k], ok, lo[k], hi[k]);   // (clamped outside the grid)
        settled = settled && ok;
    }
    if (__builtin_expect(__ballot(inside && !settled) != 0ull, 0)) {            // a non-uniform grid: walk
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            c[k] = leaf_corner_walk(gr[k], lt.n[k], xc[k], c[k]);
            lo[k] = gr[k][c[k]];
            hi[k] = gr[k][c[k] + 1];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ld.st[k] = hi[k] - lo[k];
        ld.a[k] = xc[k] - lo[k];
    }
#pragma unroll
    for (int i0 = 0; i0 < 2; ++i0)
#pragma unroll
        for (int i1 = 0; i1 < 2; ++i1)
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
                ld.val[i0 * 4 + i1 * 2 + i2] = lt.values[((size_t)(c[0] + i0) * lt.n[1] + (c[1] + i1)) * lt.n[2] + (c[2] + i2)];
}

template <bool GRAD>
__device__ __forceinline__ float leaf_finish(const LeafTable &lt, const LeafLoad &ld, float v, float sn, float cn, Q4 &q)
{
    const float nanv = __int_as_float(0x7fc00000);
    // outside the grid the traced function returns the CONSTANT float('nan') (value_interpolation.py:59-60): the
    // value is NaN, its gradient w.r.t. the state is zero -- the other horizon steps keep their finite gradients
    if (GRAD) { q.qx = 0.0f; q.qy = 0.0f; q.qv = 0.0f; q.qth = 0.0f; }
    if (!ld.inside) return nanv;
    const float (&a)[3] = ld.a, (&st)[3] = ld.st;
    const float cell = (st[0] * st[1]) * st[2];
    // p[k][i]: (-1)**(i+1) * (x - g) + (1 - i) * step
    float p[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p[k][0] = (-1.0f * a[k]) + (1.0f * st[k]);
        p[k][1] = (1.0f * a[k]) + (0.0f * st[k]);
    }
    float sum = 0.0f;
    float ga[3] = {0.0f, 0.0f, 0.0f};               // d sum / d a[k]
#pragma unroll
    for (int i0 = 0; i0 < 2; ++i0)
#pragma unroll
        for (int i1 = 0; i1 < 2; ++i1)
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                const float val = ld.val[i0 * 4 + i1 * 2 + i2];
                const float pv01 = p[0][i0] * p[1][i1];
                const float pv = pv01 * p[2][i2];
                const float num = val * pv;
                sum = sum + num / cell;
                if (GRAD) {
                    // term = (val * pv) / cell ; pv = (p0 * p1) * p2
                    const float g_num = 1.0f / cell;
                    const float g_pv = g_num * val;
                    const float g_p2 = g_pv * pv01;
                    const float g_pv01 = g_pv * p[2][i2];
                    const float g_p0 = g_pv01 * p[1][i1];
                    const float g_p1 = g_pv01 * p[0][i0];
                    ga[0] = ga[0] + g_p0 * (i0 ? 1.0f : -1.0f);
                    ga[1] = ga[1] + g_p1 * (i1 ? 1.0f : -1.0f);
                    ga[2] = ga[2] + g_p2 * (i2 ? 1.0f : -1.0f);
                }
            }
    if (GRAD) {
        q.qx = ga[0];
        q.qy = ga[1];
        if (lt.proj_kind == 1) { q.qv = ga[2] * sn; q.qth = (ga[2] * v) * cn; }
        else { q.qv = ga[2]; q.qth = 0.0f; }
    }
    return sum;
}

// value and (GRAD) gradient w.r.t. the ego state of the terminal value at (x, y, v, heading)
template <bool GRAD>
__device__ __forceinline__ float leaf_value(const LeafTable &lt, float x, float y, float v, float sn, float cn, Q4 &q)
{
    LeafLoad ld;
    leaf_prepare(lt, x, y, v, sn, ld);
    return leaf_finish<GRAD>(lt, ld, v, sn, cn, q);     // (callers fill lt.g0 / lt.scale with leaf_guess_setup)
}

} // namespace ocd
