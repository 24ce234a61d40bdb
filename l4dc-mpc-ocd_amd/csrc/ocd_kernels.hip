// ocd_kernels.hip -- hand-written gfx950 kernels of the batched MPC planner.
//
// What runs here (reference file:line, relative to the reference tree):
//   NaivePlanner.generate_plan / mpc_reward   interact_drive/planner/naive_planner.py:33-77,81-164
//   car_dynamics_step                         interact_drive/simulation_utils.py:9-21
//   ThreeLaneTestCar.features                 experiments/merging.py:32-83
//   _f / smooth_threshold / smooth_bump       interact_drive/math_utils.py:7-31,59-97,135-180
//   LinearRewardCar.reward_fn                 interact_drive/car/linear_reward_car.py:49-55
//   CarWorld.step, Car.step, FixedPlanCar     interact_drive/world.py:79-109, car/car.py:76-87,
//                                             car/fixed_plan_car.py:25-39
//   ReplanningCarWorld                        experiments/replanning_world.py:24-36
//   MPC_ORD.eval_weights_for_init             interact_drive/reward_design/mpc_ord.py:67-106
//
// Mapping.  A workgroup is K wavefronts, K = number of
// control initialisations (3, or 6 with extra_inits); wavefront k optimises
// initialisation k.  Inside a wavefront the lanes are cut into segments; a
// segment is one trajectory (one (candidate, init, sample) episode, or one
// world state in plan mode) and lane t of the segment owns horizon step t: its
// control u_t, the state before and after step t, the reward features at the
// post-step state and their adjoint.
//
// Per SGD iteration the only sequential work is four short recurrences
// (v/heading forward, x/y forward, x/y adjoint, v/heading adjoint).  Two
// variants of how a segment's lanes exchange the terms (template ROWSCAN):
//   LDS windows -- segments of H lanes (64/H per wavefront); every lane writes
//                  its term into a zero-padded LDS row and reads a lane-shifted
//                  window, so no step needs a predicate (x + 0 == x);
//   DPP rows    -- H <= 16, one segment per 16-lane row; H-1 rounds of
//                  row_shr:1 / row_shl:1 moves between neighbouring lanes.
// Everything else -- sincos, the feature exponentials and IEEE divisions, the
// per-step Jacobian products -- is lane-parallel.  The K wavefronts meet once
// per control step (one __syncthreads) to pick the best initialisation, then
// all of them apply the chosen control to the real dynamics.
//
// Numerics: IEEE binary32, -ffp-contract=off, operation order = the
// arithmetic contract of DESIGN.md section 3; exp/sin/cos from ocd_devmath.h.
// No MFMA: there is no dense contraction on this path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocd.h"
#include "ocd_devmath.h"
#include "ocd_kernels.h"

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

struct Q4 { float qx, qy, qv, qth; };

// 1.0f / (float)n for a tie count n in [1, 4]: the correctly rounded quotients as constants
__device__ __forceinline__ float inv_count(int n)
{
    float r = 1.0f;
    r = (n == 2) ? 0.5f : r;
    r = (n == 3) ? (1.0f / 3.0f) : r;
    r = (n == 4) ? 0.25f : r;
    return r;
}

// Conservative per-lane tests for the wave-uniform skips of reward_state.
//  fence:  _f(x - lo) and _f(-x - lo) are both 0 (value AND gradient, math_utils.py:28-31) unless one
//          argument is > 0; then S = 0/den = 0, the feature is 0*|x| and every adjoint term is +-0.
//  collision: bump_x*bump_y and its gradients are +-0 unless x_norm^2 < 1 AND y_norm^2 < 1
//          (math_utils.py:171-178); |z - c| < 1.001*w is a cheap superset of (z-c)/w squared < 1.
__device__ __forceinline__ bool needs_fence(const ocd_scenario_desc &d, float x)
{
    return ((x - d.fence_lo) > 0.0f) || (((-x) - d.fence_lo) > 0.0f);
}

template <int NO>
__device__ __forceinline__ bool needs_collision(float x, float y, const BumpGeom (&bg)[NO > 0 ? NO : 1])
{
    bool need = false;
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const float dx = x - bg[j].cx, dy = y - bg[j].cy;
        const bool nx = ((dx < 0.0f) ? -dx : dx) < bg[j].wx * 1.001f;
        const bool ny = ((dy < 0.0f) ? -dy : dy) < bg[j].wy * 1.001f;
        need = need || (nx && ny);
    }
    return need;
}

// reward of one world state and (GRAD) its gradient w.r.t. the ego state
// (merging.py:44-83, linear_reward_car.py:49-55, targetSpeedRewardMaximizerCar.py:50-56)
template <int NO, int L, bool GRAD>
__device__ __forceinline__ float reward_state(const ocd_scenario_desc &d, const float (&w)[OCD_MAX_FEATURES],
                                              float x, float y, float v, float sn, float cn,
                                              const BumpGeom (&bg)[NO > 0 ? NO : 1], Q4 &q,
                                              float *feats /* nullptr or [D] global */,
                                              const bool do_col = true, const bool do_fence = true,
                                              const bool unify = false)
{
    // do_col / do_fence are WAVE-UNIFORM: false only when the caller has proved that, for every live
    // lane, the collision bumps / the fence thresholds are identically zero together with their
    // gradients (see needs_collision / needs_fence), so skipping them changes no bit of any result.
    if (L == 0) {                                  // OCD_REWARD_TARGET_SPEED (the planner KAT car)
        const float dv = v - d.target_speed;
        const float sq = dv * dv;
        if (GRAD) { q.qx = 0.0f; q.qy = 0.0f; q.qth = 0.0f; q.qv = (-1.0f * 2.0f) * dv; }
        return 0.0f - sq;
    }
    float phi[OCD_MAX_FEATURES];

    const float tgt = d.target_speed;
    const float bound = 4.0f * (tgt * tgt);
    const float vel = v * sn;
    const float dv = vel - tgt;
    const float sq = dv * dv;
    const bool pass0 = sq <= bound;
    phi[0] = min_tf(sq, bound);

    constexpr int LA = L > 0 ? L : 1;
    float rl[LA], pl[LA];
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

    BumpTape bx[NO > 0 ? NO : 1], by[NO > 0 ? NO : 1];
    float bxv[NO > 0 ? NO : 1], byv[NO > 0 ? NO : 1], col[NO > 0 ? NO : 1];
    float pcol = 0.0f;
    int ntie_col = NO;
    ThrTape tp_f;
    const bool side_p = (x - d.fence_lo) > 0.0f;
    float Ssum = 0.0f, ax = 0.0f, pf = 0.0f;
    // unify (WAVE-UNIFORM, one scripted car): every live lane needs at most ONE of {fence, collision}.
    // Both are built from two "exp(-1/u + c)" units -- _f(x_diff), _f(width - x_diff) with c = 0, or the
    // x and y bumps with c = 1 -- so each lane feeds the two units of ITS feature through one shared
    // instruction stream; the other feature of that lane is exactly 0 with +-0 adjoints (see
    // needs_fence / needs_collision).  Same operations on the same values as the separate blocks:
    // m + 0.0f == m, exp(+-0) == 1.
    bool is_f = false;
    float uk1 = 0.0f, uk2 = 0.0f;                  // (-m)/u of the two units, for the backward pass
    if (NO == 1 && unify) {
        is_f = needs_fence(d, x);
        // inputs of the fence units
        const float z = side_p ? x : -x;
        const float xd = z - d.fence_lo;
        const bool pos1 = xd > 0.0f;
        const float uf1 = d.fence_shape * (pos1 ? xd : (0.0f + 0.01f));
        const float xd2 = d.fence_width - xd;
        const bool pos2 = xd2 > 0.0f;
        const float uf2 = d.fence_shape * (pos2 ? xd2 : (0.0f + 0.01f));
        // inputs of the bump units
        const float znx = (x - bg[0].cx) / bg[0].wx;
        const bool condx = (znx * znx) < 1.0f;
        const float xcx = condx ? znx : 0.0f;
        const float zny = (y - bg[0].cy) / bg[0].wy;
        const bool condy = (zny * zny) < 1.0f;
        const float xcy = condy ? zny : 0.0f;
        // the two shared units
        const float u1 = is_f ? uf1 : (1.0f - xcx * xcx);
        const float u2 = is_f ? uf2 : (1.0f - xcy * xcy);
        const float addc = is_f ? 0.0f : 1.0f;
        const float m1 = -1.0f / u1, m2 = -1.0f / u2;
        const float e1 = exp_le1(m1 + addc), e2 = exp_le1(m2 + addc);
        if (GRAD) { uk1 = (-m1) / u1; uk2 = (-m2) / u2; }
        // fence outputs (meaningful on fence lanes)
        tp_f.t1.pos = pos1; tp_f.t1.m = m1; tp_f.t1.e = e1; tp_f.t1.u = u1;
        tp_f.t2.pos = pos2; tp_f.t2.m = m2; tp_f.t2.e = e2; tp_f.t2.u = u2;
        const float F1 = pos1 ? e1 : 0.0f, F2 = pos2 ? e2 : 0.0f;
        tp_f.den = F1 + F2;
        tp_f.S = F1 / tp_f.den;
        ax = (x < 0.0f) ? -x : x;
        Ssum = is_f ? tp_f.S : 0.0f;
        pf = is_f ? (tp_f.S * ax) : 0.0f;
        // bump outputs (meaningful on the other lanes)
        bx[0].cond = condx; bx[0].xc = xcx; bx[0].q = u1; bx[0].m = m1; bx[0].e = e1;
        by[0].cond = condy; by[0].xc = xcy; by[0].q = u2; by[0].m = m2; by[0].e = e2;
        bxv[0] = condx ? e1 : 0.0f;
        byv[0] = condy ? e2 : 0.0f;
        col[0] = is_f ? 0.0f : (bxv[0] * byv[0]);
        pcol = col[0];
        ntie_col = 1;
    } else {
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
        if (do_fence) {
            Ssum = thr_fwd(side_p ? x : -x, d.fence_lo, d.fence_width, d.fence_shape, tp_f);
            ax = (x < 0.0f) ? -x : x;
            pf = Ssum * ax;
        }
    }

    // reduce_sum(weights * feats), left to right over [phi0, lanes..., min, collision, fences]
    float r = w[0] * phi[0];
#pragma unroll
    for (int l = 0; l < L; ++l) r = r + w[1 + l] * pl[l];
    const float w_min = w[L + 1], w_col = w[L + 2], w_f = w[L + 3];
    r = r + w_min * pmin;
    if (do_col || unify) r = r + w_col * pcol;      // skipped terms are exactly +-0
    if (do_fence || unify) r = r + w_f * pf;
    if (feats) {
        feats[0] = phi[0];
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
    if (NO == 1 && unify) {
        // collision adjoint (zero on fence lanes, where col == pcol == 0 and both bump values are gated)
        const float share = (col[0] == pcol) ? (inv_count(1) * w_col) : 0.0f;
        const float g_bx = is_f ? 0.0f : (share * byv[0]);
        const float g_by = is_f ? 0.0f : (share * bxv[0]);
        {
            const float g_e = bx[0].cond ? g_bx : 0.0f;
            const float g_q = (g_e * bx[0].e) * uk1;
            const float g_xc = ((-g_q) * 2.0f) * bx[0].xc;
            const float g_zn = bx[0].cond ? g_xc : 0.0f;
            const float cx_term = g_zn / bg[0].wx;
            qx = is_f ? qx : (qx + cx_term);
        }
        {
            const float g_e = by[0].cond ? g_by : 0.0f;
            const float g_q = (g_e * by[0].e) * uk2;
            const float g_xc = ((-g_q) * 2.0f) * by[0].xc;
            const float g_zn = by[0].cond ? g_xc : 0.0f;
            const float cy_term = g_zn / bg[0].wy;
            qy = is_f ? qy : (qy + cy_term);
        }
        // fence adjoint (skipped on the other lanes, where it is +-0)
        const float g_Ssum = w_f * ax;
        const float g_ax = w_f * Ssum;
        const float g_F1a = g_Ssum / tp_f.den;
        const float g_den = g_Ssum * ((-tp_f.S) / tp_f.den);
        const float ga = f_bwd(g_F1a, d.fence_shape, tp_f.t1, uk1);
        const float gb = f_bwd(g_den, d.fence_shape, tp_f.t1, uk1);
        const float gc = f_bwd(g_den, d.fence_shape, tp_f.t2, uk2);
        const float g_z = (ga + gb) + (-gc);
        const float sgn = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
        const float qx_f = (qx + (side_p ? g_z : -g_z)) + g_ax * sgn;
        qx = is_f ? qx_f : qx;
    } else {
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
        const float g_z = thr_bwd(g_Ssum, d.fence_shape, tp_f);
        qx = qx + (side_p ? g_z : -g_z);
        const float sgn = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
        qx = qx + g_ax * sgn;
    }
    }
    q.qx = qx; q.qy = qy;
    return r;
}

// ---------------------------------------------------------------- segment exchange through LDS
// The four recurrences of an SGD iteration are prefix / suffix scans over the H lanes of a segment.
// Each lane publishes its term in LDS and then runs the scan itself, reading a lane-shifted window
// of a ZERO-PADDED row, so that no step needs a predicate:
//
//      row = [ H-1 zeros | term_0 .. term_{H-1} | H-1 zeros ]          (ROW = 3H-2 elements)
//
//   forward  (lane t needs term_0..term_{t-1}, in that order): step i reads element i+t, i.e. H-1-t
//            zeros first, then term_0..term_{t-1};  x + 0 == x exactly, so the leading steps are no-ops.
//   backward (lane t needs term_{H-1}..term_{t+1}, in that order): step i reads element 2H-2+t-i, i.e.
//            t zeros (from the upper pad) first, then term_{H-1}..term_{t+1}; the adjoint recurrences
//            started from 0 map zero inputs to 0, so the leading steps are no-ops as well.
//   The one recurrence with no neutral element (v' = v + (a - f v^2) dt) multiplies its increment by a
//   per-lane 0/1 mask inside the fma: fma(delta, 1, v) == v + delta and fma(delta, 0, v) == v, bit for bit.
//
// Wave-synchronous: DS operations of one wavefront execute in program order, so no s_barrier is
// needed inside the SGD loop; wave_barrier() only pins the compiler's schedule.
template <int H>
struct Geo {
    static constexpr int SEGS = 64 / H;              // trajectories per wavefront (max)
    static constexpr int ROW = 3 * H - 2;            // padded elements per segment row
    static constexpr int ROWS = SEGS + 1;            // +1: lanes past the last segment park here
    // floats per wavefront: one float4 plane followed by one float2 plane
    static constexpr int PLANE4 = ROWS * ROW * 4;
    static constexpr int PLANE2 = ROWS * ROW * 2;
    static constexpr int WAVE_FLOATS = (PLANE4 + PLANE2 + 3) & ~3;   // keeps every wavefront's float4 plane 16-byte aligned
    static constexpr int SEL_FLOATS = ROWS * 4;      // selection record per (buffer, wavefront)
};

__device__ __forceinline__ bool finite_(float v)
{
    return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u;
}

// DPP moves inside a 16-lane row: lane t receives lane t-1 (row_shr:1) or lane t+1 (row_shl:1); the
// first / last lane of the row has no source and keeps `old` (bound_ctrl off).  Used by the ROWSCAN
// variant, where every trajectory owns one row and lane t of the row is horizon step t.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_below(float old, float src) { return dpp_move<0x111>(old, src); }   // row_shr:1
__device__ __forceinline__ float from_above(float old, float src) { return dpp_move<0x101>(old, src); }   // row_shl:1

// neighbour value from the lane below (lane-1) without touching LDS: DPP wave_shr:1
__device__ __forceinline__ float lane_below(float v)
{
    const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    return __int_as_float(r);
}

// ---------------------------------------------------------------- the kernel
// ROWSCAN = false: up to 64/H trajectories per wavefront, recurrences through the zero-padded LDS windows
//                  (big batches: several wavefronts per SIMD hide the LDS round trips).
// ROWSCAN = true : H <= 16, one trajectory per 16-lane DPP row (up to 4 per wavefront), recurrences by
//                  row_shr:1 / row_shl:1 moves between neighbouring lanes -- no LDS round trip on the
//                  critical path of a wavefront that has its SIMD to itself (small batches).
template <int H, int NO, int L, bool ROWSCAN>
__global__ void __launch_bounds__(64 * OCD_MAX_CTRL_INITS)
mpc_kernel(const KernelParams p)
{
    using G = Geo<H>;
    constexpr int ROW = G::ROW;
    constexpr int NOA = NO > 0 ? NO : 1;
    const ocd_scenario_desc &d = p.d;

    extern __shared__ float4 lds_raw[];
    float *lds = reinterpret_cast<float *>(lds_raw);
    const int K = p.K;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    static_assert(!ROWSCAN || H <= 16, "ROWSCAN keeps a trajectory inside one 16-lane DPP row");
    const int seg = ROWSCAN ? (lane >> 4) : (lane / H);      // PACKED: G::SEGS for the parked tail lanes
    const int t = ROWSCAN ? (lane & 15) : (lane - seg * H);
    const bool in_h = t < H;                                 // ROWSCAN: lanes H..15 of a row idle along
    // zero the pads once (data slots are always written before they are read)
    for (int i = threadIdx.x; i < K * G::WAVE_FLOATS; i += blockDim.x) lds[i] = 0.0f;
    __syncthreads();
    float4 *plane4 = reinterpret_cast<float4 *>(lds + (size_t)wave * G::WAVE_FLOATS) + seg * ROW;
    float2 *plane2 = reinterpret_cast<float2 *>(lds + (size_t)wave * G::WAVE_FLOATS + G::PLANE4) + seg * ROW;
    float4 *const own4 = plane4 + (H - 1 + t);               // this lane's data slot
    float2 *const own2 = plane2 + (H - 1 + t);
    const float2 *const fwd2 = plane2 + t;                   // forward window: element i+t at step i
    const float4 *const bwd4 = plane4 + (2 * H - 2 + t);     // backward window: element 2H-2+t-i at step i
    const float2 *const bwd2 = plane2 + (2 * H - 2 + t);
    const float2 *const data2 = plane2 + (H - 1);            // the segment's H terms, in order
    float *sel = lds + (size_t)K * G::WAVE_FLOATS;           // [2][K][ROWS][4] selection records
    // 0/1 masks of the forward speed recurrence: step i updates lane t iff i >= H-1-t
    float mfw[H > 1 ? H - 1 : 1];
#pragma unroll
    for (int i = 0; i < H - 1; ++i) {
        mfw[i] = (i >= H - 1 - t) ? 1.0f : 0.0f;
        asm volatile("" : "+v"(mfw[i]));                     // keep them in registers, do not rematerialise
    }

    // segs_used <= SEGS trajectories per wavefront: small batches are spread over more wavefronts
    // (one trajectory each) so that the uniform feature skips act per trajectory; big batches pack.
    const long long prob_raw = (long long)blockIdx.x * p.segs_used + seg;
    const bool row_live = (seg < p.segs_used) && (prob_raw < p.n_problems);
    const bool live = in_h && row_live;
    const long long prob = row_live ? prob_raw : (p.n_problems - 1); // parked lanes shadow a real problem

    const float dt = d.dt, dt2 = d.dt_sq, fr = d.ego_friction, lr = d.learning_rate;
    constexpr int D = L > 0 ? L + 4 : 0;

    // ---- problem inputs -------------------------------------------------
    float ex, ey, ev, eth;                    // ego state
    float ox[NOA], oy[NOA], ov[NOA], oth[NOA];
    float w[OCD_MAX_FEATURES];
    int sample = 0;
    long long e_glob = 0;
    if (p.mode == OCD_MODE_ROLLOUT && !p.from_state) {
        e_glob = p.ep_begin + prob;           // flat (p, n, s) index
        const long long s_ = e_glob % p.S, n_ = (e_glob / p.S) % p.N, p_ = e_glob / ((long long)p.S * p.N);
        sample = (int)s_;
        const float *ini = p.ego_states + 4 * n_;
        ex = ini[0]; ey = ini[1]; ev = ini[2]; eth = ini[3];
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            ox[j] = d.other_init[j][0]; oy[j] = d.other_init[j][1];
            ov[j] = d.other_init[j][2]; oth[j] = d.other_init[j][3];
        }
#pragma unroll
        for (int k = 0; k < OCD_MAX_FEATURES; ++k) w[k] = (p.weights && k < D) ? p.weights[p_ * D + k] : 0.0f;
    } else {
        const float *ws = p.ego_states + prob * (NO + 1) * 4;
        ex = ws[0]; ey = ws[1]; ev = ws[2]; eth = ws[3];
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            ox[j] = ws[4 * (j + 1)]; oy[j] = ws[4 * (j + 1) + 1];
            ov[j] = ws[4 * (j + 1) + 2]; oth[j] = ws[4 * (j + 1) + 3];
        }
        const float *wp = p.weights ? (p.weights + (p.weights_per_problem ? prob * D : 0)) : nullptr;
#pragma unroll
        for (int k = 0; k < OCD_MAX_FEATURES; ++k) w[k] = (wp && k < D) ? wp[k] : 0.0f;
        sample = p.sample_fixed;
    }
    float wd[OCD_MAX_FEATURES];               // designer weights (uniform)
#pragma unroll
    for (int k = 0; k < OCD_MAX_FEATURES; ++k) wd[k] = d.designer_weights[k];

    const int T = p.T;
    float G_ret = 0.0f;
    const BumpGeom bg0 = {0.0f, 1.0f, 0.0f, 1.0f};

    if (p.mode == OCD_MODE_ROLLOUT && p.traj_out && live && wave == 0 && t == 0) {
        float *tr = p.traj_out + (size_t)prob * (T + 1) * (NO + 1) * 4;
        tr[0] = ex; tr[1] = ey; tr[2] = ev; tr[3] = eth;
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            tr[4 * (j + 1)] = ox[j]; tr[4 * (j + 1) + 1] = oy[j]; tr[4 * (j + 1) + 2] = ov[j]; tr[4 * (j + 1) + 3] = oth[j];
        }
    }

    for (int step = 0; step < T; ++step) {
        if (p.mode == OCD_MODE_ROLLOUT) {
            // ReplanningCarWorld.step: self.t += 1; teleport when self.t == critical_t
            if (d.teleport_step > 0 && (p.t0 + step + 1) == d.teleport_step) {
                const int car = d.teleport_car[sample];
#pragma unroll
                for (int j = 0; j < NO; ++j) {
                    if (car == j + 1) {
                        ox[j] = d.teleport_state[0]; oy[j] = d.teleport_state[1];
                        ov[j] = d.teleport_state[2]; oth[j] = d.teleport_state[3];
                    }
                }
            }
            // designer reward of the pre-step state (mpc_ord.py:99)
            BumpGeom bgd[NOA];
            bgd[0] = bg0;
#pragma unroll
            for (int j = 0; j < NO; ++j) bgd[j] = bump_geom(ox[j], oy[j], d.bump_half_x, d.bump_half_y);
            float s_, c_;
            sincos_(eth, s_, c_);
            Q4 qd;
            const float r = reward_state<NO, L, false>(d, wd, ex, ey, ev, s_, c_, bgd, qd, nullptr);
            G_ret = G_ret + r;
        }

        // ---- planner's model of the scripted cars over the horizon (naive_planner.py:51-66) ----
        BumpGeom bg[NOA];
        bg[0] = bg0;
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            float px = ox[j], py = oy[j], pv = ov[j], pth = oth[j];
            float cap_x = px, cap_y = py;
            if (p.other_plans) {
                for (int tt = 0; tt < H; ++tt) {
                    float s_, c_;
                    sincos_(pth, s_, c_);
                    const float acc = p.other_plans[(j * H + tt) * 2], angv = p.other_plans[(j * H + tt) * 2 + 1];
                    const float dist = pv * dt + (0.5f * acc) * dt2;
                    px = px + c_ * dist;
                    py = py + s_ * dist;
                    pv = pv + acc * dt;
                    pth = pth + angv * dt;
                    cap_x = (tt == t) ? px : cap_x;
                    cap_y = (tt == t) ? py : cap_y;
                }
            } else {
                float s_, c_;
                sincos_(pth, s_, c_);
                const float incx = (c_ * pv) * dt, incy = (s_ * pv) * dt;
                for (int tt = 0; tt < H; ++tt) {
                    px = px + incx;
                    py = py + incy;
                    cap_x = (tt == t) ? px : cap_x;
                    cap_y = (tt == t) ? py : cap_y;
                }
            }
            bg[j] = bump_geom(cap_x, cap_y, d.bump_half_x, d.bump_half_y);
        }

        // ---- this wavefront's control initialisation (naive_planner.py:107-116) ----
        float s0, c0;
        sincos_(eth, s0, c0);
        const float a_coast = fr * (ev * ev);
        const int k3 = wave % 3;
        float ua = (wave >= 3) ? a_coast : 0.0f;
        float uw = (k3 == 0) ? 0.0f : ((k3 == 1) ? -0.65f : 0.65f);

        // fma(delta, 0, v) == v needs a finite delta; in the masked steps delta is a function of the
        // CURRENT speed ev only, so one wave-uniform test per control step selects the exact fallback
        // (the world state itself can overflow after enough hard-braking steps; see the tests).
        const bool ev_finite = __ballot(!finite_(ev)) == 0ull;
        float loss = 0.0f;
        const int n_iter = d.n_iter;
        for (int it = 0; it <= n_iter; ++it) {
            // ===== forward =====
            const float a1 = min_tf(ua, 4.0f);
            const float a_c = max_tf(a1, -8.0f);
            const float w1 = min_tf(uw, 4.0f);
            const float w_c = max_tf(w1, -4.0f);
            const bool pass_a = (ua <= 4.0f) && (a1 >= -8.0f);
            const bool pass_w = (uw <= 4.0f) && (w1 >= -4.0f);
            const float wdt = w_c * dt;

            float v = ev, th = eth;
            if (ROWSCAN) {
                // every lane advances its own state by its own control and hands the result to the lane
                // above; after t rounds lane t holds the state before step t (lane 0 keeps the current state)
#pragma unroll
                for (int i = 0; i < H - 1; ++i) {
                    const float v_next = v + (a_c - fr * (v * v)) * dt;
                    const float th_next = th + wdt;
                    v = from_below(ev, v_next);
                    th = from_below(eth, th_next);
                }
            } else {
                *own2 = make_float2(a_c, wdt);
                __builtin_amdgcn_wave_barrier();
                if (ev_finite) {
#pragma unroll
                    for (int i = 0; i < H - 1; ++i) {
                        const float2 aw = fwd2[i];
                        const float delta = (aw.x - fr * (v * v)) * dt;
                        v = fma_(delta, mfw[i], v);        // masked (leading) steps see v = ev: delta is finite
                        th = th + aw.y;
                    }
                } else {                                   // a non-finite current speed: exact selects
#pragma unroll
                    for (int i = 0; i < H - 1; ++i) {
                        const float2 aw = fwd2[i];
                        const float vn_ = v + (aw.x - fr * (v * v)) * dt;
                        v = (i >= H - 1 - t) ? vn_ : v;
                        th = th + aw.y;
                    }
                }
            }
            // own step t: (v, th) is the state before it
            const float v2 = v * v;
            const float fv2 = fr * v2;
            const float acc = a_c - fv2;
            const float vdt = v * dt;
            const float hA = 0.5f * acc;
            const float hAdt2 = hA * dt2;
            const float dd = vdt + hAdt2;
            const float vn = v + acc * dt;
            const float thn = th + wdt;
            float sn, cn;
            sincos_(thn, sn, cn);
            float s_pre, c_pre;
            if (ROWSCAN) {
                s_pre = from_below(s0, sn);                // lane 0 of the row keeps sin/cos of the current heading
                c_pre = from_below(c0, cn);
            } else {
                s_pre = lane_below(sn);
                c_pre = lane_below(cn);
                s_pre = (t == 0) ? s0 : s_pre;
                c_pre = (t == 0) ? c0 : c_pre;
            }
            const float cd = c_pre * dd;
            const float sd = s_pre * dd;
            float x = ex, y = ey;
            if (ROWSCAN) {
                const float cd_b = from_below(0.0f, cd);   // increment of the step below (0 for lane 0)
                const float sd_b = from_below(0.0f, sd);
#pragma unroll
                for (int i = 0; i < H - 1; ++i) {
                    x = from_below(ex, x) + cd_b;          // lane 0: ex + 0
                    y = from_below(ey, y) + sd_b;
                }
            } else {
                __builtin_amdgcn_wave_barrier();
                *own2 = make_float2(cd, sd);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < H - 1; ++i) {
                    const float2 c2 = fwd2[i];
                    x = x + c2.x;
                    y = y + c2.y;
                }
            }
            const float xn = x + cd;
            const float yn = y + sd;

            Q4 q;
            constexpr bool lane_feats = L > 0;
            const bool do_fence = lane_feats && (p.no_skips || __ballot(live && needs_fence(d, xn)) != 0ull);
            const bool do_col = lane_feats && (NO > 0) && (p.no_skips || __ballot(live && needs_collision<NO>(xn, yn, bg)) != 0ull);
            // both needed, but by disjoint sets of lanes: one shared evaluation (see reward_state)
            const bool unify = (NO == 1) && lane_feats && do_col && do_fence && !p.no_unify &&
                               (__ballot(live && needs_fence(d, xn) && needs_collision<NO>(xn, yn, bg)) == 0ull);
            if (it == n_iter) {
                // ===== last pass: objective only (naive_planner.py:154) =====
                const float r = reward_state<NO, L, false>(d, w, xn, yn, vn, sn, cn, bg, q, nullptr, do_col, do_fence, unify);
                float Rsum = 0.0f;
                if (ROWSCAN) {
                    // running sum up the row: after H-1 rounds lane t holds ((0 + r_0) + r_1) + ... + r_t
                    float S = 0.0f + r;
#pragma unroll
                    for (int i = 0; i < H - 1; ++i) {
                        // the DPP move must execute in ALL lanes (a lane that skipped it would be an
                        // invalid source for its neighbour): move first, select afterwards
                        const float below = from_below(0.0f, S);
                        S = (t == 0) ? S : (below + r);
                    }
                    Rsum = S;                              // complete in lane H-1, which publishes the loss
                } else {
                    __builtin_amdgcn_wave_barrier();
                    *own2 = make_float2(r, 0.0f);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int j = 0; j < H; ++j) Rsum = Rsum + data2[j].x;
                }
                loss = -Rsum;
                break;
            }
            reward_state<NO, L, true>(d, w, xn, yn, vn, sn, cn, bg, q, nullptr, do_col, do_fence, unify);

            // ===== backward =====
            float Lx = 0.0f, Ly = 0.0f;
            if (ROWSCAN) {
                // idle lanes (t >= H) contribute nothing to the adjoints flowing down the row
                if (!in_h) { q.qx = 0.0f; q.qy = 0.0f; q.qv = 0.0f; q.qth = 0.0f; }
                const float qx_a = from_above(0.0f, q.qx); // adjoint term of the step above (0 for the top lane)
                const float qy_a = from_above(0.0f, q.qy);
#pragma unroll
                for (int i = 0; i < H - 1; ++i) {
                    Lx = qx_a + from_above(0.0f, Lx);
                    Ly = qy_a + from_above(0.0f, Ly);
                }
            } else {
                __builtin_amdgcn_wave_barrier();
                *own2 = make_float2(q.qx, q.qy);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < H - 1; ++i) {
                    const float2 qq = bwd2[-i];
                    Lx = qq.x + Lx;
                    Ly = qq.y + Ly;
                }
            }
            const float Ax = q.qx + Lx;
            const float Ay = q.qy + Ly;
            const float g_c = Ax * dd;
            const float g_s = Ay * dd;
            const float g_d = Ax * c_pre + Ay * s_pre;
            const float tau = (-g_c) * s_pre + g_s * c_pre;
            const float gv1 = g_d * dt;
            const float gA1 = (g_d * dt2) * 0.5f;
            float Lv = 0.0f, Lth = 0.0f;
            if (ROWSCAN) {
                const float gA1_m = in_h ? gA1 : 0.0f, gv1_m = in_h ? gv1 : 0.0f;
                const float v_m = in_h ? v : 0.0f, tau_m = in_h ? tau : 0.0f;
                const float qth_a = from_above(0.0f, q.qth);
                const float tau_a = from_above(0.0f, tau_m);
#pragma unroll
                for (int i = 0; i < H - 1; ++i) {
                    // what this lane's step sends down to the lane below, from what it has received so far
                    const float Av_ = q.qv + Lv;
                    const float gA_ = gA1_m + Av_ * dt;
                    const float gv2_ = (-gA_) * fr;
                    const float gv3_ = (gv2_ * 2.0f) * v_m;
                    const float Lv_down = (gv1_m + Av_) + gv3_;
                    Lv = from_above(0.0f, Lv_down);
                    Lth = (qth_a + from_above(0.0f, Lth)) + tau_a;
                }
            } else {
                __builtin_amdgcn_wave_barrier();
                *own4 = make_float4(q.qv, gA1, gv1, v);
                *own2 = make_float2(q.qth, tau);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < H - 1; ++i) {
                    const float4 b = bwd4[-i];             // (qv, gA1, gv1, v) of step j = H-1-i+t, or zeros
                    const float2 a = bwd2[-i];             // (qth, tau)
                    const float Av_ = b.x + Lv;
                    const float gA_ = b.y + Av_ * dt;
                    const float gv2_ = (-gA_) * fr;
                    const float gv3_ = (gv2_ * 2.0f) * b.w;
                    Lv = (b.z + Av_) + gv3_;
                    const float Ath_ = a.x + Lth;
                    Lth = Ath_ + a.y;
                }
            }
            const float Av = q.qv + Lv;
            const float gA = gA1 + Av * dt;
            const float Ath = q.qth + Lth;
            const float grad_a = pass_a ? gA : 0.0f;
            const float grad_w = pass_w ? (Ath * dt) : 0.0f;
            // SGD on loss = -R:  u <- u + lr * dR/du
            ua = ua + lr * grad_a;
            uw = uw + lr * grad_w;
            __builtin_amdgcn_wave_barrier();
        }

        // the objective's horizon sum is complete in every lane (LDS variant) or in lane H-1 (ROWSCAN)
        const bool has_loss = ROWSCAN ? (t == H - 1) : (t == 0);
        // ---- per-initialisation outputs (plan mode, parity tests) ----
        if (p.mode == OCD_MODE_PLAN && live) {
            if (p.all_plans_out) {
                float *o = p.all_plans_out + (((size_t)prob * K + wave) * H + t) * 2;
                o[0] = ua; o[1] = uw;
            }
            if (p.all_losses_out && has_loss) p.all_losses_out[(size_t)prob * K + wave] = loss;
        }

        // ---- first-index argmin over the K initialisations (naive_planner.py:161-162) ----
        float *selb = sel + (size_t)(step & 1) * K * G::SEL_FLOATS;
        {
            float *rec = selb + ((size_t)wave * G::ROWS + seg) * 4;
            if (has_loss) rec[0] = loss;
            if (t == 0) { rec[1] = ua; rec[2] = uw; }
        }
        __syncthreads();
        int best = 0;
        float bl = selb[((size_t)0 * G::ROWS + seg) * 4];
        for (int k = 1; k < K; ++k) {
            const float lk = selb[((size_t)k * G::ROWS + seg) * 4];
            if (lk < bl) { bl = lk; best = k; }
        }
        const float *brec = selb + ((size_t)best * G::ROWS + seg) * 4;
        const float ca = brec[1], cw = brec[2];

        if (p.mode == OCD_MODE_PLAN) {
            if (live && wave == best) {
                float *o = p.plans_out + ((size_t)prob * H + t) * 2;
                o[0] = ua; o[1] = uw;
                if (t == 0) {
                    if (p.best_loss_out) p.best_loss_out[prob] = bl;
                    if (p.best_init_out) p.best_init_out[prob] = best;
                }
            }
        } else {
            // ---- every car steps through the real dynamics (world.py:106-107) ----
            float nx, ny, nv, nth;
            dyn_step(ex, ey, ev, eth, c0, s0, ca, cw, dt, dt2, fr, nx, ny, nv, nth);
            ex = nx; ey = ny; ev = nv; eth = nth;
#pragma unroll
            for (int j = 0; j < NO; ++j) {
                const int gstep = p.t0 + step;    // FixedPlanCar.t (fixed_plan_car.py:25-31)
                const bool in_plan = gstep < d.other_plan_len[j];
                const float u0 = in_plan ? d.other_plan[j][gstep & (OCD_MAX_PLAN - 1)][0] : d.other_default[j][0];
                const float u1 = in_plan ? d.other_plan[j][gstep & (OCD_MAX_PLAN - 1)][1] : d.other_default[j][1];
                float s_, c_;
                sincos_(oth[j], s_, c_);
                dyn_step(ox[j], oy[j], ov[j], oth[j], c_, s_, u0, u1, dt, dt2, d.other_friction[j], nx, ny, nv, nth);
                ox[j] = nx; oy[j] = ny; ov[j] = nv; oth[j] = nth;
            }
            if (live && wave == 0 && t == 0) {
                if (p.ctrl_out) {
                    float *o = p.ctrl_out + ((size_t)prob * T + step) * 2;
                    o[0] = ca; o[1] = cw;
                }
                if (p.traj_out) {
                    float *tr = p.traj_out + ((size_t)prob * (T + 1) + step + 1) * (NO + 1) * 4;
                    tr[0] = ex; tr[1] = ey; tr[2] = ev; tr[3] = eth;
#pragma unroll
                    for (int j = 0; j < NO; ++j) {
                        tr[4 * (j + 1)] = ox[j]; tr[4 * (j + 1) + 1] = oy[j];
                        tr[4 * (j + 1) + 2] = ov[j]; tr[4 * (j + 1) + 3] = oth[j];
                    }
                }
            }
        }
    }
    if (p.mode == OCD_MODE_ROLLOUT && live && wave == 0 && t == 0) p.returns_out[prob] = G_ret;
}

// ---------------------------------------------------------------- small kernels
template <int NO, int L>
__global__ void reward_kernel(const KernelParams p, float *feats_out, float *reward_out)
{
    constexpr int NOA = NO > 0 ? NO : 1;
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.n_problems) return;
    const ocd_scenario_desc &d = p.d;
    constexpr int D = L > 0 ? L + 4 : 0;
    const float *ws = p.ego_states + b * (NO + 1) * 4;
    float w[OCD_MAX_FEATURES];
#pragma unroll
    for (int k = 0; k < OCD_MAX_FEATURES; ++k) w[k] = (p.weights && k < D) ? p.weights[k] : 0.0f;
    BumpGeom bg[NOA];
    bg[0] = BumpGeom{0.0f, 1.0f, 0.0f, 1.0f};
#pragma unroll
    for (int j = 0; j < NO; ++j) bg[j] = bump_geom(ws[4 * (j + 1)], ws[4 * (j + 1) + 1], d.bump_half_x, d.bump_half_y);
    float s_, c_;
    sincos_(ws[3], s_, c_);
    Q4 q;
    const float r = reward_state<NO, L, false>(d, w, ws[0], ws[1], ws[2], s_, c_, bg, q,
                                            feats_out ? feats_out + b * D : nullptr);
    if (reward_out) reward_out[b] = r;
}

// car_dynamics_step for a batch of (state, control) pairs (simulation_utils.py:9-21,73-123)
__global__ void dynamics_kernel(const float *st, const float *u, float dt, float dt2, float fr, float *out, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 s4 = reinterpret_cast<const float4 *>(st)[i];
    const float2 u2 = reinterpret_cast<const float2 *>(u)[i];
    float s_, c_;
    sincos_(s4.w, s_, c_);
    float4 o;
    dyn_step(s4.x, s4.y, s4.z, s4.w, c_, s_, u2.x, u2.y, dt, dt2, fr, o.x, o.y, o.z, o.w);
    reinterpret_cast<float4 *>(out)[i] = o;
}

__global__ void math_kernel(const float *in, float *e, float *s, float *c, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    if (e) e[i] = exp_(x);
    float sv, cv;
    sincos_(x, sv, cv);
    if (s) s[i] = sv;
    if (c) c[i] = cv;
}

} // namespace ocd

// ---------------------------------------------------------------- launch table
namespace ocd {

template <int H, int NO, int L>
static hipError_t launch_mpc(const KernelParams &p_in, hipStream_t st)
{
    using G = Geo<H>;
    KernelParams p = p_in;
    const int K = p.K;
    const size_t lds = ((size_t)K * G::WAVE_FLOATS + (size_t)2 * K * G::SEL_FLOATS) * sizeof(float);
    // Variant and packing.  What decides the kernel time of a small or medium batch is how many workgroups
    // the busiest CU gets (a workgroup is K wavefronts on the CU's 4 SIMDs; a second one doubles up two
    // SIMDs): measured on 256 CUs, 1024 trajectories take 2.45 ms as 256 workgroups and 3.6 ms as 342.
    // So: pack just enough trajectories per wavefront for one workgroup per CU; prefer the DPP-row variant
    // (faster per wavefront, at most 4 trajectories each, H <= 16) when it needs no more rounds of
    // workgroups than the LDS variant (up to 64/H each), else take the denser LDS packing.
    // scan_mode 1 / 2 and segs_per_wave force the choice (tests, sweeps).
    const long long cus = 256;
    auto ceil_div = [](long long a, long long b) { return (a + b - 1) / b; };
    auto clampi = [](long long v, long long lo, long long hi) { return (int)(v < lo ? lo : (v > hi ? hi : v)); };
    const int want = clampi(ceil_div(p.n_problems, cus), 1, 64);
    const int segs_l = p.segs_used > 0 ? clampi(p.segs_used, 1, G::SEGS) : clampi(want, 1, G::SEGS);
    const int segs_r = p.segs_used > 0 ? clampi(p.segs_used, 1, 4) : clampi(want, 1, 4);
    const long long rounds_l = ceil_div(ceil_div(p.n_problems, segs_l), cus);
    const long long rounds_r = ceil_div(ceil_div(p.n_problems, segs_r), cus);
    bool rows = false;
    if (H <= 16 && p.scan_mode != 1) rows = (p.scan_mode == 2) || rounds_r == 1 || rounds_r < rounds_l;
    p.segs_used = rows ? segs_r : segs_l;
    const long long blocks = (p.n_problems + p.segs_used - 1) / p.segs_used;
    if (rows) {
        if constexpr (H <= 16) hipLaunchKernelGGL((mpc_kernel<H, NO, L, true>), dim3((unsigned)blocks), dim3(64 * K), lds, st, p);
    } else {
        hipLaunchKernelGGL((mpc_kernel<H, NO, L, false>), dim3((unsigned)blocks), dim3(64 * K), lds, st, p);
    }
    return hipGetLastError();
}

#define OCD_CASE(HH, NN, LL) if (H == HH && NO == NN && L == LL) return launch_mpc<HH, NN, LL>(p, st);

hipError_t launch_mpc_dispatch(int H, int NO, int L, const KernelParams &p_in, hipStream_t st, bool *supported)
{
    const KernelParams &p = p_in;
    *supported = true;
    OCD_KERNEL_TABLE(OCD_CASE)
    *supported = false;
    return hipSuccess;
}

#define OCD_RCASE(NN, LL) if (NO == NN && L == LL) { hipLaunchKernelGGL((reward_kernel<NN, LL>), dim3(nb), dim3(bs), 0, st, p, feats, rew); return hipGetLastError(); }

hipError_t launch_reward(int NO, int L, const KernelParams &p, float *feats, float *rew, hipStream_t st, bool *supported)
{
    *supported = true;
    const unsigned bs = 256;
    const unsigned nb = (unsigned)((p.n_problems + bs - 1) / bs);
    OCD_REWARD_TABLE(OCD_RCASE)
    *supported = false;
    return hipSuccess;
}

hipError_t launch_dynamics(const float *states, const float *controls, float dt, float dt_sq, float friction,
                           float *out, long long n, hipStream_t st)
{
    const unsigned bs = 256;
    const unsigned nb = (unsigned)((n + bs - 1) / bs);
    hipLaunchKernelGGL(dynamics_kernel, dim3(nb), dim3(bs), 0, st, states, controls, dt, dt_sq, friction, out, n);
    return hipGetLastError();
}

hipError_t launch_math(const float *in, float *e, float *s, float *c, long long n, hipStream_t st)
{
    const unsigned bs = 256;
    const unsigned nb = (unsigned)((n + bs - 1) / bs);
    hipLaunchKernelGGL(math_kernel, dim3(nb), dim3(bs), 0, st, in, e, s, c, n);
    return hipGetLastError();
}

} // namespace ocd
