// ocd_debug_kernels.hip -- the reward evaluations of ocd_device.h side by side, gfx950 (TEST SUPPORT, never on the path).
//
// The planner kernels evaluate the same feature arithmetic (merging.py:44-83, math_utils.py:28-31,87-95,166-178 and its
// reverse mode) through five hand-written forms that must stay bit-identical wherever their preconditions hold:
//   reward_state   every feature of a lane, scalar, full divisions                       (the definition)
//   reward_one     one active feature per lane through one shared pair of units (straight-line / with sub-skips; full or
//                  shortened divisions; one scripted car: also (x - cx) / wx by the control step's reciprocals)
//   reward_fc      fence + at most one car per lane, two packed pairs (full / shortened)
//   reward_fcc     fence + both cars (two scripted cars), three packed pairs (full / shortened)   [through reward_every]
//   reward_two     at most two active features per lane (two scripted cars), two packed pairs: the lane's car + the second
//                  car or the fence (shortened; full / reciprocal quotients by the bump widths)       [adjoint only]
//   work items     reward_base_grad + one feature_item_grad per active (state, feature), compacted through LDS and evaluated
//                  64 at a time -- the gradient passes of the chunked kernel's shared-SIMD builds; a state inside both
//                  cars' boxes as a pair of neighbouring items (adjoint only; one or two scripted cars)
// The planner tests reach them through whole plans; this kernel evaluates ALL of them on caller-supplied world states, one
// state per lane, and says per lane which forms' preconditions hold -- tests/test_gpu_feature_variants.py holds every
// valid (state, form) pair to reward_state's value and adjoint, bit for bit, so a change of the contract (or another
// asm shortcut) is checked form by form before it is checked plan by plan.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocd.h"
#include "ocd_device.h"
#include "ocd_kernels.h"

namespace ocd {

enum { FV_VARIANTS = 11, FV_VALUES = 5 };   // (r, qx, qy, qv, qth) per form

template <int NO, int L>
__global__ void __launch_bounds__(64) feature_variants_kernel(const KernelParams p, float *out, int32_t *valid)
{
    static_assert(NO >= 1 && L >= 1, "lane-feature reward with scripted cars");
    const long long b_raw = (long long)blockIdx.x * 64 + threadIdx.x;
    const bool live = b_raw < p.n_problems;
    const long long b = live ? b_raw : p.n_problems - 1;
    const unsigned long long live_mask = __ballot(live);
    const ocd_scenario_desc &d = p.d;
    constexpr int D = feat_dim(L);
    const float *ws = p.ego_states + b * (NO + 1) * 4;
    float w[OCD_MAX_FEATURES];
#pragma unroll
    for (int k = 0; k < OCD_MAX_FEATURES; ++k) w[k] = (p.weights && k < D) ? p.weights[k] : 0.0f;
    const float x = ws[0], y = ws[1], v = ws[2];
    float sn, cn;
    sincos_(ws[3], sn, cn);
    BumpGeom bg[NO];
    BumpRecip br[NO];
    bool nc[NO];
    bool degenerate = false, widths_ok = true, tiny = false;
    int n_in = 0;
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        bg[j] = bump_geom(ws[4 * (j + 1)], ws[4 * (j + 1) + 1], d.bump_half_x, d.bump_half_y);
        br[j] = BumpRecip{refined_recip(bg[j].wx), refined_recip(bg[j].wy)};
        nc[j] = needs_collision1(x, y, bg[j]);
        n_in += nc[j] ? 1 : 0;
        degenerate = degenerate || bump_widths_degenerate(bg[j]);
        widths_ok = widths_ok && bump_widths_guarded(bg[j]);
        tiny = tiny || __builtin_fabsf(x - bg[j].cx) < 7.888609052210118e-31f || __builtin_fabsf(y - bg[j].cy) < 7.888609052210118e-31f;
    }
    const bool nf = needs_fence(d, x);
    const LaneGradConst<L> lgc = lane_grad_const<L>(w, d);
    const PkConsts pkc = pk_consts();
    const bool guard_f = lgc.x_hi > 0.0f && (!nf || __builtin_fabsf(x) < lgc.x_hi);
    const bool one_ok = (n_in + (nf ? 1 : 0)) <= 1 && !degenerate;
    const bool fc_ok = n_in <= 1 && !degenerate;
    const bool has_col = (__ballot(n_in > 0) & live_mask) != 0ull, has_f = (__ballot(nf) & live_mask) != 0ull;
    constexpr bool ZN = NO == 1;

    float res[FV_VARIANTS][FV_VALUES];
    bool ok[FV_VARIANTS];
    Q4 q;
    // 0: the definition
    res[0][0] = reward_state<NO, L, false>(d, w, x, y, v, sn, cn, bg, q, nullptr, true, true);
    (void)reward_state<NO, L, true>(d, w, x, y, v, sn, cn, bg, q, nullptr, true, true);
    res[0][1] = q.qx; res[0][2] = q.qy; res[0][3] = q.qv; res[0][4] = q.qth; ok[0] = true;
    // 1: one feature per lane, straight line, full divisions
    res[1][0] = reward_one<NO, L, false, false>(d, w, x, y, v, sn, cn, bg, br, nc, nf, true, true, q, pkc, lgc, live_mask);
    (void)reward_one<NO, L, true, false>(d, w, x, y, v, sn, cn, bg, br, nc, nf, true, true, q, pkc, lgc, live_mask);
    res[1][1] = q.qx; res[1][2] = q.qy; res[1][3] = q.qv; res[1][4] = q.qth; ok[1] = one_ok;
    // 2: ... shortened divisions (one car: also the reciprocal form of (x - cx) / wx)
    res[2][0] = res[1][0];
    (void)reward_one<NO, L, true, false, false, true, ZN>(d, w, x, y, v, sn, cn, bg, br, nc, nf, true, true, q, pkc, lgc, live_mask);
    res[2][1] = q.qx; res[2][2] = q.qy; res[2][3] = q.qv; res[2][4] = q.qth;
    ok[2] = one_ok && guard_f && (!ZN || (widths_ok && !tiny));
    // 3: ... with the wave-uniform sub-skips of the throughput builds
    res[3][0] = reward_one<NO, L, false, true>(d, w, x, y, v, sn, cn, bg, br, nc, nf, has_col, has_f, q, pkc, lgc, live_mask);
    (void)reward_one<NO, L, true, true>(d, w, x, y, v, sn, cn, bg, br, nc, nf, has_col, has_f, q, pkc, lgc, live_mask);
    res[3][1] = q.qx; res[3][2] = q.qy; res[3][3] = q.qv; res[3][4] = q.qth; ok[3] = one_ok;
    // 4 / 5: fence + at most one car per lane (full / shortened, the latter with the precomputed lane-gradient factors)
    res[4][0] = reward_fc<NO, L, false>(d, w, x, y, v, sn, cn, bg, nc, q, pkc);
    (void)reward_fc<NO, L, true>(d, w, x, y, v, sn, cn, bg, nc, q, pkc);
    res[4][1] = q.qx; res[4][2] = q.qy; res[4][3] = q.qv; res[4][4] = q.qth; ok[4] = fc_ok;
    res[5][0] = res[4][0];
    (void)reward_fc<NO, L, true, true>(d, w, x, y, v, sn, cn, bg, nc, q, pkc, &lgc, live_mask);
    res[5][1] = q.qx; res[5][2] = q.qy; res[5][3] = q.qv; res[5][4] = q.qth; ok[5] = fc_ok && guard_f;
    // 6 / 7: every feature, packed where there is a packed form (two scripted cars: reward_fcc)
    res[6][0] = reward_every<NO, L, false>(d, w, x, y, v, sn, cn, bg, q, pkc);
    (void)reward_every<NO, L, true>(d, w, x, y, v, sn, cn, bg, q, pkc, &lgc, live_mask);
    res[6][1] = q.qx; res[6][2] = q.qy; res[6][3] = q.qv; res[6][4] = q.qth; ok[6] = true;
    res[7][0] = res[6][0];
    (void)reward_every<NO, L, true, true>(d, w, x, y, v, sn, cn, bg, q, pkc, &lgc, live_mask);
    res[7][1] = q.qx; res[7][2] = q.qy; res[7][3] = q.qv; res[7][4] = q.qth; ok[7] = guard_f && !degenerate;
    // 8: the work-item form (mpc_chunk_kernel's gradient passes; same order in the list: singles, pairs from an even slot, fences)
    res[8][0] = res[0][0];
    ok[8] = false;
    if constexpr (NO <= 2) {
        constexpr int ZERO = 3 * 64 + 64, FIELDS = (NO == 1) ? 8 : 6;
        __shared__ float item_lds[FIELDS][ZERO + 1];
        if (threadIdx.x == 0) { item_lds[0][ZERO] = 0.0f; item_lds[1][ZERO] = 0.0f; }
        __syncthreads();
        reward_base_grad<L>(d, w, x, v, sn, cn, q, lgc, live_mask);
        const bool both = n_in == 2, single = n_in == 1;
        int jc = 0;
#pragma unroll
        for (int j = 1; j < NO; ++j) jc = nc[j] ? j : jc;
        const unsigned long long ms = __ballot(single && live), mp = __ballot(both && live), mf = __ballot(nf && live);
        int n_items = 0, slot_c = ZERO, slot_d = ZERO, slot_f = ZERO;
        if (single && live) {
            const int ic = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ms >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ms, 0u));
            item_lds[0][ic] = x - bg[jc].cx; item_lds[1][ic] = y - bg[jc].cy; item_lds[2][ic] = bg[jc].wx; item_lds[3][ic] = bg[jc].wy;
            item_lds[4][ic] = w[L + 2]; item_lds[5][ic] = 0.0f;
            if constexpr (NO == 1) { item_lds[6][ic] = br[0].rx; item_lds[7][ic] = br[0].ry; }
            slot_c = ic;
        }
        n_items += __popcll(ms);
        n_items += n_items & 1;
        if constexpr (NO == 2) {
            if (both && live) {
                const int id = n_items + 2 * (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mp >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mp, 0u));
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    item_lds[0][id + j] = x - bg[j].cx; item_lds[1][id + j] = y - bg[j].cy;
                    item_lds[2][id + j] = bg[j].wx; item_lds[3][id + j] = bg[j].wy; item_lds[4][id + j] = w[L + 2]; item_lds[5][id + j] = 2.0f;
                }
                slot_c = id; slot_d = id + 1;
            }
            n_items += 2 * __popcll(mp);
        }
        if (nf && live) {
            const int jf = n_items + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mf >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mf, 0u));
            item_lds[0][jf] = x; item_lds[4][jf] = w[L + 3]; item_lds[5][jf] = 1.0f;
            slot_f = jf;
        }
        n_items += __popcll(mf);
        __syncthreads();
        for (int base = 0; base < n_items; base += 64) {
            const int i = base + (int)threadIdx.x;
            float o1, o2, irx = 1.0f, iry = 1.0f;
            const float ia = item_lds[0][i], idy = item_lds[1][i], iwx = item_lds[2][i], iwy = item_lds[3][i], iws = item_lds[4][i], ity = item_lds[5][i];
            if constexpr (NO == 1) { irx = item_lds[6][i]; iry = item_lds[7][i]; }
            feature_item_grad<NO, NO == 1>(d, ity == 1.0f, ity == 2.0f, ia, idy, iwx, iwy, irx, iry, iws, pkc, o1, o2);
            item_lds[0][i] = o1; item_lds[1][i] = o2;
        }
        __syncthreads();
        const float c1 = item_lds[0][slot_c], c2 = item_lds[1][slot_c], d1 = item_lds[0][slot_d], d2 = item_lds[1][slot_d];
        const float f1 = item_lds[0][slot_f], f2 = item_lds[1][slot_f];
        res[8][1] = (((q.qx + c1) + d1) + f1) + f2; res[8][2] = (q.qy + c2) + d2; res[8][3] = q.qv; res[8][4] = q.qth;
        // (a lane outside the guards spoils only its own items)
        ok[8] = guard_f && !degenerate && (!ZN || (widths_ok && !tiny));
    } else {
#pragma unroll
        for (int c = 1; c < FV_VALUES; ++c) res[8][c] = 0.0f;
    }
    // 9 / 10: at most two active features per lane, two scripted cars (reward_two: the latency builds' both-boxes steps);
    // adjoint only: r repeats form 0
    res[9][0] = res[0][0]; res[10][0] = res[0][0];
    ok[9] = false; ok[10] = false;
#pragma unroll
    for (int c = 1; c < FV_VALUES; ++c) { res[9][c] = 0.0f; res[10][c] = 0.0f; }
    if constexpr (NO == 2) {
        const bool two_ok = (n_in + (nf ? 1 : 0)) <= 2 && !degenerate && guard_f;
        reward_two<NO, L, true, false>(d, w, x, y, v, sn, cn, bg, br, nc, q, pkc, lgc, live_mask);
        res[9][1] = q.qx; res[9][2] = q.qy; res[9][3] = q.qv; res[9][4] = q.qth; ok[9] = two_ok;
        reward_two<NO, L, true, true>(d, w, x, y, v, sn, cn, bg, br, nc, q, pkc, lgc, live_mask);
        res[10][1] = q.qx; res[10][2] = q.qy; res[10][3] = q.qv; res[10][4] = q.qth; ok[10] = two_ok && widths_ok && !tiny;
    }
    if (!live) return;
#pragma unroll
    for (int k = 0; k < FV_VARIANTS; ++k) {
        valid[b * FV_VARIANTS + k] = ok[k] ? 1 : 0;
#pragma unroll
        for (int c = 0; c < FV_VALUES; ++c) out[(b * FV_VARIANTS + k) * FV_VALUES + c] = res[k][c];
    }
}

#define OCD_FVCASE(NN, LL) if (NO == NN && L == LL) { hipLaunchKernelGGL((feature_variants_kernel<NN, LL>), dim3(nb), dim3(64), 0, st, p, out, valid); return hipGetLastError(); }

hipError_t launch_feature_variants(int NO, int L, const KernelParams &p, float *out, int32_t *valid, hipStream_t st, bool *supported)
{
    *supported = true;
    const unsigned nb = (unsigned)((p.n_problems + 63) / 64);
    OCD_FVCASE(1, 2) OCD_FVCASE(1, 3) OCD_FVCASE(2, 2) OCD_FVCASE(2, 3) OCD_FVCASE(3, 3)
    *supported = false;
    return hipSuccess;
}

} // namespace ocd
