// ocd_chunk_kernel.hip -- the planner kernel for batches beyond one lane-per-step wavefront per SIMD (V_CHUNK), gfx950.
//
// Same algorithm, arithmetic contract and entry points as ocd::mpc_kernel (ocd_kernels.hip; reference:
// naive_planner.py:33-77,81-164, simulation_utils.py:9-21, merging.py:32-83, mpc_ord.py:67-106); what
// changes is the mapping.  With one lane per horizon step every lane runs all H-1 rounds of the four
// horizon recurrences -- O(H^2) lane-work, two thirds of the kernel time at H = 25.  Here a lane owns a
// CHUNK of S consecutive steps, NC = ceil(H/S) lanes per (trajectory, control initialisation) pair (S need not divide
// H: the last lane of a segment then owns fewer steps and padding, see SL below):
//
//   * the recurrences run NC-1 rounds, each walking the lane's S steps sequentially and handing the
//     chunk's end value to the neighbouring lane (wave_shr:1 / wave_shl:1 + a select at the segment
//     boundary, as in V_SEG): the rounds of a wavefront cost about what they cost before, but the
//     wavefront now holds S times as many problems (64/NC segments instead of 64/H);
//   * the lane-parallel work of a step (sincos, reward features, Jacobian products) is done S times per
//     lane, at full lane utilisation -- it becomes 80 % of a pass, which is what the algorithm needs.
//
// ONE wavefront per workgroup; segment (j, k) = trajectory j, initialisation k; the first-index argmin
// over the K initialisations is taken inside the wavefront (ds_bpermute).
// Sums keep the reference's sequential order (the chunk's partial value travels up / down the segment),
// so results are bit-identical to the other variants and to the oracle.
//
// The builds that share a SIMD with other wavefronts (round 5): a wavefront holds S x 64 (lane, step) pairs and on
// BASELINE's configurations about 30 % of them have an active fence / collision feature in a given pass, so the gradient
// passes do not push every pair through the two "exp(-1/u + c)" units: the active (pair, feature) terms go to a
// work-item list in LDS (6-10 KB per wavefront), are evaluated 64 at a time and come back as two adjoint terms each
// (horizon_pass; ocd_device.h: reward_base_grad / feature_item_grad).  Config 5 whole 37.6 -> 30.8 ms, config 4 whole
// 18.2 -> 14.2 ms, 16 x config 3 13.4 -> 11.7 ms (profiles/r05_items_ab.txt).  A wavefront alone on its SIMD (LAT) keeps
// the straight-line evaluation of every pair: there the list's LDS round trips are not hidden (measured: +-0 ... +20 %).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ocd.h"
#include "ocd_chunk_chains.h"
#include "ocd_device.h"
#include "ocd_kernels.h"
#include "ocd_lane.h"

#ifndef OCD_CHUNK_OCC
#define OCD_CHUNK_OCC
#endif

namespace ocd {

// LAT: the launch puts at most one wavefront on a SIMD (the per-GPU shares of BASELINE configs 4 / 5: 1 024
//   wavefronts), so it lasts as long as its slowest wavefront and nothing hides that wavefront's branches and its
//   ballot -> SALU -> branch stalls (ocd_kernels.hip, LAT).  Per step ONE test chooses between one feature per lane
//   (reward_one, straight line, no sub-skips) and the multi-feature evaluations (reward_fc / reward_every); the build
//   also claims its SIMD (see the a255 clobber below).
//   The diagnostics knobs no_feature_skips / no_unified_features select the LAT = false build.
// OCC3: compiled for THREE wavefronts per SIMD (<= 168 VGPRs; hipcc spills 20-70 dwords of the per-step tape to
//   scratch, which three wavefronts hide) -- FOUR (128 VGPRs) for S <= 2: launches with more wavefronts than the
//   unconstrained build keeps resident (two per SIMD at S > 2, three at S <= 2) take it: config 4 whole 19.6 -> 18.7 ms,
//   config 5 whole 41.0 -> 39.3 ms in round 3; 24 576 / 28 672 episodes of config 3's shape at S = 5: 10.5 -> 9.0 ms,
//   10.4 -> 9.0 ms in round 5; smaller launches lose 3-4 % and keep the unconstrained build.  Same code, same results.
template <int HT, int NO, int L, int S, bool LAT = false, bool OCC3 = false>
__global__ void __launch_bounds__(64, OCC3 ? (S <= 2 ? 4 : 3) : 1) OCD_CHUNK_OCC
mpc_chunk_kernel(const KernelParams p)
{
    static_assert(!(LAT && OCC3), "the latency build runs alone on its SIMD");
    constexpr int H = HT;
    constexpr int NC = (HT + S - 1) / S;                       // lanes per (trajectory, initialisation)
    // steps of a segment's LAST lane; S does not have to divide H: the rest of that lane's chunk is padding ("dead"
    // steps s >= SL): evaluated with the wavefront, kept out of its feature decisions, and entered on the tape as
    // zeros (zero step length, zero feature adjoint), which leaves every adjoint at the +0 it starts from
    constexpr int SL = HT - (NC - 1) * S;
    static_assert(SL >= 1 && SL <= S, "the last lane owns at least one step");
    constexpr int NOA = NO > 0 ? NO : 1;
    constexpr bool lane_feats = L > 0;
    constexpr int D = feat_dim(L);
    const ocd_scenario_desc &d = p.d;
    const int K = p.K;
    const int lane = threadIdx.x & 63;
    const int seg = lane / NC;
    const int c = lane - seg * NC;                             // chunk index: steps c*S .. c*S+S-1
    const bool first = c == 0, last = c == NC - 1;
    const int slot = seg / K;                                  // trajectory slot inside the wavefront
    const int kinit = seg - slot * K;

    const long long prob_raw = (long long)blockIdx.x * p.segs_used + slot;
    const bool live = (slot < p.segs_used) && (prob_raw < p.n_problems);
    const long long prob = live ? prob_raw : (p.n_problems - 1);   // parked lanes shadow a real problem
    const unsigned long long live_mask = __ballot(live);
    const unsigned long long last_mask = __ballot(last), first_mask = __ballot(first);
    const unsigned long long real_mask = live_mask & ~last_mask;        // the lanes whose padding steps (if any) count
#ifdef OCD_NO_ASM_CHAINS
    constexpr bool asm_chains = false, asm_bwd_vth = false;
#else
    constexpr bool asm_chains = chunk_chain_supported<S, NC - 1>::value;   // hand-scheduled rounds (ocd_chunk_chains.h)
    constexpr bool asm_bwd_vth = chunk_chain_supported<S, NC - 1>::bwd_vth;
#endif

    float dt = d.dt, fr = d.ego_friction;
    const float dt2 = d.dt_sq, lr = d.learning_rate;
    // the step length and the friction coefficient enter ~120 multiplications per pass: in the builds that share a SIMD they
    // live in vector registers -- with an SGPR source a multiplication is a half-rate instruction beside other wavefronts
    // (profiles/r03_issue_table.txt; config 5 whole -5 %); a wavefront alone on its SIMD issues every class at the same rate
    if constexpr (!LAT) asm volatile("v_mov_b32 %0, %2\nv_mov_b32 %1, %3" : "=v"(dt), "=v"(fr) : "s"(d.dt), "s"(d.ego_friction));

    // the work-item list of the shared-SIMD builds' gradient passes (see horizon_pass): ITEM_CAP items of 6 (one scripted
    // car: 8) operands -- four operands, a (weight, kind) pair that the item's two results overwrite, and the pair of
    // reciprocals; slot ITEM_ZERO holds (+0, +0) for the (lane, step) pairs without the feature, beyond the last round's
    // reach (ITEM_CAP - 1 + 63)
#ifdef OCD_NO_ITEMS
    constexpr bool use_items = false;
#else
    constexpr bool use_items = lane_feats && NO > 0 && !LAT;
#endif
    constexpr int ITEM_CAP = 64 * (S < 4 ? S : 4), ITEM_ZERO = ITEM_CAP + 64, ITEM_PAIR_BIT = 1 << 16;
    // a state inside BOTH cars' boxes as a pair of items: the builds with room for it (measured: replanning H = 15 at S = 3
    // -5.6 %; at S = 5 the extra paths cost the three-per-SIMD build of merging H = 25 +14 %: there such a step takes reward_state)
    constexpr bool pair_items = NO == 2 && S <= 3;
    __shared__ float4 item_a[use_items ? ITEM_ZERO + 1 : 1];                 // (x - cx | x, y - cy, wx, wy)
    __shared__ float2 item_b[use_items ? ITEM_ZERO + 1 : 1];                 // (weight, kind) -> the item's two results
    __shared__ float2 item_r[use_items && NO == 1 ? ITEM_ZERO + 1 : 1];      // one scripted car: the reciprocals of wx, wy
    if constexpr (use_items) {
        if (lane == 0) item_b[ITEM_ZERO] = float2{0.0f, 0.0f};
        __syncthreads();
    }

    // ---- problem inputs (as in mpc_kernel) ----
    float ex, ey, ev, eth;
    float ox[NOA], oy[NOA], ov[NOA], oth[NOA];
    float w[OCD_MAX_FEATURES];
    int tp_idx = 0;
    if (p.mode == OCD_MODE_ROLLOUT && !p.from_state) {
        long long p_, n_;
        const bool row_ok = episode_rows(p, prob, p_, n_, tp_idx);
        const float *ini = p.ego_states + 4 * n_;
        ex = row_ok ? ini[0] : __builtin_nanf(""); ey = ini[1]; ev = ini[2]; eth = ini[3];
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
        tp_idx = p.sample_fixed;
    }
    const LaneGradConst<L> lgc = lane_grad_const<L>(w, d);
    float wd[OCD_MAX_FEATURES];
#pragma unroll
    for (int k = 0; k < OCD_MAX_FEATURES; ++k) wd[k] = d.designer_weights[k];

    const int T = p.T;
    const PkConsts pkc = pk_consts();
    ScConsts scc;                              // coefficient pairs of the two-wide sin / cos polynomials (latency builds)
    if constexpr (LAT) scc = sc_consts();
    // The latency build wants its SIMD to itself, and the workgroup dispatcher does not promise that: in the diagnostic
    // build (make stamps; a third of the register file per wavefront) it put two of the 1 024 single-wavefront
    // workgroups on every tenth SIMD and none on as many others, and those wavefronts took 1.5x as long
    // (tools/stamp_profile.py prints the placement from HW_ID).  Claiming the accumulation registers -- never touched,
    // the clobber only raises the kernel's register allocation above half of the SIMD's 512 -- makes a second wavefront
    // on a SIMD impossible, so the dispatcher has to use every SIMD.  (The product build timed the same before and
    // after: it was being placed evenly; this makes that a property of the kernel instead of luck.)
    if constexpr (LAT) asm volatile("" ::: "a255");
    OCD_STAMP_DECL
    float G_ret = 0.0f;
    const BumpGeom bg0 = {0.0f, 1.0f, 0.0f, 1.0f};
    const bool writer = live && kinit == 0 && first;

    if (p.mode == OCD_MODE_ROLLOUT && p.traj_out && writer) {
        float *tr = p.traj_out + (size_t)prob * (T + 1) * (NO + 1) * 4;
        tr[0] = ex; tr[1] = ey; tr[2] = ev; tr[3] = eth;
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            tr[4 * (j + 1)] = ox[j]; tr[4 * (j + 1) + 1] = oy[j]; tr[4 * (j + 1) + 2] = ov[j]; tr[4 * (j + 1) + 3] = oth[j];
        }
    }

    for (int step = 0; step < T; ++step) {
        if (p.mode == OCD_MODE_ROLLOUT) {
            if (d.teleport_step > 0 && (p.t0 + step + 1) == d.teleport_step) {
                const int car = d.teleport_car[tp_idx & (OCD_MAX_SAMPLES - 1)];
#pragma unroll
                for (int j = 0; j < NO; ++j) {
                    if (car == j + 1) {
                        ox[j] = d.teleport_state[0]; oy[j] = d.teleport_state[1];
                        ov[j] = d.teleport_state[2]; oth[j] = d.teleport_state[3];
                    }
                }
            }
            BumpGeom bgd[NOA];
            bgd[0] = bg0;
#pragma unroll
            for (int j = 0; j < NO; ++j) bgd[j] = bump_geom(ox[j], oy[j], d.bump_half_x, d.bump_half_y);
            float s_, c_;
            sincos_(eth, s_, c_);
            Q4 qd;
            const float r = reward_state<NO, L, false, true>(d, wd, ex, ey, ev, s_, c_, bgd, qd, nullptr);
            G_ret = G_ret + r;
        }

        // ---- planner's model of the scripted cars at this lane's S steps (naive_planner.py:51-66) ----
        BumpGeom bg[S][NOA];
        float wx1[S][NOA], wy1[S][NOA];
#pragma unroll
        for (int s = 0; s < S; ++s) bg[s][0] = bg0;
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            float px = ox[j], py = oy[j], pv = ov[j], pth = oth[j];
            float cap_x[S], cap_y[S];
#pragma unroll
            for (int s = 0; s < S; ++s) { cap_x[s] = px; cap_y[s] = py; }
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
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        cap_x[s] = (tt == c * S + s) ? px : cap_x[s];
                        cap_y[s] = (tt == c * S + s) ? py : cap_y[s];
                    }
                }
            } else {
                float s_, c_;
                sincos_(pth, s_, c_);
                const float incx = (c_ * pv) * dt, incy = (s_ * pv) * dt;
                for (int tt = 0; tt < H; ++tt) {
                    px = px + incx;
                    py = py + incy;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        cap_x[s] = (tt == c * S + s) ? px : cap_x[s];
                        cap_y[s] = (tt == c * S + s) ? py : cap_y[s];
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < S; ++s) bg[s][j] = bump_geom(cap_x[s], cap_y[s], d.bump_half_x, d.bump_half_y);
        }
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int j = 0; j < NOA; ++j) { wx1[s][j] = bg[s][j].wx * 1.001f; wy1[s][j] = bg[s][j].wy * 1.001f; }
        // one scripted car: the refined reciprocals of its half-widths in this control step (reward_one's FASTDIV form divides by
        // them in every pass) and the lanes whose widths are outside its guard
        BumpRecip br[S][NOA];
        unsigned long long widths_beyond = 0ull, widths_degenerate = 0ull, zn_unguarded = 0ull;
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int j = 0; j < NOA; ++j) {
                br[s][j] = BumpRecip{0.0f, 0.0f};
                if constexpr (lane_feats && NO > 0)
                    widths_degenerate |= __builtin_amdgcn_ballot_w64(bump_widths_degenerate(bg[s][j])) & ((s >= SL) ? real_mask : live_mask);
                if constexpr (lane_feats && NO == 1) {
                    br[s][j].rx = refined_recip(bg[s][j].wx);
                    br[s][j].ry = refined_recip(bg[s][j].wy);
                    widths_beyond |= __builtin_amdgcn_ballot_w64(!bump_widths_guarded(bg[s][j]));
                }
                if constexpr (lane_feats && NO == 2 && LAT) {   // reward_two's reciprocal quotients (round 6)
                    br[s][j].rx = refined_recip(bg[s][j].wx);
                    br[s][j].ry = refined_recip(bg[s][j].wy);
                    zn_unguarded |= __builtin_amdgcn_ballot_w64(!bump_widths_guarded(bg[s][j]));
                }
            }
        // a degenerate width (ocd_device.h: bump_widths_degenerate) sends every pass of this control step to the evaluation of
        // every feature: through `beyond` in the straight-line builds, as the "no_feature_skips" knob does in the others
        widths_beyond |= widths_degenerate;
        // x_hi = 0 (descriptor outside LaneGradConst's conditions): no shortened division on any lane (see ocd_kernels.hip)
        if constexpr (lane_feats) widths_beyond |= (lgc.x_hi > 0.0f) ? 0ull : ~0ull;
        const bool full_step = p.no_skips || widths_degenerate != 0ull;

        // ---- control initialisation of this segment (naive_planner.py:107-116) ----
        float s0, c0;
        sincos_(eth, s0, c0);
        // extra_inits coast at friction * self.car.state[2] ** 2 (naive_planner.py:114): the car's own speed, which
        // a caller planning from a foreign init_state passes apart (ocd_plan_batch_from); else the state's ego speed
        const float v_car = (p.init_speed != nullptr) ? p.init_speed[prob] : ev;
        const float a_coast = fr * (v_car * v_car);
        const int k3 = kinit % 3;
        float ua[S], uw[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            ua[s] = (kinit >= 3) ? a_coast : 0.0f;
            uw[s] = (k3 == 0) ? 0.0f : ((k3 == 1) ? -0.65f : 0.65f);
        }
        float loss = 0.0f;

        auto horizon_pass = [&](auto grad_tag) __attribute__((always_inline)) {
            constexpr bool GRAD = decltype(grad_tag)::value;
            OCD_STAMP(0);                                  // everything outside the passes
            // ===== forward =====
            float a_c[S], wdt[S];
            bool pass_a[S], pass_w[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float a1 = min_tf(ua[s], 4.0f);
                a_c[s] = max_tf(a1, -8.0f);
                const float w1 = min_tf(uw[s], 4.0f);
                const float w_c = max_tf(w1, -4.0f);
                pass_a[s] = (ua[s] <= 4.0f) && (a1 >= -8.0f);
                pass_w[s] = (uw[s] <= 4.0f) && (w1 >= -4.0f);
                wdt[s] = w_c * dt;
            }
            // speed / heading at the start of the chunk: NC-1 rounds of "walk my S steps, hand the end to the lane above"
            float vs = ev, ths = eth;
            if constexpr (asm_chains) {
                chunk_fwd_vth<S, NC - 1, !LAT>(vs, ths, ev, eth, a_c, wdt, fr, dt, first_mask);
            } else {
#pragma unroll
                for (int r = 0; r < NC - 1; ++r) {
                    float v = vs, th = ths;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        v = v + (a_c[s] - fr * (v * v)) * dt;
                        th = th + wdt[s];
                    }
                    const float vb = wave_below(v), tb = wave_below(th);
                    vs = first ? ev : vb;
                    ths = first ? eth : tb;
                }
            }
            OCD_STAMP(1);                                  // clip, speed / heading recurrence
            // the lane's own S steps
            float vpre[S], dd[S], vn[S], sn[S], cn[S];
            {
                float v = vs, th = ths;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float v2 = v * v;
                    const float fv2 = fr * v2;
                    const float acc = a_c[s] - fv2;
                    const float vdt = v * dt;
                    const float hA = 0.5f * acc;
                    const float hAdt2 = hA * dt2;
                    vpre[s] = v;
                    dd[s] = vdt + hAdt2;
                    v = v + acc * dt;
                    th = th + wdt[s];
                    vn[s] = v;
                    // latency builds: two-wide polynomial chains + bit-select quadrant swap (-1.1 ... -1.4 % at the per-GPU
                    // shares of configs 4 / 5); the throughput builds keep the scalar form: the coefficient pairs
                    // cost 8 VGPRs, which the three-per-SIMD builds pay in scratch (config 5 whole +8 %, measured)
                    if constexpr (LAT) sincos_pk(th, sn[s], cn[s], scc);
                    else sincos_(th, sn[s], cn[s]);
                }
                if constexpr (SL < S) {
#pragma unroll
                    for (int s = SL; s < S; ++s) { dd[s] = last ? 0.0f : dd[s]; vpre[s] = last ? 0.0f : vpre[s]; }
                }
            }
            // sin / cos of the heading BEFORE each step: the previous step's, across the lane boundary for s = 0
            float s_pre[S], c_pre[S];
            {
                const float sb = wave_below(sn[S - 1]), cb = wave_below(cn[S - 1]);
                s_pre[0] = first ? s0 : sb;
                c_pre[0] = first ? c0 : cb;
#pragma unroll
                for (int s = 1; s < S; ++s) { s_pre[s] = sn[s - 1]; c_pre[s] = cn[s - 1]; }
                if constexpr (SL < S) {                        // padding steps: an all-zero tape entry
#pragma unroll
                    for (int s = SL; s < S; ++s) { s_pre[s] = last ? 0.0f : s_pre[s]; c_pre[s] = last ? 0.0f : c_pre[s]; }
                }
            }
            float cd[S], sd[S];
#pragma unroll
            for (int s = 0; s < S; ++s) { cd[s] = c_pre[s] * dd[s]; sd[s] = s_pre[s] * dd[s]; }
            OCD_STAMP(2);                                  // own steps, sincos
            // position at the start of the chunk
            float xs = ex, ys = ey;
            if constexpr (asm_chains) {
                chunk_fwd_xy<S, NC - 1>(xs, ys, ex, ey, cd, sd, first_mask);
            } else {
#pragma unroll
                for (int r = 0; r < NC - 1; ++r) {
                    float x = xs, y = ys;
#pragma unroll
                    for (int s = 0; s < S; ++s) { x = x + cd[s]; y = y + sd[s]; }
                    const float xb = wave_below(x), yb = wave_below(y);
                    xs = first ? ex : xb;
                    ys = first ? ey : yb;
                }
            }

            OCD_STAMP(3);                                  // position recurrence
            // ===== reward features at the S post-step states =====
            Q4 q[S];
            float rw[S];
            if constexpr (lane_feats && LAT && !(use_items && GRAD)) {
                // per step ONE test chooses between one feature per lane (straight line, no sub-skips) and the rarer
                // multi-feature evaluations; no "none active" path.  (Hoisting the masks of all S steps in front of
                // the evaluations hides the ballot latency but keeps 5 x (1 + NO) lane masks alive: SGPR spills.)
                float x = xs, y = ys;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    x = x + cd[s];
                    y = y + sd[s];
                    const float xn = x, yn = y;
                    bool nc[NOA];
                    nc[0] = false;
                    const bool nf = needs_fence(d, xn);
                    const unsigned long long lm = (s >= SL) ? real_mask : live_mask;
                    const unsigned long long mf = __builtin_amdgcn_ballot_w64(nf) & lm;
                    unsigned long long mc_any = 0ull, multi_f = 0ull, multi_c = 0ull, tiny_n = 0ull, triple = 0ull;
#pragma unroll
                    for (int j = 0; j < NO; ++j) {
                        const float dx = xn - bg[s][j].cx, dy = yn - bg[s][j].cy;
                        const bool ncx = __builtin_fabsf(dx) < wx1[s][j], ncy = __builtin_fabsf(dy) < wy1[s][j];
                        nc[j] = ncx && ncy;
                        const unsigned long long mj = __builtin_amdgcn_ballot_w64(ncx) & __builtin_amdgcn_ballot_w64(ncy) & lm;
                        triple |= (mj & mc_any & mf);      // fence and both cars on one lane
                        multi_f |= (mj & mf);
                        multi_c |= (mj & mc_any);
                        mc_any |= mj;
                        if constexpr (NO == 1)             // a zero / tiny numerator of the shortened (x - cx) / wx
                            tiny_n |= __builtin_amdgcn_ballot_w64(__builtin_fabsf(dx) < 7.888609052210118e-31f) |
                                  __builtin_amdgcn_ballot_w64(__builtin_fabsf(dy) < 7.888609052210118e-31f);
                    }
                    // beyond the guards of the shortened divisions (LaneGradConst::x_hi, quot2_by_recip): the full ones
                    const unsigned long long beyond = (__builtin_amdgcn_ballot_w64(!(__builtin_fabsf(xn) < lgc.x_hi)) & mf) |
                                                      ((tiny_n | widths_beyond) & lm);
                    OCD_STAMP(4);                          // choice of the evaluation
                    if (__builtin_expect((multi_f | multi_c | beyond) != 0ull, 0)) {
                        if (__builtin_expect(beyond != 0ull, 0)) {
                            if ((multi_c | widths_degenerate) != 0ull)
                                rw[s] = reward_every<NO, L, GRAD>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], q[s], pkc, &lgc, lm);
                            else if (multi_f != 0ull)
                                rw[s] = reward_fc<NO, L, GRAD>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], nc, q[s], pkc, &lgc, lm);
                            else
                                rw[s] = reward_one<NO, L, GRAD, false>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], br[s], nc, nf, true,
                                                                       true, q[s], pkc, lgc, lm);
                        } else if (multi_c != 0ull) {
                            // a lane inside both boxes: two unit pairs (reward_two, round 6) unless some lane has the fence as
                            // well, a car's width is outside the reciprocal quotients' guard or a numerator is zero / tiny
                            if constexpr (NO == 2 && GRAD) {
                                const float tiny = 7.888609052210118e-31f;         // (tested here: only these steps pay for it)
                                const unsigned long long zn_tiny =
                                    (__builtin_amdgcn_ballot_w64(__builtin_fabsf(xn - bg[s][0].cx) < tiny) | __builtin_amdgcn_ballot_w64(__builtin_fabsf(yn - bg[s][0].cy) < tiny) |
                                     __builtin_amdgcn_ballot_w64(__builtin_fabsf(xn - bg[s][NOA - 1].cx) < tiny) | __builtin_amdgcn_ballot_w64(__builtin_fabsf(yn - bg[s][NOA - 1].cy) < tiny)) & lm;
                                if (__builtin_expect((triple | zn_unguarded | zn_tiny) != 0ull, 0)) {
                                    rw[s] = reward_every<NO, L, GRAD, GRAD>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], q[s], pkc, &lgc, lm);
                                } else {
                                    rw[s] = 0.0f;
                                    reward_two<NO, L, true, true>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], br[s], nc, q[s], pkc, lgc, lm);
                                }
                            } else {
                                rw[s] = reward_every<NO, L, GRAD, GRAD>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], q[s], pkc, &lgc, lm);
                            }
                        } else
                            rw[s] = reward_fc<NO, L, GRAD, GRAD>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], nc, q[s], pkc, &lgc, lm);
                        OCD_STAMP(5); OCD_STAMP_COUNT(12);
                    } else {
                        rw[s] = reward_one<NO, L, GRAD, false, false, GRAD>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], br[s], nc, nf, true,
                                                                            true, q[s], pkc, lgc, lm);
                        OCD_STAMP(6); OCD_STAMP_COUNT(13);
                    }
                    if constexpr (GRAD) {
                        if (s >= SL) { q[s].qx = last ? 0.0f : q[s].qx; q[s].qy = last ? 0.0f : q[s].qy; q[s].qv = last ? 0.0f : q[s].qv; q[s].qth = last ? 0.0f : q[s].qth; }
                    }
                }
            } else if constexpr (use_items && GRAD) {
                // Active features as work items (ocd_device.h: reward_base_grad / feature_item_grad): every (lane, step) pair
                // gets the features every state has; its active fence / collision terms go to a list in LDS, are evaluated
                // 64 at a time and come back as two adjoint terms each.  A state inside BOTH cars' boxes appends its two
                // collision items side by side at an even slot (the evaluation compares the two lanes' products; pair_items
                // builds -- in the others such a step goes the way of the next sentence).  A step beyond the guards of the
                // shortened divisions, with a degenerate car width, under the diagnostics knobs or beyond the list's capacity
                // evaluates every feature of every lane (reward_state), as before.
                float x = xs, y = ys;
                int n_items = 0;                               // wave-uniform
                bool any_pair = false;                         // wave-uniform: some state of this pass is inside both boxes
                int slot_c[S], slot_f[S];                      // (slot_c of such a state: its first item | ITEM_PAIR_BIT)
                const float w_col = w[L + 2], w_f = w[L + 3];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    x = x + cd[s];
                    y = y + sd[s];
                    const float xn = x, yn = y;
                    const bool nf = needs_fence(d, xn);
                    const unsigned long long lm = (s >= SL) ? real_mask : live_mask;
                    const unsigned long long mf = __ballot(nf) & lm;
                    unsigned long long mc_any = 0ull, multi_c = 0ull, tiny_n = 0ull;
                    float idx = 0.0f, idy = 0.0f, iwx = 1.0f, iwy = 1.0f, irx = 1.0f, iry = 1.0f;
                    float dxl[NOA], dyl[NOA];
                    const bool in_lm = (s >= SL) ? (live && !last) : live;        // this lane's bit of lm
                    bool nc_any = false, nc_both = false;
#pragma unroll
                    for (int j = 0; j < NO; ++j) {
                        const float dx = xn - bg[s][j].cx, dy = yn - bg[s][j].cy;
                        const bool ncj = (__builtin_fabsf(dx) < wx1[s][j]) && (__builtin_fabsf(dy) < wy1[s][j]);
                        const unsigned long long mj = __ballot(ncj) & lm;
                        multi_c |= (mj & mc_any);
                        mc_any |= mj;
                        nc_both = nc_both || (nc_any && ncj);
                        nc_any = nc_any || ncj;
                        dxl[j] = dx; dyl[j] = dy;
                        if (j == 0) { idx = dx; idy = dy; iwx = bg[s][0].wx; iwy = bg[s][0].wy; irx = br[s][0].rx; iry = br[s][0].ry; }
                        else { idx = ncj ? dx : idx; idy = ncj ? dy : idy; iwx = ncj ? bg[s][j].wx : iwx; iwy = ncj ? bg[s][j].wy : iwy; }
                        if constexpr (NO == 1)                 // a zero / tiny numerator of the shortened (x - cx) / wx
                            tiny_n |= __ballot(__builtin_fabsf(dx) < 7.888609052210118e-31f) | __ballot(__builtin_fabsf(dy) < 7.888609052210118e-31f);
                    }
                    const unsigned long long beyond = (__ballot(!(__builtin_fabsf(xn) < lgc.x_hi)) & mf) | (((tiny_n & mc_any) | widths_beyond) & lm);
                    const int n_add = __popcll(mf) + __popcll(mc_any) + (pair_items ? __popcll(multi_c) + 1 : 0);
                    if (full_step || p.no_unify || beyond != 0ull || (!pair_items && multi_c != 0ull) || n_items + n_add > ITEM_CAP) {
                        OCD_STAMP(4);
                        rw[s] = reward_state<NO, L, GRAD>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], q[s], nullptr, true, true);
                        slot_c[s] = ITEM_ZERO; slot_f[s] = ITEM_ZERO;
                        OCD_STAMP(5); OCD_STAMP_COUNT(12);
                    } else {
                        rw[s] = 0.0f;
                        reward_base_grad<L>(d, w, xn, vn[s], sn[s], cn[s], q[s], lgc, lm);
                        if (pair_items && __builtin_expect(multi_c != 0ull, 0)) {
                            // some state is inside both boxes: the single items first, then the pairs from an even slot
                            const unsigned long long ms = mc_any & ~multi_c;
                            const bool ac = nc_any && !nc_both && in_lm;
                            const int ic = n_items + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ms >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ms, 0u));
                            if (ac) { item_a[ic] = float4{idx, idy, iwx, iwy}; item_b[ic] = float2{w_col, 0.0f}; }
                            n_items += __popcll(ms);
                            n_items += n_items & 1;
                            const bool ad = nc_both && in_lm;
                            const int id = n_items + 2 * (int)__builtin_amdgcn_mbcnt_hi((unsigned)(multi_c >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)multi_c, 0u));
                            if (ad) {
#pragma unroll
                                for (int j = 0; j < (pair_items ? 2 : 0); ++j) {
                                    item_a[id + j] = float4{dxl[j], dyl[j], bg[s][j].wx, bg[s][j].wy};
                                    item_b[id + j] = float2{w_col, 2.0f};
                                }
                            }
                            n_items += 2 * __popcll(multi_c);
                            slot_c[s] = ad ? (id | ITEM_PAIR_BIT) : (ac ? ic : ITEM_ZERO);
                            any_pair = true;
                        } else {
                            const bool ac = nc_any && in_lm;
                            const int ic = n_items + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mc_any >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mc_any, 0u));
                            slot_c[s] = ac ? ic : ITEM_ZERO;
                            if (ac) {
                                item_a[ic] = float4{idx, idy, iwx, iwy}; item_b[ic] = float2{w_col, 0.0f};
                                if constexpr (NO == 1) item_r[ic] = float2{irx, iry};
                            }
                            n_items += __popcll(mc_any);
                        }
                        const bool af = nf && in_lm;
                        const int jf = n_items + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mf >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mf, 0u));
                        slot_f[s] = af ? jf : ITEM_ZERO;
                        if (af) { item_a[jf].x = xn; item_b[jf] = float2{w_f, 1.0f}; }
                        n_items += __popcll(mf);
                        OCD_STAMP(4);
                    }
                }
                __syncthreads();                               // (one wavefront per workgroup: orders the LDS accesses)
                for (int base = 0; base < n_items; base += 64) {
                    const int i = base + lane;
                    float o1, o2;
                    const float4 ia4 = item_a[i];
                    const float2 ib2 = item_b[i];
                    const float ia = ia4.x, idy_ = ia4.y, iwx_ = ia4.z, iwy_ = ia4.w, iws = ib2.x, ity = ib2.y;
                    float irx_ = 1.0f, iry_ = 1.0f;
                    if constexpr (NO == 1) { const float2 ir2 = item_r[i]; irx_ = ir2.x; iry_ = ir2.y; }
                    if (pair_items && __builtin_expect(any_pair, 0))
                        feature_item_grad<NO, NO == 1>(d, ity == 1.0f, ity == 2.0f, ia, idy_, iwx_, iwy_, irx_, iry_, iws, pkc, o1, o2);
                    else
                        feature_item_grad<NO, NO == 1>(d, ity == 1.0f, false, ia, idy_, iwx_, iwy_, irx_, iry_, iws, pkc, o1, o2);
                    item_b[i] = float2{o1, o2};
                    OCD_STAMP_COUNT(13);
                }
                __syncthreads();
                OCD_STAMP(6);
                if (pair_items && __builtin_expect(any_pair, 0)) {
                    // reduce_max's gradient of a state inside both boxes: car 0's terms, then car 1's (reward_state's order)
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const bool pr = (slot_c[s] & ITEM_PAIR_BIT) != 0;
                        const int sc = slot_c[s] & (ITEM_PAIR_BIT - 1), sd2 = pr ? sc + 1 : ITEM_ZERO;
                        const float2 rc = item_b[sc], rd = item_b[sd2], rf = item_b[slot_f[s]];
                        q[s].qx = (((q[s].qx + rc.x) + rd.x) + rf.x) + rf.y;
                        q[s].qy = (q[s].qy + rc.y) + rd.y;
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float2 rc = item_b[slot_c[s]], rf = item_b[slot_f[s]];
                        q[s].qx = ((q[s].qx + rc.x) + rf.x) + rf.y;
                        q[s].qy = q[s].qy + rc.y;
                    }
                }
                if constexpr (SL < S) {
#pragma unroll
                    for (int s = SL; s < S; ++s) { q[s].qx = last ? 0.0f : q[s].qx; q[s].qy = last ? 0.0f : q[s].qy; q[s].qv = last ? 0.0f : q[s].qv; q[s].qth = last ? 0.0f : q[s].qth; }
                }
                OCD_STAMP(7);
            } else {
                float x = xs, y = ys;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    x = x + cd[s];
                    y = y + sd[s];
                    const float xn = x, yn = y;
                    if constexpr (lane_feats) {
                        bool nc[NOA];
                        nc[0] = false;
                        const bool nf = needs_fence(d, xn);
                        const unsigned long long lm = (s >= SL) ? real_mask : live_mask;
                        unsigned long long mf = __ballot(nf) & lm, mc_any = 0ull, multi_f = 0ull, multi_c = 0ull;
#pragma unroll
                        for (int j = 0; j < NO; ++j) {
                            const float dx = xn - bg[s][j].cx, dy = yn - bg[s][j].cy;
                            nc[j] = (__builtin_fabsf(dx) < wx1[s][j]) && (__builtin_fabsf(dy) < wy1[s][j]);
                            const unsigned long long mj = __ballot(nc[j]) & lm;
                            multi_f |= (mj & mf);          // fence and a car on one lane
                            multi_c |= (mj & mc_any);      // two cars on one lane
                            mc_any |= mj;
                        }
                        const bool has_f = mf != 0ull, has_col = mc_any != 0ull;
                        OCD_STAMP(4);                      // choice of the evaluation
                        if (full_step || multi_c != 0ull || (p.no_unify && (has_f || has_col))) {
                            // (reward_every's packed form needs more registers than these builds have to spare)
                            rw[s] = reward_state<NO, L, GRAD>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], q[s], nullptr, true, true);
                            OCD_STAMP(5); OCD_STAMP_COUNT(12);
                        } else if (multi_f != 0ull) {
                            rw[s] = reward_fc<NO, L, GRAD>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], nc, q[s], pkc);
                            OCD_STAMP(5); OCD_STAMP_COUNT(12);
                        } else if (has_f || has_col) {
                            // the shortened reciprocals (ocd_devmath.h: recip_pair_guarded) unless a fence lane is beyond
                            // their guard (LaneGradConst::x_hi); they need fewer registers than the full divisions
                            // (one scripted car: also (x - cx) / wx by the reciprocals of this control step, quot2_by_recip)
                            constexpr bool ZN1 = GRAD && NO == 1;
                            unsigned long long tiny_n = 0ull;
                            if constexpr (ZN1)
                                tiny_n = (__ballot(__builtin_fabsf(xn - bg[s][0].cx) < 7.888609052210118e-31f) |
                                          __ballot(__builtin_fabsf(yn - bg[s][0].cy) < 7.888609052210118e-31f) | widths_beyond) & lm;
                            if (!GRAD || ((__ballot(!(__builtin_fabsf(xn) < lgc.x_hi)) & mf) | tiny_n) != 0ull)
                                rw[s] = reward_one<NO, L, GRAD>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], br[s], nc, nf, has_col, has_f, q[s], pkc, lgc, lm);
                            else
                                rw[s] = reward_one<NO, L, GRAD, true, false, GRAD, ZN1>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], br[s], nc, nf, has_col, has_f, q[s], pkc, lgc, lm);
                            OCD_STAMP(6); OCD_STAMP_COUNT(13);
                            if (has_col) OCD_STAMP_COUNT(11);
                            if (has_f) OCD_STAMP_COUNT(15);
                        } else {
                            rw[s] = reward_state<NO, L, GRAD>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], q[s], nullptr, false, false);
                            OCD_STAMP(7); OCD_STAMP_COUNT(14);
                        }
                    } else {
                        rw[s] = reward_state<NO, L, GRAD>(d, w, xn, yn, vn[s], sn[s], cn[s], bg[s], q[s], nullptr);
                    }
                    if constexpr (GRAD) {
                        if (s >= SL) { q[s].qx = last ? 0.0f : q[s].qx; q[s].qy = last ? 0.0f : q[s].qy; q[s].qv = last ? 0.0f : q[s].qv; q[s].qth = last ? 0.0f : q[s].qth; }
                    }
                }
            }

            if constexpr (!GRAD) {
                // ===== objective only: r = 0; r += reward, t = 0..H-1, in that order =====
                float Rs = 0.0f;                               // partial sum at the start of the chunk
                float R = 0.0f;
#pragma unroll
                for (int r = 0; r < NC; ++r) {
                    R = Rs;
#pragma unroll
                    for (int s = 0; s < S; ++s) R = (s >= SL && last) ? R : (R + rw[s]);
                    if (r < NC - 1) {
                        const float Rb = wave_below(R);
                        Rs = first ? 0.0f : Rb;
                    }
                }
                loss = -R;                                     // complete in the segment's last lane
            } else {
                // ===== backward =====
                // position adjoint arriving at the END of the chunk (from the later chunks)
                float LxE = 0.0f, LyE = 0.0f;
                if constexpr (asm_chains) {
                    float qxa[S], qya[S];
#pragma unroll
                    for (int s = 0; s < S; ++s) { qxa[s] = q[s].qx; qya[s] = q[s].qy; }
                    chunk_bwd_xy<S, NC - 1>(LxE, LyE, qxa, qya, last_mask);
                } else {
#pragma unroll
                    for (int r = 0; r < NC - 1; ++r) {
                        float Lx = LxE, Ly = LyE;
#pragma unroll
                        for (int s = S - 1; s >= 0; --s) { Lx = q[s].qx + Lx; Ly = q[s].qy + Ly; }
                        const float xa = wave_above(Lx), ya = wave_above(Ly);
                        LxE = last ? 0.0f : xa;
                        LyE = last ? 0.0f : ya;
                    }
                }
                OCD_STAMP(8);                              // position adjoint recurrence
                float tau[S], gv1[S], gA1[S];
                {
                    float Lx = LxE, Ly = LyE;
#pragma unroll
                    for (int s = S - 1; s >= 0; --s) {
                        const float Ax = q[s].qx + Lx;
                        const float Ay = q[s].qy + Ly;
                        const float g_c = Ax * dd[s];
                        const float g_s = Ay * dd[s];
                        const float g_d = Ax * c_pre[s] + Ay * s_pre[s];
                        tau[s] = (-g_c) * s_pre[s] + g_s * c_pre[s];
                        gv1[s] = g_d * dt;
                        gA1[s] = (g_d * dt2) * 0.5f;
                        Lx = Ax; Ly = Ay;
                    }
                }
                // speed / heading adjoint arriving at the end of the chunk
                float LvE = 0.0f, LthE = 0.0f;
                if constexpr (asm_bwd_vth) {
                    float qva[S], qtha[S];
#pragma unroll
                    for (int s = 0; s < S; ++s) { qva[s] = q[s].qv; qtha[s] = q[s].qth; }
                    chunk_bwd_vth<S, NC - 1, !LAT>(LvE, LthE, qva, qtha, gA1, gv1, vpre, tau, fr, dt, last_mask);
                } else {
#pragma unroll
                    for (int r = 0; r < NC - 1; ++r) {
                        float Lv = LvE, Lth = LthE;
#pragma unroll
                        for (int s = S - 1; s >= 0; --s) {
                            const float Av_ = q[s].qv + Lv;
                            const float gA_ = gA1[s] + Av_ * dt;
                            const float gv2_ = (-gA_) * fr;
                            const float gv3_ = (gv2_ * 2.0f) * vpre[s];
                            Lv = (gv1[s] + Av_) + gv3_;
                            Lth = (q[s].qth + Lth) + tau[s];
                        }
                        const float va = wave_above(Lv), ta = wave_above(Lth);
                        LvE = last ? 0.0f : va;
                        LthE = last ? 0.0f : ta;
                    }
                }
                OCD_STAMP(9);                              // Jacobian products, speed / heading adjoint recurrence
                {
                    float Lv = LvE, Lth = LthE;
#pragma unroll
                    for (int s = S - 1; s >= 0; --s) {
                        const float Av = q[s].qv + Lv;
                        const float gA = gA1[s] + Av * dt;
                        const float Ath = q[s].qth + Lth;
                        const float grad_a = pass_a[s] ? gA : 0.0f;
                        const float grad_w = pass_w[s] ? (Ath * dt) : 0.0f;
                        const float gv2_ = (-gA) * fr;
                        const float gv3_ = (gv2_ * 2.0f) * vpre[s];
                        Lv = (gv1[s] + Av) + gv3_;
                        Lth = Ath + tau[s];
                        // SGD on loss = -R:  u <- u + lr * dR/du
                        ua[s] = ua[s] + lr * grad_a;
                        uw[s] = uw[s] + lr * grad_w;
                    }
                }
                OCD_STAMP(10);                             // own steps' adjoint, control update
            }
        };

        const int n_iter = d.n_iter;
        for (int it = 0; it < n_iter; ++it) horizon_pass(bool_c<true>{});
        horizon_pass(bool_c<false>{});

        // ---- per-initialisation outputs (plan mode, parity tests) ----
        if (p.mode == OCD_MODE_PLAN && live) {
            if (p.all_plans_out) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if (c * S + s >= H) continue;
                    float *o = p.all_plans_out + (((size_t)prob * K + kinit) * H + (c * S + s)) * 2;
                    o[0] = ua[s]; o[1] = uw[s];
                }
            }
            if (p.all_losses_out && last) p.all_losses_out[(size_t)prob * K + kinit] = loss;
        }

        // ---- first-index argmin over the K initialisations (naive_planner.py:161-162) ----
        int best = 0;
        const int base = slot * K * NC;                          // first lane of this trajectory's K segments
        float bl = lane_read(loss, base + NC - 1);
        for (int k = 1; k < K; ++k) {
            const float lk = lane_read(loss, base + k * NC + NC - 1);
            if (lk < bl) { bl = lk; best = k; }
        }
        const float ca = lane_read(ua[0], base + best * NC);
        const float cw = lane_read(uw[0], base + best * NC);

        if (p.mode == OCD_MODE_PLAN) {
            if (live && kinit == best) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if (c * S + s >= H) continue;
                    float *o = p.plans_out + ((size_t)prob * H + (c * S + s)) * 2;
                    o[0] = ua[s]; o[1] = uw[s];
                }
                if (first) {
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
                const int gstep = p.t0 + step;
                const bool in_plan = gstep < d.other_plan_len[j];
                const float u0 = in_plan ? d.other_plan[j][gstep & (OCD_MAX_PLAN - 1)][0] : d.other_default[j][0];
                const float u1 = in_plan ? d.other_plan[j][gstep & (OCD_MAX_PLAN - 1)][1] : d.other_default[j][1];
                float s_, c_;
                sincos_(oth[j], s_, c_);
                dyn_step(ox[j], oy[j], ov[j], oth[j], c_, s_, u0, u1, dt, dt2, d.other_friction[j], nx, ny, nv, nth);
                ox[j] = nx; oy[j] = ny; ov[j] = nv; oth[j] = nth;
            }
            if (writer) {
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
    if (p.mode == OCD_MODE_ROLLOUT && writer) p.returns_out[prob] = G_ret;
#ifdef OCD_STAMPS
    OCD_STAMP_LAST;
    if constexpr (LAT) {                                       // (slot 14 counts a path the latency build does not have)
        unsigned hw, xcc;                                      // where this wavefront ran: HW_ID | XCC_ID << 32
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st_acc[14] = (unsigned long long)hw | ((unsigned long long)(xcc & 0xf) << 32) | (1ull << 40);
    }
    if (p.debug && lane == 0)
        for (int i = 0; i < 16; ++i) p.debug[(size_t)blockIdx.x * 16 + i] = st_acc[i];
#endif
}

// ---------------------------------------------------------------- launch
template <int HT, int NO, int L, int S>
static hipError_t launch_chunk(const KernelParams &p_in, hipStream_t st)
{
    KernelParams p = p_in;
    constexpr int NC = (HT + S - 1) / S;
    const int cap = 64 / (p.K * NC);                            // trajectories per wavefront
    if (cap < 1) return hipErrorInvalidConfiguration;
    const long long simds = 4ll * (p.n_cus > 0 ? p.n_cus : 256);
    // spread over the SIMDs first (as V_SEG does): a batch that fills fewer than one wavefront per SIMD at full
    // packing runs with fewer trajectories per wavefront on more SIMDs
    long long want = (p.n_problems + simds - 1) / simds;
    int segs = p.segs_used > 0 ? p.segs_used : (int)(want < 1 ? 1 : (want > cap ? cap : want));
    segs = segs < 1 ? 1 : (segs > cap ? cap : segs);
    p.segs_used = segs;
    const unsigned blocks = (unsigned)((p.n_problems + segs - 1) / segs);
    const bool lat = L > 0 && NO > 0 && !p.no_skips && !p.no_unify && !p.no_latency_build && (long long)blocks <= simds;
    // (the unconstrained build of S > 2 holds two wavefronts on a SIMD: beyond two per SIMD the three-per-SIMD build keeps
    //  them all resident; S <= 2: three resident unconstrained, four in the OCC3 build)
    const bool occ3 = !lat && (long long)blocks > (S > 2 ? 2 : 3) * simds - (S > 2 ? 0 : 1) && !p.no_latency_build;
    note_launch(p, 4, S, segs, blocks, lat ? 1 : (occ3 ? 3 : 0), HT, 0, 1);
    if (lat) OCD_LAUNCH((mpc_chunk_kernel<HT, NO, L, S, true>), dim3(blocks), dim3(64), 0, st, p);
    else if (occ3) OCD_LAUNCH((mpc_chunk_kernel<HT, NO, L, S, false, true>), dim3(blocks), dim3(64), 0, st, p);
    else OCD_LAUNCH((mpc_chunk_kernel<HT, NO, L, S>), dim3(blocks), dim3(64), 0, st, p);
    return launch_status(p);
}

// Cost of chunk size S for n trajectories (round-3 sweeps, tools/sweep_sizes.sh, profiles/r03_sweep_sizes.txt):
//   instructions per pass of a wavefront ~ A(H) + B(NO) * S  (A: the recurrence rounds and the per-pass fixed part,
//   B: the lane-parallel work of one step), measured 430 / 730 / 1 330 at H = 10 / 15 / 25 and 350 / 550 per step with
//   one / two scripted cars;
//   a SIMD with w wavefronts needs w / g(w) times a lone wavefront's time, g = 1, 1.19, 1.26, 1.31 ... 1.41 at 8 (the
//   issue table of DESIGN.md section 4); a launch lasts as long as its busiest SIMD: w = ceil(wavefronts / SIMDs).
static double chunk_cost(int H, int K, int NO, int S, long long n, long long simds)
{
    const int NC = (H + S - 1) / S;
    const int cap = 64 / (K * NC);
    if (cap < 1) return 1e30;
    const long long waves = (n + cap - 1) / cap;
    const long long w = waves <= simds ? 1 : (waves + simds - 1) / simds;
    static const double g[9] = {1.0, 1.0, 1.19, 1.26, 1.31, 1.33, 1.36, 1.38, 1.41};
    const double a = 60.0 * H - 170.0;
    const double per_pass = (a < 200.0 ? 200.0 : a) + S * (350.0 + 200.0 * (NO > 1 ? NO - 1 : 0));
    return per_pass * (double)w / g[w > 8 ? 8 : w];
}

#define OCD_CPICK(HH, NN, LL, SS)                                                                         \
    if (H == HH && NO == NN && L == LL && (want == 0 || want == SS)) {                                    \
        const double c_ = chunk_cost(HH, p.K, NN, SS, p.n_problems, slots);                                   \
        if (c_ < best_cost) { best_cost = c_; best = SS; }                                                \
    }
#define OCD_CCASE(HH, NN, LL, SS) if (H == HH && NO == NN && L == LL && best == SS) return launch_chunk<HH, NN, LL, SS>(p, st);

hipError_t launch_chunk_dispatch(int H, int NO, int L, const KernelParams &p, hipStream_t st, bool launch, int want,
                                 int *chunk)
{
    const long long slots = 4ll * (p.n_cus > 0 ? p.n_cus : 256);   // SIMDs
    int best = 0;
    double best_cost = 1e29;
    OCD_CHUNK_TABLE(OCD_CPICK)
    *chunk = best;
    if (!launch || best == 0) return hipSuccess;
    OCD_CHUNK_TABLE(OCD_CCASE)
    return hipSuccess;
}

} // namespace ocd
