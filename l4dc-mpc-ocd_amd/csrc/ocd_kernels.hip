// ocd_kernels.hip -- hand-written gfx950 kernels of the batched MPC planner.
//
// What runs here (reference file:line, relative to the reference tree):
//   NaivePlanner.generate_plan / mpc_reward   interact_drive/planner/naive_planner.py:33-77,81-164
//   car_dynamics_step                         interact_drive/simulation_utils.py:9-21
//   ThreeLaneTestCar.features                 experiments/merging.py:32-83
//   _f / smooth_threshold / smooth_bump       interact_drive/math_utils.py:7-31,59-97,135-180
//   LinearRewardCar.reward_fn                 interact_drive/car/linear_reward_car.py:49-55
//   ValueFeature.interpolate_value            interact_drive/reward_design/value_interpolation.py:28-61
//   CarWorld.step, Car.step, FixedPlanCar     interact_drive/world.py:79-109, car/car.py:76-87,
//                                             car/fixed_plan_car.py:25-39
//   ReplanningCarWorld                        experiments/replanning_world.py:24-36
//   MPC_ORD.eval_weights_for_init             interact_drive/reward_design/mpc_ord.py:67-106
//
// Mapping.  A trajectory is one (candidate, init, sample) episode (or one world
// state in plan mode); it is optimised from K control initialisations (3, or 6 with extra_inits).
// Lane t of a segment owns horizon step t of one (trajectory, initialisation) pair: its control u_t,
// the state before and after step t, the reward features at the post-step state and their adjoint.
// Per SGD iteration the only sequential work is four short recurrences (v/heading forward, x/y
// forward, x/y adjoint, v/heading adjoint); everything else -- sincos, the feature exponentials and
// IEEE divisions, the per-step Jacobian products -- is lane-parallel.  Three variants of how the lanes
// of a segment exchange the recurrence terms, and of where the K initialisations live:
//   V_LDS -- K wavefronts per workgroup (wavefront k = initialisation k), segments of H lanes (64/H
//            per wavefront); every lane writes its term into a zero-padded LDS row and reads a
//            lane-shifted window, so no step needs a predicate (x + 0 == x).  Any H; throughput.
//   V_ROW -- K wavefronts per workgroup, H <= 16, one trajectory per 16-lane DPP row; H-1 rounds of
//            row_shr:1 / row_shl:1 moves between neighbouring lanes.  Smallest batches.
//   V_SEG -- ONE wavefront per workgroup, K*H <= 64: segment (j, k) = trajectory j, initialisation k;
//            wave_shr:1 / wave_shl:1 moves with a select at the segment boundary; the first-index
//            argmin over the K initialisations is taken inside the wavefront (ds_bpermute), no
//            workgroup barrier.  Batches of about one wavefront per SIMD.
//
// Numerics: IEEE binary32, -ffp-contract=off, operation order = the arithmetic contract of
// DESIGN.md section 3; exp/sin/cos from ocd_devmath.h.  No MFMA: there is no dense contraction here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/ocd.h"
#include "ocd_chains.h"
#include "ocd_lane.h"
#include "ocd_device.h"
#include "ocd_kernels.h"

namespace ocd {

// ---------------------------------------------------------------- segment exchange through LDS (V_LDS)
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
//   per-lane 0/1 mask inside the fma: fma(delta, 1, v) == v + delta and fma(delta, 0, v) == v, bit for bit
//   (kernels specialised on H), or selects (generic kernel).
//
// Wave-synchronous: DS operations of one wavefront execute in program order, so no s_barrier is
// needed inside the SGD loop; wave_barrier() only pins the compiler's schedule.
struct Geo {
    int SEGS;          // trajectories per wavefront (max)
    int ROW;           // padded elements per segment row
    int ROWS;          // +1: lanes past the last segment park here
    int PLANE4, PLANE2;
    int WAVE_FLOATS;   // floats per wavefront: one float4 plane followed by one float2 plane
    int SEL_FLOATS;    // selection record per (buffer, wavefront)
};

__host__ __device__ constexpr Geo geo_lds(int H)
{
    Geo g{};
    g.SEGS = 64 / H;
    g.ROW = 3 * H - 2;
    g.ROWS = g.SEGS + 1;
    g.PLANE4 = g.ROWS * g.ROW * 4;
    g.PLANE2 = g.ROWS * g.ROW * 2;
    g.WAVE_FLOATS = (g.PLANE4 + g.PLANE2 + 3) & ~3;   // keeps every wavefront's float4 plane 16-byte aligned
    g.SEL_FLOATS = g.ROWS * 4;
    return g;
}

// ---------------------------------------------------------------- the kernel
// LEAF: the terminal-value lookup (leaf_evaluation): the generic kernel for every shape, and the V_ROW / V_SEG LAT
//   builds of the reference's own horizons (5, 6) on the finite_horizon / local_opt shape (OCD_LEAF_TABLE).
// LAT (V_ROW / V_SEG): the launch puts at most one wavefront on a SIMD, so it lasts as long as its slowest
//   wavefront, and a lone wavefront pays 2 issue slots for every branch it does not take, ~6 for one it
//   takes (tools/microbench/valu_latency.hip).  The slowest wavefront has a fence or collision lane in
//   nearly every pass, so this build evaluates reward_one unconditionally and straight-line (no "none
//   active" path, no has_col / has_f skips) and repairs the rare pass with a multi-feature lane afterwards.
//   The diagnostics knobs no_feature_skips / no_unified_features select the LAT = false build.
template <int HT, int NO, int L, int V, bool LEAF = false, bool LAT = false>
__global__ void __launch_bounds__(V == V_SEG ? 64 : 64 * OCD_MAX_CTRL_INITS, 1)
mpc_kernel(const KernelParams p)
{
    static_assert(!LAT || V == V_ROW || V == V_SEG, "LAT is a V_ROW / V_SEG build");
    static_assert(HT > 0 || V == V_LDS, "the generic (run-time H) kernel exchanges through LDS");
    static_assert(V != V_ROW || HT <= 16, "V_ROW keeps a trajectory inside one 16-lane DPP row");
    constexpr int NOA = NO > 0 ? NO : 1;
    constexpr bool lane_feats = L > 0;
    const ocd_scenario_desc &d = p.d;
    const int H = HT > 0 ? HT : d.horizon;
    const Geo G = geo_lds(H);
    const int ROW = G.ROW;

    extern __shared__ float4 lds_raw[];
    float *lds = reinterpret_cast<float *>(lds_raw);
    const int K = p.K;
    const int wave = (V == V_SEG) ? 0 : (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int seg = (V == V_ROW) ? (lane >> 4) : (lane / H);      // V_LDS: G.SEGS for the parked tail lanes
    const int t = (V == V_ROW) ? (lane & 15) : (lane - seg * H);
    const bool in_h = t < H;                                      // V_ROW: lanes H..15 of a row idle along
    const bool first = t == 0, last = t == H - 1;
    const unsigned long long first_mask = __ballot(first), last_mask = __ballot(last);
#ifdef OCD_NO_ASM_CHAINS                                                      // ablation builds (tools/ablation.sh)
    constexpr bool asm_chains = false;
#else
    constexpr bool asm_chains = (V != V_LDS) && chain_supported<HT>::value;   // hand-scheduled recurrences (ocd_chains.h)
#endif
    // trajectory slot inside the wavefront and control initialisation this lane works for
    const int slot = (V == V_SEG) ? (seg / K) : seg;
    const int kinit = (V == V_SEG) ? (seg - slot * K) : wave;

    // ---- V_LDS / V_ROW: LDS (recurrence windows, selection records) ----
    float4 *own4 = nullptr;
    float2 *own2 = nullptr;
    const float2 *fwd2 = nullptr, *bwd2 = nullptr, *data2 = nullptr;
    const float4 *bwd4 = nullptr;
    float *sel = nullptr;
    if (V != V_SEG) {
        // zero the pads once (data slots are always written before they are read)
        if (V == V_LDS) {
            for (int i = threadIdx.x; i < K * G.WAVE_FLOATS; i += blockDim.x) lds[i] = 0.0f;
            __syncthreads();
            float4 *plane4 = reinterpret_cast<float4 *>(lds + (size_t)wave * G.WAVE_FLOATS) + seg * ROW;
            float2 *plane2 = reinterpret_cast<float2 *>(lds + (size_t)wave * G.WAVE_FLOATS + G.PLANE4) + seg * ROW;
            own4 = plane4 + (H - 1 + t);                          // this lane's data slot
            own2 = plane2 + (H - 1 + t);
            fwd2 = plane2 + t;                                    // forward window: element i+t at step i
            bwd4 = plane4 + (2 * H - 2 + t);                      // backward window: element 2H-2+t-i at step i
            bwd2 = plane2 + (2 * H - 2 + t);
            data2 = plane2 + (H - 1);                             // the segment's H terms, in order
        }
        sel = lds + ((V == V_LDS) ? (size_t)K * G.WAVE_FLOATS : 0);   // [2][K][ROWS][4] selection records
    }
    // LEAF: the cell boundaries of the value table (n0 + n1 + n2 floats) staged in LDS behind the variant's own use
    LeafTable leaf;
    leaf.grid = p.leaf.grid; leaf.values = p.leaf.values; leaf.proj_kind = p.leaf.proj_kind;
    leaf.n[0] = p.leaf.n[0]; leaf.n[1] = p.leaf.n[1]; leaf.n[2] = p.leaf.n[2];
    if constexpr (LEAF) {
        if (p.leaf.grid_in_lds) {
            const size_t off = (V == V_SEG) ? 0 : ((V == V_ROW) ? (size_t)2 * K * G.SEL_FLOATS
                                                               : (size_t)K * G.WAVE_FLOATS + (size_t)2 * K * G.SEL_FLOATS);
            float *lg = lds + off;
            const int ng = p.leaf.n[0] + p.leaf.n[1] + p.leaf.n[2];
            for (int i = threadIdx.x; i < ng; i += blockDim.x) lg[i] = p.leaf.grid[i];
            __syncthreads();
            leaf.grid = lg;
        }
        leaf_guess_setup(leaf);
    }
    // V_LDS, H-specialised: 0/1 masks of the forward speed recurrence: step i updates lane t iff i >= H-1-t
    float mfw[HT > 1 ? HT - 1 : 1];
    if (V == V_LDS && HT > 1) {
#pragma unroll
        for (int i = 0; i < HT - 1; ++i) {
            mfw[i] = (i >= HT - 1 - t) ? 1.0f : 0.0f;
            asm volatile("" : "+v"(mfw[i]));                      // keep them in registers, do not rematerialise
        }
    }

    // segs_used trajectories per wavefront: small batches are spread over more wavefronts so that the
    // uniform feature skips act per trajectory; big batches pack.
    const long long prob_raw = (long long)blockIdx.x * p.segs_used + slot;
    const bool row_live = (slot < p.segs_used) && (prob_raw < p.n_problems);
    const bool live = in_h && row_live;
    const long long prob = row_live ? prob_raw : (p.n_problems - 1); // parked lanes shadow a real problem

    const float dt = d.dt, dt2 = d.dt_sq, fr = d.ego_friction, lr = d.learning_rate;
    constexpr int D = feat_dim(L);
    // degenerate fence (KernelParams::two_sided): the generic kernel only -- a compile-time false in every specialised one
    const bool two_sided = (HT == 0) && p.two_sided != 0;

    // ---- problem inputs -------------------------------------------------
    float ex, ey, ev, eth;                    // ego state
    float ox[NOA], oy[NOA], ov[NOA], oth[NOA];
    float w[OCD_MAX_FEATURES];
    int tp_idx = 0;                           // which entry of the teleport cycle applies to this episode
    if (p.mode == OCD_MODE_ROLLOUT && !p.from_state) {
        long long p_, n_;
        const bool row_ok = episode_rows(p, prob, p_, n_, tp_idx);   // flat (p, n, s) index, or the caller's episode index
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
    const LaneGradConst<L> lgc = lane_grad_const<L>(w, d);   // lane-feature gradient factors of this trajectory
    float wd[OCD_MAX_FEATURES];               // designer weights (uniform)
#pragma unroll
    for (int k = 0; k < OCD_MAX_FEATURES; ++k) wd[k] = d.designer_weights[k];

    const int T = p.T;
    const PkConsts pkc = pk_consts();         // constants of the packed exp (ocd_devmath.h), pinned in registers
    ScConsts scc;                             // coefficient pairs of the two-wide sin / cos polynomials: V_ROW latency
    if constexpr (LAT && !(V == V_SEG && asm_chains)) scc = sc_consts();   //   builds (V_SEG with its asm chains runs the scalar ones, see ocd_devmath.h; without them -- the -DOCD_NO_ASM_CHAINS ablation arm -- it calls sincos_pk too)
    // The latency builds run one wavefront per SIMD (the launcher takes them only then, round 6) and claim theirs: touching
    // nothing, the clobber raises the kernel's register allocation above half a SIMD's file, so no second wavefront -- of this
    // launch or of ANOTHER one -- fits beside it (ocd_chunk_kernel.hip does the same).  Two launches on two streams then sit
    // side by side on different SIMDs instead of sharing them: the lockstep path's groups (csrc/ocd_cma.c: without the claim
    // the second launch took 1.92 ms beside a 1.18 ms first one, profiles/r06_two_streams.txt).
    if constexpr (LAT) asm volatile("" ::: "a255");
    OCD_STAMP_DECL
    float G_ret = 0.0f;
    const BumpGeom bg0 = {0.0f, 1.0f, 0.0f, 1.0f};
    const bool writer = live && kinit == 0 && first;          // one lane per trajectory writes its outputs
    constexpr bool has_leaf = LEAF;                           // terminal value replaces the last step's reward
    // lanes whose reward features count (the last horizon step is scored by the terminal value instead)
    const bool feat_live = live && !(has_leaf && last);
    const unsigned long long feat_mask = __ballot(feat_live);
    const unsigned long long force_full = p.force_full, force_full_any = p.force_full_any;

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
            // ReplanningCarWorld.step: self.t += 1; teleport when self.t == critical_t
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
            // designer reward of the pre-step state (mpc_ord.py:99)
            BumpGeom bgd[NOA];
            bgd[0] = bg0;
#pragma unroll
            for (int j = 0; j < NO; ++j) bgd[j] = bump_geom(ox[j], oy[j], d.bump_half_x, d.bump_half_y);
            float s_, c_;
            sincos_(eth, s_, c_);
            Q4 qd;
            const float r = reward_state<NO, L, false, true>(d, wd, ex, ey, ev, s_, c_, bgd, qd, nullptr, true, true, two_sided);
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
        float wx1[NOA], wy1[NOA];                // 1.001 * bump half-widths (needs_collision1)
#pragma unroll
        for (int j = 0; j < NOA; ++j) { wx1[j] = bg[j].wx * 1.001f; wy1[j] = bg[j].wy * 1.001f; }
        // one scripted car: the refined reciprocals of its half-widths in this control step (reward_one's FASTDIV form divides by
        // them in every pass) and the lanes whose widths are outside its guard
        BumpRecip br[NOA];
        unsigned long long widths_beyond = 0ull, widths_degenerate = 0ull;
#pragma unroll
        for (int j = 0; j < NOA; ++j) {
            br[j] = BumpRecip{0.0f, 0.0f};
            if constexpr (lane_feats && NO > 0) widths_degenerate |= __builtin_amdgcn_ballot_w64(bump_widths_degenerate(bg[j])) & feat_mask;
            if constexpr (lane_feats && NO == 1) {
                br[j].rx = refined_recip(bg[j].wx);
                br[j].ry = refined_recip(bg[j].wy);
                widths_beyond |= __builtin_amdgcn_ballot_w64(!bump_widths_guarded(bg[j]));
            }
        }
        // a degenerate width (ocd_device.h: bump_widths_degenerate) sends every pass of this control step to the evaluation of
        // every feature: through `beyond` in the straight-line builds, through the knob mask in the others
        widths_beyond |= widths_degenerate;
        // a descriptor outside LaneGradConst::x_hi's conditions (x_hi = 0) has no shortened division anywhere: the
        // evaluations that run fence units on lanes outside the fence region (reward_fc / reward_fcc / reward_every)
        // rely on those conditions too, so every live lane counts as beyond the guard, not only the fence lanes
        if constexpr (lane_feats) widths_beyond |= (lgc.x_hi > 0.0f) ? 0ull : ~0ull;
        const unsigned long long force_full_step = force_full | widths_degenerate;

        // ---- this lane's control initialisation (naive_planner.py:107-116) ----
        float s0, c0;
        sincos_(eth, s0, c0);
        // extra_inits coast at friction * self.car.state[2] ** 2 (naive_planner.py:114): the car's own speed, which
        // a caller planning from a foreign init_state passes apart (ocd_plan_batch_from); else the state's ego speed
        const float v_car = (p.init_speed != nullptr) ? p.init_speed[prob] : ev;
        const float a_coast = fr * (v_car * v_car);
        const int k3 = kinit % 3;
        float ua = (kinit >= 3) ? a_coast : 0.0f;
        float uw = (k3 == 0) ? 0.0f : ((k3 == 1) ? -0.65f : 0.65f);

        // V_LDS: fma(delta, 0, v) == v needs a finite delta; in the masked steps delta is a function of the
        // CURRENT speed ev only, so one wave-uniform test per control step selects the exact fallback
        // (the world state itself can overflow after enough hard-braking steps; see the tests).
        const bool ev_finite = __ballot(!finite_((0.0f - fr * (ev * ev)) * dt) || !finite_(ev)) == 0ull;
        float loss = 0.0f;

        // one pass over the horizon: GRAD = an SGD step on (ua, uw); !GRAD = the objective only (loss)
        auto horizon_pass = [&](auto grad_tag) __attribute__((always_inline)) {
            constexpr bool GRAD = decltype(grad_tag)::value;
            OCD_STAMP(0);                                  // everything outside the passes
            // ===== forward =====
            const float a1 = min_tf(ua, 4.0f);
            const float a_c = max_tf(a1, -8.0f);
            const float w1 = min_tf(uw, 4.0f);
            const float w_c = max_tf(w1, -4.0f);
            const bool pass_a = (ua <= 4.0f) && (a1 >= -8.0f);
            const bool pass_w = (uw <= 4.0f) && (w1 >= -4.0f);
            const float wdt = w_c * dt;

            float v = ev, th = eth;
            if constexpr (asm_chains) {
                if (V == V_ROW) row_fwd_vth<HT>(v, th, a_c, wdt, fr, dt);
                else seg_fwd_vth<HT>(v, th, ev, eth, a_c, wdt, fr, dt, first_mask);
            } else if (V == V_ROW) {
                // every lane advances its own state by its own control and hands the result to the lane
                // above; after t rounds lane t holds the state before step t (lane 0 keeps the current state)
#pragma unroll
                for (int i = 0; i < H - 1; ++i) {
                    const float v_next = v + (a_c - fr * (v * v)) * dt;
                    const float th_next = th + wdt;
                    v = row_below(v, v_next);
                    th = row_below(th, th_next);
                }
            } else if (V == V_SEG) {
#pragma unroll
                for (int i = 0; i < H - 1; ++i) {
                    const float v_next = v + (a_c - fr * (v * v)) * dt;
                    const float th_next = th + wdt;
                    const float vb = wave_below(v_next), tb = wave_below(th_next);
                    v = first ? ev : vb;
                    th = first ? eth : tb;
                }
            } else {
                *own2 = make_float2(a_c, wdt);
                __builtin_amdgcn_wave_barrier();
                if (HT > 1 && ev_finite) {
#pragma unroll
                    for (int i = 0; i < HT - 1; ++i) {
                        const float2 aw = fwd2[i];
                        const float delta = (aw.x - fr * (v * v)) * dt;
                        v = fma_(delta, mfw[i], v);        // masked (leading) steps see v = ev: delta is finite
                        th = th + aw.y;
                    }
                } else {                                   // generic kernel / a non-finite increment: exact selects
#pragma unroll
                    for (int i = 0; i < H - 1; ++i) {
                        const float2 aw = fwd2[i];
                        const float vn_ = v + (aw.x - fr * (v * v)) * dt;
                        v = (i >= H - 1 - t) ? vn_ : v;
                        th = th + aw.y;
                    }
                }
            }
            OCD_STAMP(1);                                  // speed / heading recurrence
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
            float s_pre, c_pre, cd, sd;
#ifdef OCD_NO_SINCOS_PK
            constexpr bool sincos_fused = false;
            float x = ex, y = ey;
            sincos_(thn, sn, cn);
#else
            // V_SEG: quadrant fix-up, boundary shift, the step's increments AND the position recurrence in one
            // hand-scheduled statement (ocd_chains.h: seg_sincos_fwd_xy)
            constexpr bool sincos_fused = (V == V_SEG) && asm_chains;
            float x = ex, y = ey;
            if constexpr (sincos_fused) seg_sincos_fwd_xy<HT>(thn, scc, s0, c0, dd, ex, ey, first_mask, sn, cn, s_pre, c_pre, sd, cd, x, y);
            else if constexpr (LAT) sincos_pk(thn, sn, cn, scc);   // V_ROW latency build: two-wide chains, bit-select swap
            else sincos_(thn, sn, cn);
#endif
            if constexpr (!sincos_fused) {
                if (V == V_ROW) {
                    s_pre = row_below(s0, sn);             // lane 0 of the row keeps sin/cos of the current heading
                    c_pre = row_below(c0, cn);
                } else {
                    s_pre = wave_below(sn);
                    c_pre = wave_below(cn);
                    s_pre = first ? s0 : s_pre;
                    c_pre = first ? c0 : c_pre;
                }
                cd = c_pre * dd;
                sd = s_pre * dd;
            }
            OCD_STAMP(2);                                  // own step, sincos
            Q4 q;
            // LAT build of V_ROW at H = 10: the target-speed adjoint goes into the hazard slots of the position
            // recurrence (ocd_chains.h)
            constexpr bool phi0_in_chain = LAT && GRAD && asm_chains && V == V_ROW && HT == 10 && lane_feats;
            if constexpr (phi0_in_chain) {
                const float tgt = d.target_speed;
                row_fwd_xy_phi0_h10(x, y, row_below(0.0f, cd), row_below(0.0f, sd), vn, sn, cn, tgt, 4.0f * (tgt * tgt),
                                    w[0], q.qv, q.qth);
            } else if constexpr (sincos_fused) {
                // (the position recurrence ran inside seg_sincos_fwd_xy)
            } else if constexpr (asm_chains) {
                if (V == V_ROW) row_fwd_xy<HT>(x, y, row_below(0.0f, cd), row_below(0.0f, sd));
                else seg_fwd_xy<HT>(x, y, ex, ey, cd, sd, first_mask);
            } else if (V == V_ROW) {
#pragma unroll
                for (int i = 0; i < H - 1; ++i) {
                    x = row_below(x, x + cd);              // lane 0 keeps ex
                    y = row_below(y, y + sd);
                }
            } else if (V == V_SEG) {
#pragma unroll
                for (int i = 0; i < H - 1; ++i) {
                    const float xb = wave_below(x + cd), yb = wave_below(y + sd);
                    x = first ? ex : xb;
                    y = first ? ey : yb;
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
            OCD_STAMP(3);                                  // position recurrence

            // terminal value: corner search and the eight table loads start here, the features of the other lanes
            // issue while they are in flight, leaf_finish consumes them afterwards
            LeafLoad lload;
            if constexpr (has_leaf) leaf_prepare(leaf, xn, yn, vn, sn, lload);

            // ===== reward features at the post-step state =====
            // wave-uniform choice of the evaluation: none of {fence, collisions} active on any live lane /
            // at most one of them per lane (reward_one) / everything (reward_state)
            float r = 0.0f;
            if constexpr (lane_feats) {
                bool nc[NOA];
                nc[0] = false;
                const bool nf = needs_fence(d, xn);
                // multi_f: some lane has the fence AND a car active; multi_c: some lane is inside two cars' boxes
                unsigned long long mf = __builtin_amdgcn_ballot_w64(nf) & feat_mask, mc_any = 0ull, multi_f = 0ull, multi_c = 0ull;
                unsigned long long tiny_n = 0ull;
#pragma unroll
                for (int j = 0; j < NO; ++j) {
                    const float dx = xn - bg[j].cx, dy = yn - bg[j].cy;
                    const bool ncx = __builtin_fabsf(dx) < wx1[j], ncy = __builtin_fabsf(dy) < wy1[j];
                    nc[j] = ncx && ncy;
                    const unsigned long long mj = __builtin_amdgcn_ballot_w64(ncx) & __builtin_amdgcn_ballot_w64(ncy) & feat_mask;
                    multi_f |= (mj & mf);
                    multi_c |= (mj & mc_any);
                    mc_any |= mj;
                    if constexpr (LAT && NO == 1)          // a zero / tiny numerator of the shortened (x - cx) / wx
                        tiny_n |= __builtin_amdgcn_ballot_w64(__builtin_fabsf(dx) < 7.888609052210118e-31f) |
                                  __builtin_amdgcn_ballot_w64(__builtin_fabsf(dy) < 7.888609052210118e-31f);
                }
                const bool has_f = mf != 0ull, has_col = mc_any != 0ull;
                const unsigned long long any_feat = mf | mc_any;
                // LAT: a fence lane beyond the guard of the shortened reciprocals (LaneGradConst::x_hi) -> full divisions
                unsigned long long beyond = 0ull;
                if constexpr (LAT)
                    beyond = (__builtin_amdgcn_ballot_w64(!(__builtin_fabsf(xn) < lgc.x_hi)) & mf) | ((tiny_n | widths_beyond) & feat_mask);
                OCD_STAMP(4);                              // choice of the evaluation
                if constexpr (LAT && NO >= 2) {
                    // several scripted cars: the reference's scenarios of that kind (replanning, merging) put cars where
                    // their collision box overlaps the fence region, a pass with a multi-feature lane is COMMON (most
                    // passes of the slowest wavefronts): decided before the evaluation, one evaluation per pass
                    if (__builtin_expect((multi_f | multi_c | beyond) != 0ull, 0)) {
                        if (__builtin_expect(beyond != 0ull, 0)) {
                            if ((multi_c | widths_degenerate) != 0ull) r = reward_every<NO, L, GRAD>(d, w, xn, yn, vn, sn, cn, bg, q, pkc, &lgc, feat_mask);
                            else if (multi_f != 0ull) r = reward_fc<NO, L, GRAD>(d, w, xn, yn, vn, sn, cn, bg, nc, q, pkc, &lgc, feat_mask);
                            else r = reward_one<NO, L, GRAD, false, phi0_in_chain>(d, w, xn, yn, vn, sn, cn, bg, br, nc, nf, true, true, q, pkc, lgc, feat_mask);
                        } else if (multi_c != 0ull) r = reward_every<NO, L, GRAD, GRAD>(d, w, xn, yn, vn, sn, cn, bg, q, pkc, &lgc, feat_mask);
                        else r = reward_fc<NO, L, GRAD, GRAD>(d, w, xn, yn, vn, sn, cn, bg, nc, q, pkc, &lgc, feat_mask);
                        OCD_STAMP(5); OCD_STAMP_COUNT(12); // every feature / fence + one car per lane / full divisions
                    } else {
                        r = reward_one<NO, L, GRAD, false, phi0_in_chain, GRAD>(d, w, xn, yn, vn, sn, cn, bg, br, nc, nf, true, true, q, pkc, lgc, feat_mask);
                        OCD_STAMP(6); OCD_STAMP_COUNT(13); // one feature per lane
                    }
                } else if constexpr (LAT) {
                    // one scripted car (finite_horizon, local_opt: its collision box and the fence region do not overlap,
                    // a multi-feature lane is rare or impossible): one feature per lane, straight line; the rare pass is
                    // repaired afterwards, out of line -- the hot path carries no trace of it
                    r = reward_one<NO, L, GRAD, false, phi0_in_chain, GRAD>(d, w, xn, yn, vn, sn, cn, bg, br, nc, nf, true, true, q, pkc, lgc, feat_mask);
                    OCD_STAMP(6); OCD_STAMP_COUNT(13);     // one feature per lane
                    if (__builtin_expect((multi_f | multi_c | beyond) != 0ull, 0)) {
                        if ((multi_c | widths_degenerate) != 0ull) r = reward_every<NO, L, GRAD>(d, w, xn, yn, vn, sn, cn, bg, q, pkc, &lgc, feat_mask);
                        else if (multi_f != 0ull) r = reward_fc<NO, L, GRAD>(d, w, xn, yn, vn, sn, cn, bg, nc, q, pkc, &lgc, feat_mask);
                        else r = reward_one<NO, L, GRAD, false, phi0_in_chain>(d, w, xn, yn, vn, sn, cn, bg, br, nc, nf, true, true, q, pkc, lgc, feat_mask);
                        OCD_STAMP(5); OCD_STAMP_COUNT(12); // every feature / fence + one car per lane / full divisions
                    }
                } else {
                    // (the diagnostics knobs enter as two wave-uniform masks: two scalar tests decide the path)
                    const unsigned long long full_m = multi_c | force_full_step | (any_feat & force_full_any);
                    if (__builtin_expect((any_feat | force_full_step) != 0ull, 1)) {
                        if (__builtin_expect(full_m != 0ull, 0)) {
                            r = reward_state<NO, L, GRAD>(d, w, xn, yn, vn, sn, cn, bg, q, nullptr, true, true, two_sided);
                            OCD_STAMP(5); OCD_STAMP_COUNT(12);     // every feature
                        } else if (multi_f != 0ull) {
                            r = reward_fc<NO, L, GRAD>(d, w, xn, yn, vn, sn, cn, bg, nc, q, pkc);
                            OCD_STAMP(5); OCD_STAMP_COUNT(12);     // fence + one car per lane
                        } else {
                            // the shortened reciprocals (ocd_devmath.h: recip_pair_guarded) unless a fence lane is beyond
                            // their guard (LaneGradConst::x_hi)
                            // (one scripted car: also (x - cx) / wx by the reciprocals of this control step, quot2_by_recip)
                            constexpr bool ZN1 = GRAD && NO == 1;
                            unsigned long long tiny_n = 0ull;
                            if constexpr (ZN1)
                                tiny_n = (__ballot(__builtin_fabsf(xn - bg[0].cx) < 7.888609052210118e-31f) |
                                          __ballot(__builtin_fabsf(yn - bg[0].cy) < 7.888609052210118e-31f) | widths_beyond) & feat_mask;
                            if (!GRAD || ((__ballot(!(__builtin_fabsf(xn) < lgc.x_hi)) & mf) | tiny_n) != 0ull)
                                r = reward_one<NO, L, GRAD, true>(d, w, xn, yn, vn, sn, cn, bg, br, nc, nf, has_col, has_f, q, pkc, lgc, feat_mask);
                            else
                                r = reward_one<NO, L, GRAD, true, false, GRAD, ZN1>(d, w, xn, yn, vn, sn, cn, bg, br, nc, nf, has_col, has_f, q, pkc, lgc, feat_mask);
                            OCD_STAMP(6); OCD_STAMP_COUNT(13);     // one feature per lane
                            if (has_col) OCD_STAMP_COUNT(11);
                            if (has_f) OCD_STAMP_COUNT(15);
                        }
                    } else {
                        r = reward_state<NO, L, GRAD>(d, w, xn, yn, vn, sn, cn, bg, q, nullptr, false, false);
                        OCD_STAMP(7); OCD_STAMP_COUNT(14);     // neither fence nor collision
                    }
                }
            } else {
                r = reward_state<NO, L, GRAD>(d, w, xn, yn, vn, sn, cn, bg, q, nullptr);
            }
            if constexpr (has_leaf) {                      // naive_planner.py:69-70
                Q4 ql;
                const float rl_ = leaf_finish<GRAD>(leaf, lload, vn, sn, cn, ql);
                r = last ? rl_ : r;
                if (GRAD) {
                    q.qx = last ? ql.qx : q.qx; q.qy = last ? ql.qy : q.qy;
                    q.qv = last ? ql.qv : q.qv; q.qth = last ? ql.qth : q.qth;
                }
            }

            if constexpr (!GRAD) {
                // ===== objective only (naive_planner.py:154): r = 0; r += reward, t = 0..H-1 =====
                float Rsum = 0.0f;
                if (V == V_ROW) {
                    // running sum up the row: after H-1 rounds lane t holds ((0 + r_0) + r_1) + ... + r_t
                    float S = 0.0f + r;
#pragma unroll
                    for (int i = 0; i < H - 1; ++i) {
                        // the DPP move must execute in ALL lanes (a lane that skipped it would be an
                        // invalid source for its neighbour): move first, select afterwards
                        const float below = row_below(0.0f, S);
                        S = first ? S : (below + r);
                    }
                    Rsum = S;                              // complete in lane H-1, which publishes the loss
                } else if (V == V_SEG) {
                    float S = 0.0f + r;
#pragma unroll
                    for (int i = 0; i < H - 1; ++i) {
                        const float below = wave_below(S);
                        S = first ? S : (below + r);
                    }
                    Rsum = S;                              // complete in lane H-1 of the segment
                } else {
                    __builtin_amdgcn_wave_barrier();
                    *own2 = make_float2(r, 0.0f);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int j = 0; j < H; ++j) Rsum = Rsum + data2[j].x;
                    __builtin_amdgcn_wave_barrier();
                }
                loss = -Rsum;
            } else {
                // ===== backward =====
                float Lx = 0.0f, Ly = 0.0f;
                if (V == V_ROW) {
                    // idle lanes (t >= H) contribute nothing to the adjoints flowing down the row
                    // (selects, not a divergent branch: a branch costs a lone wavefront 2-6 issue slots)
                    q.qx = in_h ? q.qx : 0.0f; q.qy = in_h ? q.qy : 0.0f;
                    q.qv = in_h ? q.qv : 0.0f; q.qth = in_h ? q.qth : 0.0f;
                    const float qx_a = row_above(0.0f, q.qx); // adjoint term of the step above (0 for the top lane)
                    const float qy_a = row_above(0.0f, q.qy);
#pragma unroll
                    for (int i = 0; i < H - 1; ++i) {
                        Lx = qx_a + row_above(0.0f, Lx);
                        Ly = qy_a + row_above(0.0f, Ly);
                    }
                } else if (V == V_SEG && asm_chains) {
                    if constexpr (asm_chains) seg_bwd_xy<HT>(Lx, Ly, q.qx, q.qy, last_mask);
                } else if (V == V_SEG) {
#pragma unroll
                    for (int i = 0; i < H - 1; ++i) {
                        const float ax_ = wave_above(q.qx + Lx), ay_ = wave_above(q.qy + Ly);
                        Lx = last ? 0.0f : ax_;            // the top lane of a segment receives nothing
                        Ly = last ? 0.0f : ay_;
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
                OCD_STAMP(8);                              // position adjoint recurrence
                const float Ax = q.qx + Lx;
                const float Ay = q.qy + Ly;
                const float g_c = Ax * dd;
                const float g_s = Ay * dd;
                const float g_d = Ax * c_pre + Ay * s_pre;
                const float tau = (-g_c) * s_pre + g_s * c_pre;
                const float gv1 = g_d * dt;
                const float gA1 = (g_d * dt2) * 0.5f;
                float Lv = 0.0f, Lth = 0.0f;
                if (V == V_ROW) {
                    const float gA1_m = in_h ? gA1 : 0.0f, gv1_m = in_h ? gv1 : 0.0f;
                    const float v_m = in_h ? v : 0.0f, tau_m = in_h ? tau : 0.0f;
                    const float qth_a = row_above(0.0f, q.qth);
                    const float tau_a = row_above(0.0f, tau_m);
#pragma unroll
                    for (int i = 0; i < H - 1; ++i) {
                        // what this lane's step sends down to the lane below, from what it has received so far
                        const float Av_ = q.qv + Lv;
                        const float gA_ = gA1_m + Av_ * dt;
                        const float gv2_ = (-gA_) * fr;
                        const float gv3_ = (gv2_ * 2.0f) * v_m;
                        const float Lv_down = (gv1_m + Av_) + gv3_;
                        Lv = row_above(0.0f, Lv_down);
                        Lth = (qth_a + row_above(0.0f, Lth)) + tau_a;
                    }
                } else if (V == V_SEG && asm_chains) {
                    if constexpr (asm_chains) seg_bwd_vth<HT>(Lv, Lth, q.qv, q.qth, gA1, gv1, v, tau, fr, dt, last_mask);
                } else if (V == V_SEG) {
#pragma unroll
                    for (int i = 0; i < H - 1; ++i) {
                        const float Av_ = q.qv + Lv;
                        const float gA_ = gA1 + Av_ * dt;
                        const float gv2_ = (-gA_) * fr;
                        const float gv3_ = (gv2_ * 2.0f) * v;
                        const float Lv_down = (gv1 + Av_) + gv3_;
                        const float Lth_down = (q.qth + Lth) + tau;
                        const float lv_ = wave_above(Lv_down), lt_ = wave_above(Lth_down);
                        Lv = last ? 0.0f : lv_;
                        Lth = last ? 0.0f : lt_;
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
                OCD_STAMP(9);                              // Jacobian products, speed / heading adjoint recurrence
                const float Av = q.qv + Lv;
                const float gA = gA1 + Av * dt;
                const float Ath = q.qth + Lth;
                const float grad_a = pass_a ? gA : 0.0f;
                const float grad_w = pass_w ? (Ath * dt) : 0.0f;
                // SGD on loss = -R:  u <- u + lr * dR/du
                ua = ua + lr * grad_a;
                uw = uw + lr * grad_w;
                if (V == V_LDS) __builtin_amdgcn_wave_barrier();
                OCD_STAMP(10);                             // control update
            }
        };

        const int n_iter = d.n_iter;
        if constexpr (LAT) {                              // two passes per taken back-edge
            int it = 0;
            for (; it + 1 < n_iter; it += 2) { horizon_pass(bool_c<true>{}); horizon_pass(bool_c<true>{}); }
            if (it < n_iter) horizon_pass(bool_c<true>{});
        } else {
            for (int it = 0; it < n_iter; ++it) horizon_pass(bool_c<true>{});
        }
        horizon_pass(bool_c<false>{});

        // the objective's horizon sum is complete in every lane (V_LDS) or in lane H-1 (V_ROW, V_SEG)
        const bool has_loss = (V == V_LDS) ? first : last;
        // ---- per-initialisation outputs (plan mode, parity tests) ----
        if (p.mode == OCD_MODE_PLAN && live) {
            if (p.all_plans_out) {
                float *o = p.all_plans_out + (((size_t)prob * K + kinit) * H + t) * 2;
                o[0] = ua; o[1] = uw;
            }
            if (p.all_losses_out && has_loss) p.all_losses_out[(size_t)prob * K + kinit] = loss;
        }

        // ---- first-index argmin over the K initialisations (naive_planner.py:161-162) ----
        int best = 0;
        float bl, ca, cw;
        if (V == V_SEG) {
            const int base = slot * K * H;                      // first lane of this trajectory's K segments
            bl = lane_read(loss, base + H - 1);
            for (int k = 1; k < K; ++k) {
                const float lk = lane_read(loss, base + k * H + H - 1);
                if (lk < bl) { bl = lk; best = k; }
            }
            ca = lane_read(ua, base + best * H);
            cw = lane_read(uw, base + best * H);
        } else {
            float *selb = sel + (size_t)(step & 1) * K * G.SEL_FLOATS;
            {
                float *rec = selb + ((size_t)wave * G.ROWS + seg) * 4;
                if (has_loss) rec[0] = loss;
                if (first) { rec[1] = ua; rec[2] = uw; }
            }
            __syncthreads();
            bl = selb[((size_t)0 * G.ROWS + seg) * 4];
            for (int k = 1; k < K; ++k) {
                const float lk = selb[((size_t)k * G.ROWS + seg) * 4];
                if (lk < bl) { bl = lk; best = k; }
            }
            const float *brec = selb + ((size_t)best * G.ROWS + seg) * 4;
            ca = brec[1]; cw = brec[2];
        }

        if (p.mode == OCD_MODE_PLAN) {
            if (live && kinit == best) {
                float *o = p.plans_out + ((size_t)prob * H + t) * 2;
                o[0] = ua; o[1] = uw;
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
                const int gstep = p.t0 + step;    // FixedPlanCar.t (fixed_plan_car.py:25-31)
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
    if constexpr (LAT) {                                       // (slot 14 counts a path the latency builds do not have)
        unsigned hw, xcc;                                      // where this wavefront ran: HW_ID | XCC_ID << 32
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st_acc[14] = (unsigned long long)hw | ((unsigned long long)(xcc & 0xf) << 32) | (1ull << 40);
    }
    if (p.debug && lane == 0)
        for (int i = 0; i < 16; ++i) p.debug[((size_t)blockIdx.x * (V == V_SEG ? 1 : K) + wave) * 16 + i] = st_acc[i];
#endif
}

// ---------------------------------------------------------------- small kernels
template <int NO, int L>
__global__ void reward_kernel(const KernelParams p, float *feats_out, float *reward_out)
{
    constexpr int NOA = NO > 0 ? NO : 1;
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.n_problems) return;
    const ocd_scenario_desc &d = p.d;
    constexpr int D = feat_dim(L);
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
    const float r = reward_state<NO, L, false, true>(d, w, ws[0], ws[1], ws[2], s_, c_, bg, q,    // reward_fn: scored form
                                            feats_out ? feats_out + b * D : nullptr, true, true, p.two_sided != 0);
    if (reward_out) reward_out[b] = r;
}

// mpc_reward (naive_planner.py:33-77) and its gradient w.r.t. caller-supplied controls: one thread per
// problem, sequential over the horizon exactly like the traced graph (the planner kernel above computes
// the same quantities with one lane per horizon step).  Not a hot path: the tape lives in scratch.
template <int NO, int L>
__global__ void objective_kernel(const KernelParams p, const float *controls, float *reward_out, float *grad_out,
                                 float *traj_out)
{
    constexpr int NOA = NO > 0 ? NO : 1;
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.n_problems) return;
    const ocd_scenario_desc &d = p.d;
    constexpr int D = feat_dim(L);
    const int H = d.horizon;
    const float dt = d.dt, dt2 = d.dt_sq, fr = d.ego_friction;
    const float *ws = p.ego_states + b * (NO + 1) * 4;
    const float *wp = p.weights ? (p.weights + (p.weights_per_problem ? b * D : 0)) : nullptr;
    float w[OCD_MAX_FEATURES];
#pragma unroll
    for (int k = 0; k < OCD_MAX_FEATURES; ++k) w[k] = (wp && k < D) ? wp[k] : 0.0f;
    const float *u = controls + b * H * 2;
    const bool has_leaf = p.leaf.values != nullptr;
    LeafTable leaf;
    leaf.grid = p.leaf.grid; leaf.values = p.leaf.values; leaf.proj_kind = p.leaf.proj_kind;
    leaf.n[0] = p.leaf.n[0]; leaf.n[1] = p.leaf.n[1]; leaf.n[2] = p.leaf.n[2];
    if (has_leaf) leaf_guess_setup(leaf);

    // scripted cars as the planner models them
    float px[NOA], py[NOA], pv[NOA], pth[NOA];
#pragma unroll
    for (int j = 0; j < NO; ++j) { px[j] = ws[4 * (j + 1)]; py[j] = ws[4 * (j + 1) + 1]; pv[j] = ws[4 * (j + 1) + 2]; pth[j] = ws[4 * (j + 1) + 3]; }

    struct Tape { float v, c, s, dd; bool pass_a, pass_w; Q4 q; };
    Tape tape[OCD_MAX_HORIZON];
    float x = ws[0], y = ws[1], v = ws[2], th = ws[3];
    float s_, c_;
    sincos_(th, s_, c_);
    float R = 0.0f;
    for (int t = 0; t < H; ++t) {
        const float a = u[2 * t], om = u[2 * t + 1];
        const float a1 = min_tf(a, 4.0f), a_c = max_tf(a1, -8.0f);
        const float w1 = min_tf(om, 4.0f), w_c = max_tf(w1, -4.0f);
        Tape &tp = tape[t];
        tp.pass_a = (a <= 4.0f) && (a1 >= -8.0f);
        tp.pass_w = (om <= 4.0f) && (w1 >= -4.0f);
        const float v2 = v * v;
        const float fv2 = fr * v2;
        const float acc = a_c - fv2;
        const float vdt = v * dt;
        const float hA = 0.5f * acc;
        const float hAdt2 = hA * dt2;
        const float dd = vdt + hAdt2;
        tp.v = v; tp.c = c_; tp.s = s_; tp.dd = dd;
        const float xn = x + c_ * dd, yn = y + s_ * dd, vn = v + acc * dt, thn = th + w_c * dt;
        BumpGeom bg[NOA];
        bg[0] = BumpGeom{0.0f, 1.0f, 0.0f, 1.0f};
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            float so, co;
            sincos_(pth[j], so, co);
            if (p.other_plans) {
                const float oa = p.other_plans[(j * H + t) * 2], ow = p.other_plans[(j * H + t) * 2 + 1];
                const float dist = pv[j] * dt + (0.5f * oa) * dt2;
                px[j] = px[j] + co * dist;
                py[j] = py[j] + so * dist;
                pv[j] = pv[j] + oa * dt;
                pth[j] = pth[j] + ow * dt;
            } else {
                px[j] = px[j] + (co * pv[j]) * dt;
                py[j] = py[j] + (so * pv[j]) * dt;
            }
            bg[j] = bump_geom(px[j], py[j], d.bump_half_x, d.bump_half_y);
        }
        float sn, cn;
        sincos_(thn, sn, cn);
        float r;
        if (has_leaf && t == H - 1) r = leaf_value<true>(leaf, xn, yn, vn, sn, cn, tp.q);
        else r = reward_state<NO, L, true>(d, w, xn, yn, vn, sn, cn, bg, tp.q, nullptr, true, true, p.two_sided != 0);
        R = R + r;
        x = xn; y = yn; v = vn; th = thn; s_ = sn; c_ = cn;
        if (traj_out) {
            float *o = traj_out + (b * H + t) * 4;
            o[0] = x; o[1] = y; o[2] = v; o[3] = th;
        }
    }
    if (reward_out) reward_out[b] = R;
    if (!grad_out) return;
    float Lx = 0.0f, Ly = 0.0f, Lv = 0.0f, Lth = 0.0f;
    for (int t = H - 1; t >= 0; --t) {
        const Tape &tp = tape[t];
        const float Ax = tp.q.qx + Lx, Ay = tp.q.qy + Ly, Av = tp.q.qv + Lv, Ath = tp.q.qth + Lth;
        const float g_c = Ax * tp.dd;
        const float g_s = Ay * tp.dd;
        const float g_d = Ax * tp.c + Ay * tp.s;
        const float tau = (-g_c) * tp.s + g_s * tp.c;
        const float gv1 = g_d * dt;
        const float gA1 = (g_d * dt2) * 0.5f;
        const float gA = gA1 + Av * dt;
        const float g_v2 = (-gA) * fr;
        const float gv3 = (g_v2 * 2.0f) * tp.v;
        grad_out[(b * H + t) * 2] = tp.pass_a ? gA : 0.0f;
        grad_out[(b * H + t) * 2 + 1] = tp.pass_w ? (Ath * dt) : 0.0f;
        Lx = Ax; Ly = Ay;
        Lv = (gv1 + Av) + gv3;
        Lth = Ath + tau;
    }
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

// The two-wide cores of the reward features beside their scalar forms, on caller-supplied operands: pair i is
// (num[2i], num[2i+1]) / (den[2i], den[2i+1]) and exp of (x[2i], x[2i+1]).  Same inline-asm sequences as the planner
// kernels run (div2_: unpadded packed Newton-Raphson steps; exp_le1_2: unpadded packed reduction + polynomial).
__global__ void packed_math_kernel(const float *num, const float *den, const float *x, float *div_scalar,
                                   float *div_packed, float *exp_scalar, float *exp_packed, long long n_pairs)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const PkConsts pk = pk_consts();
    if (num && den) {
        const v2f nn{num[2 * i], num[2 * i + 1]}, dd{den[2 * i], den[2 * i + 1]};
        const v2f q = div2_(nn, dd);
        if (div_packed) { div_packed[2 * i] = q.x; div_packed[2 * i + 1] = q.y; }
        if (div_scalar) { div_scalar[2 * i] = nn.x / dd.x; div_scalar[2 * i + 1] = nn.y / dd.y; }
    }
    if (x) {
        const v2f xx{x[2 * i], x[2 * i + 1]};
        const v2f e = exp_le1_2(xx, pk);
        if (exp_packed) { exp_packed[2 * i] = e.x; exp_packed[2 * i + 1] = e.y; }
        if (exp_scalar) { exp_scalar[2 * i] = exp_le1(xx.x); exp_scalar[2 * i + 1] = exp_le1(xx.y); }
    }
}

// the shortened divisions on caller-supplied operands (ocd_debug_guarded_division)
__global__ void guarded_division_kernel(const float *u, const float *n, const float *w, float *m_out, float *k_out,
                                        float *q_out, long long n_pairs)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    if (u) {
        v2f m, k;
        recip_pair_guarded(v2f{u[2 * i], u[2 * i + 1]}, m, k);
        if (m_out) { m_out[2 * i] = m.x; m_out[2 * i + 1] = m.y; }
        if (k_out) { k_out[2 * i] = k.x; k_out[2 * i + 1] = k.y; }
    }
    if (n && w) {
        const v2f ww{w[2 * i], w[2 * i + 1]};
        const v2f q = quot2_by_recip(v2f{n[2 * i], n[2 * i + 1]}, ww, v2f{refined_recip(ww.x), refined_recip(ww.y)});
        if (q_out) { q_out[2 * i] = q.x; q_out[2 * i + 1] = q.y; }
    }
}

} // namespace ocd

// ---------------------------------------------------------------- launch table
namespace ocd {

static long long ceil_div(long long a, long long b) { return (a + b - 1) / b; }
static int clampi(long long v, long long lo, long long hi) { return (int)(v < lo ? lo : (v > hi ? hi : v)); }

// Variant and packing, from measurements on MI355X (tools/sweep.py).  What decides
// the kernel time is how many wavefronts the busiest SIMD gets and a wavefront's own instruction stream:
//   * trajectories x K <= SIMDs: V_ROW, one trajectory per wavefront (every initialisation decides its
//     feature skips alone; K wavefronts meet once per control step);
//   * V_SEG (single-wavefront workgroups spread evenly over the SIMDs, no workgroup barrier) while it gives
//     at most one wavefront per SIMD, or up to four when it packs lanes as densely as V_LDS (K*H close to 64);
//   * V_CHUNK (where compiled) as soon as V_LDS would need more than one wavefront per SIMD, and at every size where
//     neither V_ROW nor V_SEG exists (long horizons): a lane owns S steps,
//     the wavefront S times the trajectories; launch_chunk_dispatch picks S by cost -- in effect the smallest compiled S
//     whose wavefronts fit one per SIMD, then the one with the least work on the busiest SIMD;
//   * else V_LDS, densest packing at one lane per step, LDS latency hidden by the other wavefronts.
// scan_mode 1..4 and segs_per_wave force the choice (tests, sweeps).
template <int HT, int NO, int L>
static hipError_t launch_mpc(const KernelParams &p_in, hipStream_t st)
{
    KernelParams p = p_in;
    const int H = p.d.horizon, K = p.K;
    const Geo G = geo_lds(H);
    const long long cus = p.n_cus > 0 ? p.n_cus : 256;
    const long long simds = 4 * cus;
    const long long n = p.n_problems;
    const int seg_cap = (HT > 0 && K * H <= 64) ? 64 / (K * H) : 0;    // V_SEG trajectories per wavefront
    const int row_cap = (HT > 0 && H <= 16) ? 4 : 0;                   // V_ROW
    int chunk = 0;                                                     // V_CHUNK: chunk size compiled for this shape
    (void)launch_chunk_dispatch(H, NO, L, p, st, false, p.chunk_size, &chunk);
    if (chunk && 64 / (K * ceil_div(H, chunk)) < 1) chunk = 0;
    // terminal value: the generic kernel (run-time H, V_LDS) for every shape; V_ROW / V_SEG LAT builds for the shapes of
    // OCD_LEAF_TABLE while the batch is small enough for them (they are latency builds) and no diagnostics knob is set
    constexpr bool leaf_fast = leaf_specialised<HT, NO, L>::value;
    const bool leaf = p.leaf.values != nullptr;
    const bool lat = L > 0 && NO > 0 && !p.no_skips && !p.no_unify && !p.no_latency_build;
    const size_t leaf_lds = leaf ? (size_t)(p.leaf.n[0] + p.leaf.n[1] + p.leaf.n[2]) * sizeof(float) : 0;
    if (leaf && !(leaf_fast && lat)) {
        chunk = 0;                                                     // (the chunked kernel carries no terminal value)
        p.scan_mode = 1;
    } else if (leaf) {
        chunk = 0;
        if (p.scan_mode == 4) p.scan_mode = 0;
    }
    int variant = V_LDS;
    if (p.scan_mode == 4 && chunk) variant = V_CHUNK;
    else if (p.scan_mode == 2 && row_cap) variant = V_ROW;
    else if (p.scan_mode == 3 && seg_cap) variant = V_SEG;
    else if (p.scan_mode == 0) {
        const long long waves_seg = seg_cap ? ceil_div(n, seg_cap) : 0;
        const long long waves_lds = ceil_div(n, G.SEGS) * K;
        // (round-3 sweeps, tools/sweep_sizes.sh: once the one-lane-per-step mappings need a second wavefront on a SIMD,
        //  the chunked mapping with the smallest chunk that still fits one wavefront per SIMD is ahead at every horizon;
        //  the one exception found, 6 144 trajectories at H = 10 where V_LDS is 7 % faster, is not worth a rule)
        //  At horizons with neither DPP mapping (H > 16 and K*H > 64) the chunked kernel is ahead of V_LDS at every size
        //  (H = 25, 128 ... 1 024 trajectories: 3.84-4.06 ms with S = 2 against 4.52-4.94).
        const bool chunk_wins = chunk && (waves_lds > simds || (!row_cap && !seg_cap));
        // V_ROW puts the K wavefronts of a trajectory into ONE workgroup, i.e. on one compute unit: with K = 6
        // (extra_inits) two of its four SIMDs hold two wavefronts and the workgroup waits for them at every control
        // step -- the reference's validation shapes (27 episodes, K = 6) ran 1.7 x slower than with all K
        // initialisations in one wavefront (round 4, tools/small_shapes.py: H = 6 4.40 -> 2.60 ms, H = 5 2.12 -> 1.26)
        // ... and with K <= 4 a compute unit holds floor(4 / K) such workgroups before two wavefronts share a SIMD: 257 ... 341
        // trajectories at K = 3 fit the SIMD count but not the compute units -- a quarter of them ran two workgroups, 6
        // wavefronts on 4 SIMDs, and the launch took 1.96 ms where V_SEG (one wavefront per trajectory, any SIMD) takes 1.17
        // (round 6, tools/two_stream_probe.py 12: 324 episodes of the reference's shape, the lockstep path's launches)
        const bool row_fits_cu = (K <= 4 && (n <= cus * (4 / K) || !seg_cap)) || !seg_cap;
        if (row_cap && n * K <= simds && row_fits_cu) variant = V_ROW;
        else if (seg_cap && waves_seg <= simds) variant = V_SEG;
        else if (chunk_wins) variant = V_CHUNK;
        else if (seg_cap && waves_seg * 20 <= waves_lds * 21 && waves_seg <= 4 * simds) variant = V_SEG;
        else if (row_cap && ceil_div(n, row_cap) * K <= simds + simds / 2) variant = V_ROW;   // (2 % ahead of V_LDS at 1.5 per SIMD)
    }
    if (variant == V_CHUNK) return launch_chunk_dispatch(H, NO, L, p, st, true, p.chunk_size, &chunk);
    int segs;
    if (variant == V_ROW) segs = p.segs_used > 0 ? clampi(p.segs_used, 1, row_cap) : clampi(ceil_div(n * K, simds), 1, row_cap);
    else if (variant == V_SEG) segs = p.segs_used > 0 ? clampi(p.segs_used, 1, seg_cap) : clampi(ceil_div(n, simds), 1, seg_cap);
    else segs = p.segs_used > 0 ? clampi(p.segs_used, 1, G.SEGS) : clampi(ceil_div(n, cus), 1, G.SEGS);
    p.segs_used = segs;
    const unsigned blocks = (unsigned)ceil_div(n, segs);
    // LAT builds: lane features, ONE wavefront per SIMD (they claim it, see the kernel), no diagnostics knob set.  (Until
    // round 5 V_SEG took its LAT build up to four per SIMD: -2.6 % at two, -1 % at four at config 3's shape; those sizes
    // go to the chunked kernel wherever one is compiled.)
    if constexpr (leaf_fast) {
        if (leaf && variant == V_SEG && blocks <= simds) {
            p.leaf.grid_in_lds = 1;
            note_launch(p, 3, 0, segs, blocks, 1, HT, 1, 1);
            OCD_LAUNCH((mpc_kernel<HT, NO, L, V_SEG, true, true>), dim3(blocks), dim3(64), leaf_lds, st, p);
            return launch_status(p);
        }
        if (leaf && variant == V_ROW && (long long)blocks * K <= simds && K <= 4) {
            const size_t lds = (size_t)2 * K * G.SEL_FLOATS * sizeof(float) + leaf_lds;
            p.leaf.grid_in_lds = 1;
            note_launch(p, 2, 0, segs, blocks, 1, HT, 1, K);
            OCD_LAUNCH((mpc_kernel<HT, NO, L, V_ROW, true, true>), dim3(blocks), dim3(64 * K), lds, st, p);
            return launch_status(p);
        }
    }
    if (leaf) {                                                        // generic kernel, densest packing
        p.segs_used = p_in.segs_used > 0 ? clampi(p_in.segs_used, 1, G.SEGS) : clampi(ceil_div(n, cus), 1, G.SEGS);
        const unsigned gblocks = (unsigned)ceil_div(n, p.segs_used);
        const size_t base = ((size_t)K * G.WAVE_FLOATS + (size_t)2 * K * G.SEL_FLOATS) * sizeof(float);
        p.leaf.grid_in_lds = (base + leaf_lds <= 64 * 1024) ? 1 : 0;   // (the default dynamic-LDS limit of a launch)
        note_launch(p, 1, 0, p.segs_used, gblocks, 0, 0, 1, K);
        OCD_LAUNCH((mpc_kernel<0, NO, L, V_LDS, true>), dim3(gblocks), dim3(64 * K), base + (p.leaf.grid_in_lds ? leaf_lds : 0), st, p);
        return launch_status(p);
    }
    if constexpr (HT > 0) {
        if (variant == V_SEG) {
            if constexpr (HT * 3 <= 64) {
                note_launch(p, 3, 0, segs, blocks, (lat && blocks <= simds) ? 1 : 0, HT, 0, 1);
                if (lat && blocks <= simds) OCD_LAUNCH((mpc_kernel<HT, NO, L, V_SEG, false, true>), dim3(blocks), dim3(64), 0, st, p);
                else OCD_LAUNCH((mpc_kernel<HT, NO, L, V_SEG>), dim3(blocks), dim3(64), 0, st, p);
            }
            return launch_status(p);
        }
        if (variant == V_ROW) {
            const size_t lds = (size_t)2 * K * G.SEL_FLOATS * sizeof(float);
            if constexpr (HT <= 16) {
                // (the latency build claims a SIMD per wavefront: a workgroup of more than four cannot be placed at all)
                const bool row_lat = lat && (long long)blocks * K <= simds && K <= 4;
                note_launch(p, 2, 0, segs, blocks, row_lat ? 1 : 0, HT, 0, K);
                if (row_lat) OCD_LAUNCH((mpc_kernel<HT, NO, L, V_ROW, false, true>), dim3(blocks), dim3(64 * K), lds, st, p);
                else OCD_LAUNCH((mpc_kernel<HT, NO, L, V_ROW>), dim3(blocks), dim3(64 * K), lds, st, p);
            }
            return launch_status(p);
        }
    }
    const size_t lds = ((size_t)K * G.WAVE_FLOATS + (size_t)2 * K * G.SEL_FLOATS) * sizeof(float);
    note_launch(p, 1, 0, segs, blocks, 0, HT, 0, K);
    OCD_LAUNCH((mpc_kernel<HT, NO, L, V_LDS>), dim3(blocks), dim3(64 * K), lds, st, p);
    return launch_status(p);
}

#define OCD_CASE(HH, NN, LL) if (H == HH && NO == NN && L == LL) return launch_mpc<HH, NN, LL>(p, st);
#define OCD_GCASE(NN, LL) if (NO == NN && L == LL) return launch_mpc<0, NN, LL>(p, st);

hipError_t launch_mpc_dispatch(int H, int NO, int L, const KernelParams &p_in, hipStream_t st, bool *supported)
{
    const KernelParams &p = p_in;
    *supported = true;
    if (!p.two_sided) { OCD_KERNEL_TABLE(OCD_CASE) }           // (a degenerate fence runs the generic kernel: ocd_kernels.h)
    OCD_PAIR_TABLE(OCD_GCASE)
    *supported = false;
    return hipSuccess;
}

#define OCD_RCASE(NN, LL) if (NO == NN && L == LL) { hipLaunchKernelGGL((reward_kernel<NN, LL>), dim3(nb), dim3(bs), 0, st, p, feats, rew); return hipGetLastError(); }

hipError_t launch_reward(int NO, int L, const KernelParams &p, float *feats, float *rew, hipStream_t st, bool *supported)
{
    *supported = true;
    const unsigned bs = 256;
    const unsigned nb = (unsigned)((p.n_problems + bs - 1) / bs);
    OCD_PAIR_TABLE(OCD_RCASE)
    *supported = false;
    return hipSuccess;
}

#define OCD_OCASE(NN, LL) if (NO == NN && L == LL) { hipLaunchKernelGGL((objective_kernel<NN, LL>), dim3(nb), dim3(bs), 0, st, p, controls, reward_out, grad_out, traj_out); return hipGetLastError(); }

hipError_t launch_objective(int NO, int L, const KernelParams &p, const float *controls, float *reward_out,
                            float *grad_out, float *traj_out, hipStream_t st, bool *supported)
{
    *supported = true;
    const unsigned bs = 64;
    const unsigned nb = (unsigned)((p.n_problems + bs - 1) / bs);
    OCD_PAIR_TABLE(OCD_OCASE)
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

hipError_t launch_packed_math(const float *num, const float *den, const float *x, float *div_scalar, float *div_packed,
                              float *exp_scalar, float *exp_packed, long long n_pairs, hipStream_t st)
{
    const unsigned bs = 64;                      // one wavefront per workgroup, as the planner kernels run them
    const unsigned nb = (unsigned)((n_pairs + bs - 1) / bs);
    hipLaunchKernelGGL(packed_math_kernel, dim3(nb), dim3(bs), 0, st, num, den, x, div_scalar, div_packed, exp_scalar,
                       exp_packed, n_pairs);
    return hipGetLastError();
}

hipError_t launch_guarded_division(const float *u, const float *n, const float *w, float *m_out, float *k_out, float *q_out,
                                   long long n_pairs, hipStream_t st)
{
    const unsigned bs = 64;
    const unsigned nb = (unsigned)((n_pairs + bs - 1) / bs);
    hipLaunchKernelGGL(guarded_division_kernel, dim3(nb), dim3(bs), 0, st, u, n, w, m_out, k_out, q_out, n_pairs);
    return hipGetLastError();
}

} // namespace ocd
