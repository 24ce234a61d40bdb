/*
 * oracle/ocd_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's planner path, one IEEE binary32 rounding
 * per TensorFlow op, sequential in the horizon exactly like the traced graph.
 * Build with -ffp-contract=off (see oracle/Makefile): no implicit FMA.
 *
 * Reference files followed (relative to the reference tree):
 *   interact_drive/simulation_utils.py:9-21     car_dynamics_step
 *   interact_drive/planner/naive_planner.py:33-77,81-164  mpc_reward, generate_plan
 *   experiments/merging.py:32-83                ThreeLaneTestCar.features
 *   interact_drive/math_utils.py:7-31,59-97,135-180  _f, smooth_threshold, smooth_bump
 *   interact_drive/world.py:79-109,143-159,206-218  CarWorld.step, lanes, dist2median
 *   interact_drive/car/linear_reward_car.py:49-55   reward_fn
 *   interact_drive/car/planner_car.py:54-85     _get_next_control
 *   interact_drive/car/car.py:70-87, fixed_plan_car.py:25-39, fixed_control_car.py:26-29
 *   experiments/replanning_world.py:24-36       ReplanningCarWorld.reset/step
 *   interact_drive/reward_design/mpc_ord.py:67-106  eval_weights_for_init
 *
 * Pinning.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / untimed parity legs
 * load this file; the product never does.  PINNED by every known answer the reference's own tests hold for this path
 * (tests/test_oracle_kat.py): the four dynamics KATs, the _f / smooth_threshold / smooth_bump doctests, the reward doctest,
 * both planner KATs with the first-index tie-break, the other_controls layout, the IOC tests' planning car.
 * PARITY UNPINNED by the reference for ThreeLaneTestCar.features, the scenario constants and every episode return: the
 * reference holds no test or fixture for them and TensorFlow cannot run here (no oracle/_ref: nothing to compile).  For
 * those the evidence is second opinions written from the reference's Python (tests/golden/torch_restatement.py,
 * tests/golden/torch_episode.py: float64 torch), not reference outputs.
 *
 * Reverse mode: the adjoint below is the tape of the forward ops reversed,
 * with TensorFlow's gradient rules (math_grad.py of TF 2.1):
 *   Minimum(x,y): to x iff x<=y;  Maximum(x,y): to x iff x>=y;
 *   Min/Max reduction: (indicator/num_ties)*grad  [math_ops.divide(indicators, num_selected) * grad];  Select: chosen branch only;
 *   RealDiv(x,y): gx = g/y, gy = g*((-x/y)/y);  Pow(x,2): (g*2)*x;
 *   Exp: g*y;  Sin: g*cos(x);  Cos: (-g)*sin(x);  Abs: g*sign(x).
 * Where a value has three or more consumers the order in which TensorFlow
 * sums their gradients is not knowable without TensorFlow; the fixed order
 * used here is documented in DESIGN.md section 3 and is part of the
 * arithmetic contract shared with the HIP kernels.
 */
#include "ocd_oracle.h"
#include "ocd_refmath.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXH 64
#define MAXO OCD_MAX_OTHERS

static inline float fminf_tf(float a, float b) { return (a <= b) ? a : b; }
static inline float fmaxf_tf(float a, float b) { return (a >= b) ? a : b; }

int32_t ocd_oracle_uses_libm(void)
{
#ifdef OCD_USE_LIBM
    return 1;
#else
    return 0;
#endif
}

float ocd_oracle_expf(float x) { return ocd_ref_expf(x); }
float ocd_oracle_sinf(float x) { float s, c; ocd_ref_sincosf(x, &s, &c); return s; }
float ocd_oracle_cosf(float x) { float s, c; ocd_ref_sincosf(x, &s, &c); return c; }

/* ---- math_utils._f (math_utils.py:28-31) with what its backward pass needs ---- */
typedef struct { int pos; float m, e, u; } f_tape;

static inline float f_fwd(float t, float shape, f_tape *tp)
{
    const int pos = t > 0.0f;
    const float tc = pos ? t : (0.0f + 0.01f);
    const float u = shape * tc;
    const float m = -1.0f / u;
    const float e = ocd_ref_expf(m);
    tp->pos = pos; tp->m = m; tp->e = e; tp->u = u;
    return pos ? e : 0.0f;
}

/* d F / d t applied to upstream g */
static inline float f_bwd(float g, float shape, const f_tape *tp)
{
    const float g_e = tp->pos ? g : 0.0f;
    const float g_m = g_e * tp->e;
    const float g_u = g_m * ((-tp->m) / tp->u);   /* g * ((-(-1)/u)/u) */
    const float g_tc = g_u * shape;
    return tp->pos ? g_tc : 0.0f;
}

float ocd_oracle_f(float x, float shape) { f_tape tp; return f_fwd(x, shape, &tp); }

/* ---- smooth_threshold (math_utils.py:87-95) ---- */
typedef struct { f_tape t1, t2; float F1, den, S; } thr_tape;

static inline float thr_fwd(float z, float lo, float width, float shape, thr_tape *tp)
{
    const float xd = z - lo;
    const float F1 = f_fwd(xd, shape, &tp->t1);
    const float xd2 = width - xd;
    const float F2 = f_fwd(xd2, shape, &tp->t2);
    const float den = F1 + F2;
    const float S = F1 / den;
    tp->F1 = F1; tp->den = den; tp->S = S;
    return S;
}

static inline float thr_bwd(float g_S, float shape, const thr_tape *tp)
{
    const float g_F1a = g_S / tp->den;
    const float g_den = g_S * ((-tp->S) / tp->den);  /* g * ((-F1/den)/den), F1/den == S */
    const float ga = f_bwd(g_F1a, shape, &tp->t1);   /* numerator's _f(x_diff) call   */
    const float gb = f_bwd(g_den, shape, &tp->t1);   /* denominator's _f(x_diff) call */
    const float gc = f_bwd(g_den, shape, &tp->t2);   /* _f(width - x_diff)            */
    return (ga + gb) + (-gc);
}

float ocd_oracle_smooth_threshold(float x, float lo, float width, float shape)
{
    thr_tape tp; return thr_fwd(x, lo, width, shape, &tp);
}

/* ---- smooth_bump (math_utils.py:166-178) ---- */
typedef struct { int cond; float xc, q, m, e, width; } bump_tape;

static inline float bump_fwd(float z, float start, float end, bump_tape *tp)
{
    const float width = (end - start) / 2.0f;
    const float center = (start + end) / 2.0f;
    const float zn = (z - center) / width;
    const int cond = (zn * zn) < 1.0f;
    const float xc = cond ? zn : 0.0f;
    const float q = 1.0f - xc * xc;
    const float m = -1.0f / q;
    const float arg = m + 1.0f;
    const float e = ocd_ref_expf(arg);
    tp->cond = cond; tp->xc = xc; tp->q = q; tp->m = m; tp->e = e; tp->width = width;
    return cond ? e : 0.0f;
}

static inline float bump_bwd(float g, const bump_tape *tp)
{
    const float g_e = tp->cond ? g : 0.0f;
    const float g_arg = g_e * tp->e;
    const float g_q = g_arg * ((-tp->m) / tp->q);
    const float g_xc2 = -g_q;
    const float g_xc = (g_xc2 * 2.0f) * tp->xc;
    const float g_zn = tp->cond ? g_xc : 0.0f;
    return g_zn / tp->width;
}

float ocd_oracle_smooth_bump(float x, float start, float end)
{
    bump_tape tp; return bump_fwd(x, start, end, &tp);
}

/* ---- car_dynamics_step (simulation_utils.py:9-21) ---- */
typedef struct {
    float v, c, s, d, acc;     /* pre-step speed, cos/sin of pre-step heading, distance, total_acc */
    int pass_a, pass_w;        /* clip gates of tf.minimum / tf.maximum */
} dyn_tape;

static inline void dyn_fwd(float x, float y, float v, float th, float c, float s,
                           float a, float w, float dt, float dt2, float f,
                           float *xn, float *yn, float *vn, float *thn, dyn_tape *tp)
{
    const float a1 = fminf_tf(a, 4.0f);
    const float a_c = fmaxf_tf(a1, -8.0f);
    const float w1 = fminf_tf(w, 4.0f);
    const float w_c = fmaxf_tf(w1, -4.0f);
    const float v2 = v * v;
    const float fv2 = f * v2;
    const float acc = a_c - fv2;
    const float vdt = v * dt;
    const float hA = 0.5f * acc;
    const float hAdt2 = hA * dt2;
    const float d = vdt + hAdt2;
    *xn = x + c * d;
    *yn = y + s * d;
    *vn = v + acc * dt;
    *thn = th + w_c * dt;
    if (tp) {
        tp->v = v; tp->c = c; tp->s = s; tp->d = d; tp->acc = acc;
        tp->pass_a = (a <= 4.0f) && (a1 >= -8.0f);
        tp->pass_w = (w <= 4.0f) && (w1 >= -4.0f);
    }
}

void ocd_oracle_dynamics_step(const float *st, const float *u, float dt, float dt2, float friction, float *out)
{
    float s, c;
    ocd_ref_sincosf(st[3], &s, &c);
    dyn_fwd(st[0], st[1], st[2], st[3], c, s, u[0], u[1], dt, dt2, friction,
            &out[0], &out[1], &out[2], &out[3], NULL);
}

/* ---- reward of one world state and its gradient w.r.t. the ego state ---- */
typedef struct { float qx, qy, qv, qth; } q4;

/* scored: the reward as car.reward_fn(past_state, ...) evaluates it when an episode is scored (mpc_ord.py:99) -- the lane
 * offset carries dist2median's y-term (world.py:216-217); 0: the planner's objective (naive_planner.py:33-77), which
 * keeps r = (x - p0) * -1 -- the same bits for every finite y (include/ocd.h, ABI 3). */
static float reward_state(const ocd_scenario_desc *d, const float *w,
                          float x, float y, float v, float sn, float cn,
                          const float (*oxy)[2], float *feats, q4 *q, int scored)
{
    if (d->reward_kind == OCD_REWARD_TARGET_SPEED) {
        /* r = 0; r -= (velocity - target) ** 2  (targetSpeedRewardMaximizerCar.py:50-56) */
        const float dv = v - d->target_speed;
        const float sq = dv * dv;
        if (q) { q->qx = 0.0f; q->qy = 0.0f; q->qth = 0.0f; q->qv = (-1.0f * 2.0f) * dv; }
        return 0.0f - sq;
    }
    if (d->reward_kind == OCD_REWARD_LINEAR_TARGET_SPEED) {
        /* features = stack([velocity, (velocity - target) ** 2]); reward = reduce_sum(weights * features)
         * (linearTargetSpeedPlannerCar.py:36-44, linear_reward_car.py:49-55) */
        const float dv = v - d->target_speed;
        const float sq = dv * dv;
        float r = w[0] * v;
        r = r + w[1] * sq;
        if (feats) { feats[0] = v; feats[1] = sq; }
        if (q) { q->qx = 0.0f; q->qy = 0.0f; q->qth = 0.0f; q->qv = w[0] + (w[1] * 2.0f) * dv; }
        return r;
    }
    const int L = d->n_lanes, D = L + 4, NO = d->n_cars - 1;
    float phi[OCD_MAX_FEATURES];

    /* phi_0 (merging.py:48-53) */
    const float tgt = d->target_speed;
    const float bound = 4.0f * (tgt * tgt);
    const float vel = v * sn;
    const float dv = vel - tgt;
    const float sq = dv * dv;
    const int pass0 = sq <= bound;
    phi[0] = fminf_tf(sq, bound);

    /* lanes (merging.py:55-59, world.py:216-218): r = (x-p0)*n0 + (y-p1)*n1, n = (-1, 0) */
    float rl[OCD_MAX_LANES];
    for (int l = 0; l < L; ++l) {
        const float diff = x - d->lane_center[l];
        rl[l] = diff * -1.0f;
        if (scored) rl[l] = rl[l] + (y - d->lane_origin_y) * d->lane_normal_y;   /* +-0, or NaN beyond the finite numbers */
        const float d2 = rl[l] * rl[l];
        phi[1 + l] = d2 * 10.0f;
    }
    float pmin = 0.0f; int ntie_min = 0;
    if (L > 0) {
        pmin = phi[1];
        for (int l = 1; l < L; ++l) pmin = fminf_tf(pmin, phi[1 + l]);
        for (int l = 0; l < L; ++l) ntie_min += (phi[1 + l] == pmin);
    }
    phi[L + 1] = pmin;

    /* collision (merging.py:61-78) */
    bump_tape bx[MAXO], by[MAXO];
    float bxv[MAXO], byv[MAXO], col[MAXO];
    float pcol = 0.0f; int ntie_col = 0;
    for (int j = 0; j < NO; ++j) {
        const float ox = oxy[j][0], oy = oxy[j][1];
        bxv[j] = bump_fwd(x, ox - d->bump_half_x, ox + d->bump_half_x, &bx[j]);
        byv[j] = bump_fwd(y, oy - d->bump_half_y, oy + d->bump_half_y, &by[j]);
        col[j] = bxv[j] * byv[j];
        pcol = (j == 0) ? col[0] : fmaxf_tf(pcol, col[j]);
    }
    for (int j = 0; j < NO; ++j) ntie_col += (col[j] == pcol);
    phi[L + 2] = pcol;

    /* fences (merging.py:80-82) */
    thr_tape tp_p, tp_m;
    const float Sp = thr_fwd(x, d->fence_lo, d->fence_width, d->fence_shape, &tp_p);
    const float Sm = thr_fwd(-x, d->fence_lo, d->fence_width, d->fence_shape, &tp_m);
    const float Ssum = Sp + Sm;
    const float ax = (x < 0.0f) ? -x : x;
    phi[L + 3] = Ssum * ax;

    /* reward = reduce_sum(weights * feats) (linear_reward_car.py:52-55), left to right */
    float r = w[0] * phi[0];
    for (int k = 1; k < D; ++k) r = r + w[k] * phi[k];
    if (feats) for (int k = 0; k < D; ++k) feats[k] = phi[k];
    if (!q) return r;

    /* ---- backward: d r / d (x, y, v, heading), upstream gradient 1 ---- */
    const float g_sq = pass0 ? w[0] : 0.0f;
    const float g_dv = (g_sq * 2.0f) * dv;
    q->qv = g_dv * sn;
    const float g_sn = g_dv * v;
    q->qth = g_sn * cn;

    float qx = 0.0f, qy = 0.0f;
    for (int l = 0; l < L; ++l) {
        float g = w[1 + l];
        if (phi[1 + l] == pmin) g = g + (1.0f / (float)ntie_min) * w[L + 1];
        const float g_d2 = g * 10.0f;
        const float g_r = (g_d2 * 2.0f) * rl[l];
        qx = qx + g_r * -1.0f;
    }
    for (int j = 0; j < NO; ++j) {
        const float share = (col[j] == pcol) ? ((1.0f / (float)ntie_col) * w[L + 2]) : 0.0f;
        const float g_bx = share * byv[j];
        const float g_by = share * bxv[j];
        qx = qx + bump_bwd(g_bx, &bx[j]);
        qy = qy + bump_bwd(g_by, &by[j]);
    }
    const float g_Ssum = w[L + 3] * ax;
    const float g_ax = w[L + 3] * Ssum;
    qx = qx + thr_bwd(g_Ssum, d->fence_shape, &tp_p);
    qx = qx + (-thr_bwd(g_Ssum, d->fence_shape, &tp_m));
    const float sgn = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
    qx = qx + g_ax * sgn;
    q->qx = qx; q->qy = qy;
    return r;
}

/* ---- terminal value: ValueFeature.interpolate_value (value_interpolation.py:28-61) ----
 * Trilinear interpolation of a value table over the coarse state proj(world_state); outside the grid the
 * value is NaN and its gradient zero (the else-branch returns a constant).  The reference never calls it (no scenario sets leaf_evaluation) and holds no vector for it:
 * PARITY UNPINNED; the op order below (one rounding per TensorFlow op, sum over the 8 corners in
 * itertools.product order, gradient accumulated in that same order) is the contract the HIP kernel shares. */
static struct {
    const float *grid[3];
    const float *values;
    int n[3];
    int proj_kind;
} g_leaf = {{0, 0, 0}, 0, {0, 0, 0}, 0};

void ocd_oracle_set_leaf_value(const float *grid0, int32_t n0, const float *grid1, int32_t n1,
                               const float *grid2, int32_t n2, const float *values, int32_t proj_kind)
{
    g_leaf.grid[0] = grid0; g_leaf.grid[1] = grid1; g_leaf.grid[2] = grid2;
    g_leaf.n[0] = n0; g_leaf.n[1] = n1; g_leaf.n[2] = n2;
    g_leaf.values = values;
    g_leaf.proj_kind = proj_kind;
}

static float leaf_value(float x, float y, float v, float sn, float cn, q4 *q)
{
    const float xc[3] = {x, y, (g_leaf.proj_kind == 1) ? (v * sn) : v};
    const float nanv = 0.0f / 0.0f;
    int inside = 1;
    for (int k = 0; k < 3; ++k)
        inside = inside && (xc[k] >= g_leaf.grid[k][0]) && (xc[k] <= g_leaf.grid[k][g_leaf.n[k] - 1]);
    /* outside the grid the traced function returns the CONSTANT float('nan') (value_interpolation.py:59-60): the
     * value is NaN, its gradient w.r.t. the state is zero -- the other horizon steps keep their finite gradients */
    if (q) { q->qx = 0.0f; q->qy = 0.0f; q->qv = 0.0f; q->qth = 0.0f; }
    if (!inside) return nanv;
    int c[3];
    float a[3], st[3], p[3][2];
    for (int k = 0; k < 3; ++k) {
        /* tf.where(grid <= x)[-1]: the last boundary not above x; clamped so that corner + 1 exists */
        int ci = 0;
        for (int i = 0; i < g_leaf.n[k]; ++i) if (g_leaf.grid[k][i] <= xc[k]) ci = i;
        if (ci > g_leaf.n[k] - 2) ci = g_leaf.n[k] - 2;
        c[k] = ci;
        st[k] = g_leaf.grid[k][ci + 1] - g_leaf.grid[k][ci];
        a[k] = xc[k] - g_leaf.grid[k][ci];
        p[k][0] = (-1.0f * a[k]) + (1.0f * st[k]);      /* (-1)**(i+1) * (x - g) + (1 - i) * step, i = 0 */
        p[k][1] = (1.0f * a[k]) + (0.0f * st[k]);       /* i = 1 */
    }
    const float cell = (st[0] * st[1]) * st[2];
    float sum = 0.0f, ga[3] = {0.0f, 0.0f, 0.0f};
    for (int i0 = 0; i0 < 2; ++i0)
        for (int i1 = 0; i1 < 2; ++i1)
            for (int i2 = 0; i2 < 2; ++i2) {
                const float val = g_leaf.values[((size_t)(c[0] + i0) * g_leaf.n[1] + (c[1] + i1)) * g_leaf.n[2] + (c[2] + i2)];
                const float pv01 = p[0][i0] * p[1][i1];
                const float pv = pv01 * p[2][i2];
                const float num = val * pv;
                sum = sum + num / cell;
                if (q) {
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
    if (q) {
        q->qx = ga[0];
        q->qy = ga[1];
        if (g_leaf.proj_kind == 1) { q->qv = ga[2] * sn; q->qth = (ga[2] * v) * cn; }
        else { q->qv = ga[2]; q->qth = 0.0f; }
    }
    return sum;
}

float ocd_oracle_reward(const ocd_scenario_desc *d, const float *ws, const float *w,
                        float *feats_out, float *grad_out)
{
    float oxy[MAXO][2];
    for (int j = 1; j < d->n_cars; ++j) { oxy[j - 1][0] = ws[4 * j]; oxy[j - 1][1] = ws[4 * j + 1]; }
    float s, c;
    ocd_ref_sincosf(ws[3], &s, &c);
    q4 q;
    /* without a gradient this is reward_fn as the episode scoring and the heat map call it (scored form); with one, the
     * planner's form */
    const float r = reward_state(d, w, ws[0], ws[1], ws[2], s, c, (const float (*)[2])oxy,
                                 feats_out, grad_out ? &q : NULL, grad_out == NULL);
    if (grad_out) { grad_out[0] = q.qx; grad_out[1] = q.qy; grad_out[2] = q.qv; grad_out[3] = q.qth; }
    return r;
}

/* ---- the planner's model of the scripted cars over the horizon (naive_planner.py:51-66) ----
 * oxy[t][j] = (x, y) of scripted car j AFTER horizon step t. */
static void predict_others(const ocd_scenario_desc *d, const float *ws, const float *other_plans,
                           float oxy[][MAXO][2])
{
    const int H = d->horizon, NO = d->n_cars - 1;
    const float dt = d->dt, dt2 = d->dt_sq;
    for (int j = 0; j < NO; ++j) {
        float x = ws[4 * (j + 1)], y = ws[4 * (j + 1) + 1], v = ws[4 * (j + 1) + 2], th = ws[4 * (j + 1) + 3];
        for (int t = 0; t < H; ++t) {
            float s, c;
            ocd_ref_sincosf(th, &s, &c);
            if (other_plans) {
                const float acc = other_plans[(j * H + t) * 2], angv = other_plans[(j * H + t) * 2 + 1];
                const float dist = v * dt + (0.5f * acc) * dt2;
                x = x + c * dist;
                y = y + s * dist;
                v = v + acc * dt;
                th = th + angv * dt;
            } else {
                x = x + (c * v) * dt;
                y = y + (s * v) * dt;
                v = v + 0.0f;
                th = th + 0.0f;
            }
            oxy[t][j][0] = x; oxy[t][j][1] = y;
        }
    }
}

typedef struct {
    dyn_tape dyn;
    q4 q;
} step_tape;

/* mpc_reward forward (+ optional gradient w.r.t. controls) given the predicted scripted cars */
static float mpc_reward_core(const ocd_scenario_desc *d, const float *ego, const float *w,
                             const float *u /*[H,2]*/, float oxy[][MAXO][2],
                             float *grad /*[H,2] or NULL*/, float *traj /*[H,4] or NULL*/)
{
    const int H = d->horizon;
    const float dt = d->dt, dt2 = d->dt_sq, f = d->ego_friction;
    step_tape tape[MAXH];
    float x = ego[0], y = ego[1], v = ego[2], th = ego[3];
    float s, c;
    ocd_ref_sincosf(th, &s, &c);
    float R = 0.0f;
    for (int t = 0; t < H; ++t) {
        float xn, yn, vn, thn;
        dyn_fwd(x, y, v, th, c, s, u[2 * t], u[2 * t + 1], dt, dt2, f, &xn, &yn, &vn, &thn, &tape[t].dyn);
        float sn, cn;
        ocd_ref_sincosf(thn, &sn, &cn);
        float r;
        if (g_leaf.values && t == H - 1)        /* naive_planner.py:69-70 */
            r = leaf_value(xn, yn, vn, sn, cn, grad ? &tape[t].q : NULL);
        else
            r = reward_state(d, w, xn, yn, vn, sn, cn, (const float (*)[2])oxy[t], NULL,
                             grad ? &tape[t].q : NULL, 0);
        R = R + r;                              /* r = 0; r += reward_fn(...) */
        x = xn; y = yn; v = vn; th = thn; s = sn; c = cn;
        if (traj) { traj[4 * t] = x; traj[4 * t + 1] = y; traj[4 * t + 2] = v; traj[4 * t + 3] = th; }
    }
    if (!grad) return R;

    /* reverse sweep: L* = adjoint of the post-step state coming from later steps */
    float Lx = 0.0f, Ly = 0.0f, Lv = 0.0f, Lth = 0.0f;
    for (int t = H - 1; t >= 0; --t) {
        const dyn_tape *k = &tape[t].dyn;
        const q4 *q = &tape[t].q;
        const float Ax = q->qx + Lx, Ay = q->qy + Ly, Av = q->qv + Lv, Ath = q->qth + Lth;
        /* x' = x + c*d ; y' = y + s*d */
        const float g_c = Ax * k->d;
        const float g_s = Ay * k->d;
        const float g_d = Ax * k->c + Ay * k->s;
        const float tau = (-g_c) * k->s + g_s * k->c;          /* Cos and Sin gradients */
        /* d = v*dt + (0.5*acc)*dt2 ; v' = v + acc*dt */
        const float gv1 = g_d * dt;
        const float gA1 = (g_d * dt2) * 0.5f;
        const float gA2 = Av * dt;
        const float gA = gA1 + gA2;
        const float g_v2 = (-gA) * f;                          /* acc = a_c - f*v2 */
        const float gv3 = (g_v2 * 2.0f) * k->v;
        grad[2 * t] = k->pass_a ? gA : 0.0f;
        grad[2 * t + 1] = k->pass_w ? (Ath * dt) : 0.0f;       /* heading' = heading + w_c*dt */
        Lx = Ax; Ly = Ay;
        Lv = (gv1 + Av) + gv3;
        Lth = Ath + tau;
    }
    return R;
}

float ocd_oracle_mpc_reward(const ocd_scenario_desc *d, const float *ws, const float *w,
                            const float *u, const float *other_plans, float *grad_out, float *traj_out)
{
    float oxy[MAXH][MAXO][2];
    predict_others(d, ws, other_plans, oxy);
    return mpc_reward_core(d, ws, w, u, oxy, grad_out, traj_out);
}

/* ---- NaivePlanner.generate_plan (naive_planner.py:81-164) ---- */
/* extra_inits reads the CAR's current speed, `self.car.state[2]` (naive_planner.py:114), not the speed of the
 * init_state argument: they differ when generate_plan is called with a foreign init_state.  NULL: the world
 * state's own ego speed (what CarWorld.step always gives). */
static const float *g_init_speed = 0;
void ocd_oracle_set_init_speed(const float *init_speed /* [B] or NULL */) { g_init_speed = init_speed; }

static void plan_one(const ocd_scenario_desc *d, const float *ws, const float *w,
                     const float *other_plans, float *plan_out, float *best_loss, int32_t *best_init,
                     float *all_plans /*[K,H,2] or NULL*/, float *all_losses /*[K] or NULL*/, const float *init_speed)
{
    const int H = d->horizon, K = d->extra_inits ? 6 : 3;
    float oxy[MAXH][MAXO][2];
    predict_others(d, ws, other_plans, oxy);
    const float lr = d->learning_rate;
    const float turn = 0.65f;                       /* 5 * 0.13 */
    const float v_car = init_speed ? *init_speed : ws[2];
    const float a_coast = d->ego_friction * (v_car * v_car);  /* friction * self.car.state[2] ** 2 */
    float u[MAXH * 2], g[MAXH * 2], best_u[MAXH * 2];
    float bl = 0.0f; int bi = 0;
    for (int k = 0; k < K; ++k) {
        const float a0 = (k >= 3) ? a_coast : 0.0f;
        const float w0 = (k % 3 == 0) ? 0.0f : ((k % 3 == 1) ? -turn : turn);
        for (int t = 0; t < H; ++t) { u[2 * t] = a0; u[2 * t + 1] = w0; }
        for (int it = 0; it < d->n_iter; ++it) {
            mpc_reward_core(d, ws, w, u, oxy, g, NULL);
            /* SGD on loss = -R:  u <- u - lr*(-dR/du)  ==  u + lr*dR/du  (bitwise) */
            for (int i = 0; i < 2 * H; ++i) u[i] = u[i] + lr * g[i];
        }
        const float loss = -mpc_reward_core(d, ws, w, u, oxy, NULL, NULL);
        if (all_plans) memcpy(all_plans + (size_t)k * H * 2, u, sizeof(float) * 2 * H);
        if (all_losses) all_losses[k] = loss;
        if (k == 0 || loss < bl) {                  /* losses.index(min(losses)): first minimum */
            bl = loss; bi = k; memcpy(best_u, u, sizeof(float) * 2 * H);
        }
    }
    memcpy(plan_out, best_u, sizeof(float) * 2 * H);
    if (best_loss) *best_loss = bl;
    if (best_init) *best_init = bi;
}

static int check_desc(const ocd_scenario_desc *d)
{
    if (!d || d->abi_version != OCD_ABI_VERSION) return 0;
    if (d->n_cars < 1 || d->n_cars > OCD_MAX_CARS) return 0;
    if (d->horizon < 1 || d->horizon > MAXH) return 0;
    if (d->n_lanes < 0 || d->n_lanes > OCD_MAX_LANES) return 0;
    if (d->n_iter < 0 || d->episode_len < 0) return 0;
    if (d->n_samples < 1 || d->n_samples > OCD_MAX_SAMPLES) return 0;
    if (d->reward_kind == OCD_REWARD_LANE_FEATURES && (d->n_lanes < 1 || d->n_cars < 2)) return 0;
    if (d->reward_kind == OCD_REWARD_LANE_FEATURES && !(d->lane_normal_y == 0.0f)) return 0;   /* lanes run along y */
    return 1;
}

int32_t ocd_plan_batch_cpu(const ocd_scenario_desc *d, const float *world_state,
                           const float *weights, int32_t weights_per_problem,
                           const float *other_plans,
                           float *plans_out, float *best_loss_out, int32_t *best_init_out,
                           float *all_plans_out, float *all_losses_out,
                           int64_t B, int32_t n_threads)
{
    if (!check_desc(d) || !world_state || !plans_out || B < 0) return OCD_ERR_INVALID_ARG;
    const int C = d->n_cars, H = d->horizon, D = OCD_N_FEATURES(d->reward_kind, d->n_lanes), K = d->extra_inits ? 6 : 3;
    (void)n_threads;
#ifdef _OPENMP
    const int nt = n_threads > 0 ? n_threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
#endif
    for (int64_t b = 0; b < B; ++b) {
        const float *w = weights ? (weights + (weights_per_problem ? b * D : 0)) : NULL;
        plan_one(d, world_state + b * C * 4, w, other_plans, plans_out + b * H * 2,
                 best_loss_out ? best_loss_out + b : NULL, best_init_out ? best_init_out + b : NULL,
                 all_plans_out ? all_plans_out + b * K * H * 2 : NULL,
                 all_losses_out ? all_losses_out + b * K : NULL, g_init_speed ? g_init_speed + b : NULL);
    }
    return OCD_OK;
}

/* ---- control steps of the world (mpc_ord.py:87-104 over world.py:97-109) ----
 * ws: world state [C,4] at world step index t0 (updated in place); runs T steps. */
static float run_steps(const ocd_scenario_desc *d, float *ws, const float *w_plan, int sample, int t0, int T,
                       float *traj /*[T+1,C,4] or NULL*/, float *ctrl /*[T,2] or NULL*/)
{
    const int C = d->n_cars, H = d->horizon, NO = C - 1;
    const float dt = d->dt, dt2 = d->dt_sq;
    /* PlannerCar._get_next_control: plan[j] from index 0 at EVERY step (planner_car.py:66-75) */
    float oplans[MAXO * MAXH * 2];
    if (d->check_plans) {
        for (int j = 0; j < NO; ++j)
            for (int t = 0; t < H; ++t) {
                /* plan[j] if j < len(plan) else default_control if there is one else (0, 0) (planner_car.py:66-75) */
                const float *src = (t < d->other_plan_len[j]) ? d->other_plan[j][t] : d->other_assumed_default[j];
                oplans[(j * H + t) * 2] = src[0]; oplans[(j * H + t) * 2 + 1] = src[1];
            }
    }
    float G = 0.0f;
    float plan[MAXH * 2];
    if (traj) memcpy(traj, ws, sizeof(float) * C * 4);
    for (int k = 0; k < T; ++k) {
        const int i = t0 + k;
        /* ReplanningCarWorld.step: self.t += 1; if self.t == critical_t: teleport */
        if (d->teleport_step > 0 && (i + 1) == d->teleport_step) {
            const int car = d->teleport_car[sample];
            if (car >= 1 && car < C) memcpy(ws + 4 * car, d->teleport_state, 4 * sizeof(float));
        }
        /* designer reward on past_state with the designer weights (mpc_ord.py:99) */
        const float r = ocd_oracle_reward(d, ws, d->designer_weights, NULL, NULL);
        G = G + r;                               /* sample_reward = 0; sample_reward += ... */
        /* ego plans (world.py:102-104) */
        plan_one(d, ws, w_plan, d->check_plans ? oplans : NULL, plan, NULL, NULL, NULL, NULL, NULL);
        if (ctrl) { ctrl[2 * k] = plan[0]; ctrl[2 * k + 1] = plan[1]; }
        /* all cars step through the real dynamics (world.py:106-107) */
        float nxt[OCD_MAX_CARS * 4];
        {
            float s, c;
            ocd_ref_sincosf(ws[3], &s, &c);
            dyn_fwd(ws[0], ws[1], ws[2], ws[3], c, s, plan[0], plan[1], dt, dt2, d->ego_friction,
                    &nxt[0], &nxt[1], &nxt[2], &nxt[3], NULL);
        }
        for (int j = 0; j < NO; ++j) {
            /* FixedPlanCar: control of real step i is plan[i] if i < len(plan) else default */
            const float *u = (i < d->other_plan_len[j]) ? d->other_plan[j][i] : d->other_default[j];
            const float *o = ws + 4 * (j + 1);
            float s, c;
            ocd_ref_sincosf(o[3], &s, &c);
            dyn_fwd(o[0], o[1], o[2], o[3], c, s, u[0], u[1], dt, dt2, d->other_friction[j],
                    &nxt[4 * (j + 1)], &nxt[4 * (j + 1) + 1], &nxt[4 * (j + 1) + 2], &nxt[4 * (j + 1) + 3], NULL);
        }
        memcpy(ws, nxt, sizeof(float) * C * 4);
        if (traj) memcpy(traj + (size_t)(k + 1) * C * 4, ws, sizeof(float) * C * 4);
    }
    return G;
}

/* one episode: world.reset() (every car back to its init_state, FixedPlanCar.t = 0), then T steps */
static float episode(const ocd_scenario_desc *d, const float *init, const float *w_plan, int sample,
                     float *traj, float *ctrl)
{
    float ws[OCD_MAX_CARS * 4];
    memcpy(ws, init, 4 * sizeof(float));
    for (int j = 0; j < d->n_cars - 1; ++j) memcpy(ws + 4 * (j + 1), d->other_init[j], 4 * sizeof(float));
    return run_steps(d, ws, w_plan, sample, 0, d->episode_len, traj, ctrl);
}

int32_t ocd_rollout_from_state_cpu(const ocd_scenario_desc *d, const float *world_state,
                                   const float *weights, int32_t weights_per_problem,
                                   int32_t first_step, int32_t n_steps, int32_t sample,
                                   float *returns_out, float *traj_out, float *ctrl_out, int64_t B)
{
    if (!check_desc(d) || !world_state || !returns_out || B < 0 || n_steps < 0) return OCD_ERR_INVALID_ARG;
    const int C = d->n_cars, D = OCD_N_FEATURES(d->reward_kind, d->n_lanes);
    for (int64_t b = 0; b < B; ++b) {
        float ws[OCD_MAX_CARS * 4];
        memcpy(ws, world_state + b * C * 4, sizeof(float) * C * 4);
        const float *w = weights ? (weights + (weights_per_problem ? b * D : 0)) : NULL;
        returns_out[b] = run_steps(d, ws, w, sample, first_step, n_steps,
                                   traj_out ? traj_out + b * (n_steps + 1) * C * 4 : NULL,
                                   ctrl_out ? ctrl_out + b * n_steps * 2 : NULL);
    }
    return OCD_OK;
}

int32_t ocd_rollout_episodes_cpu(const ocd_scenario_desc *d, const float *init_states,
                                 const float *cand_weights, int64_t P, int64_t N,
                                 int64_t ep_begin, int64_t ep_end,
                                 float *returns_out, float *traj_out, float *ctrl_out,
                                 int32_t n_threads, int32_t reset_phase)
{
    if (!check_desc(d) || !init_states || !returns_out) return OCD_ERR_INVALID_ARG;
    const int64_t S = d->n_samples, E = P * N * S;
    if (ep_begin < 0 || ep_end < ep_begin || ep_end > E) return OCD_ERR_INVALID_ARG;
    if (d->reward_kind != OCD_REWARD_TARGET_SPEED && !cand_weights) return OCD_ERR_INVALID_ARG;
    const int C = d->n_cars, T = d->episode_len, D = OCD_N_FEATURES(d->reward_kind, d->n_lanes);
    (void)n_threads;
#ifdef _OPENMP
    const int nt = n_threads > 0 ? n_threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
#endif
    for (int64_t e = ep_begin; e < ep_end; ++e) {
        const int64_t s = e % S, n = (e / S) % N, p = e / (S * N);
        const int64_t o = e - ep_begin;
        /* ReplanningCarWorld.reset() toggles the removed car on every reset (replanning_world.py:24-27):
         * episode e of a sequential evaluation is reset number reset_phase + e */
        const int tp = d->teleport_period > 0 ? (int)((reset_phase + e) % d->teleport_period) : (int)s;
        returns_out[o] = episode(d, init_states + 4 * n, cand_weights ? cand_weights + p * D : NULL, tp,
                                 traj_out ? traj_out + o * (T + 1) * C * 4 : NULL,
                                 ctrl_out ? ctrl_out + o * T * 2 : NULL);
    }
    return OCD_OK;
}

int32_t ocd_reward_batch_cpu(const ocd_scenario_desc *d, const float *world_state, const float *weights,
                             float *feats_out, float *reward_out, int64_t B)
{
    if (!check_desc(d) || !world_state) return OCD_ERR_INVALID_ARG;
    const int C = d->n_cars, D = OCD_N_FEATURES(d->reward_kind, d->n_lanes);
    for (int64_t b = 0; b < B; ++b) {
        const float r = ocd_oracle_reward(d, world_state + b * C * 4, weights,
                                          feats_out ? feats_out + b * D : NULL, NULL);
        if (reward_out) reward_out[b] = r;
    }
    return OCD_OK;
}
