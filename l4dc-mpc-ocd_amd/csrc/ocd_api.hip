// ocd_api.hip -- the C ABI of include/ocd.h over the gfx950 kernels.
//
// Host-side only: argument validation, kernel-argument packing, launches on
// the caller's stream, error reporting.  No CPU implementation of the planner
// lives here (or anywhere in the product): without a HIP device every compute
// entry point returns OCD_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/ocd.h"
#include "ocd_kernels.h"

#define OCD_MAX_DEVICES 16
#include <mutex>

#include <vector>

struct ocd_scenario {
    ocd_scenario_desc desc;
    int32_t K;
    int32_t D;
    // per-handle options (ocd_scenario_set_option)
    int32_t opt_segs = 0, opt_no_skips = 0, opt_scan_mode = 0, opt_no_unify = 0, opt_reset_phase = 0, opt_chunk = 0, opt_no_lat = 0, opt_concurrent = 1;
    // The planner's fixed view of the scripted cars' plans (planner_car.py:58-80:
    // plan[j] from index 0, then the assumed default) is a scenario constant; rollouts
    // read it from a small device buffer owned by the handle, one per device.
    std::mutex mu;
    float *dev_plans[OCD_MAX_DEVICES];
    // terminal-value table (ocd_scenario_set_leaf_value): host copy + one device copy per device
    std::vector<float> leaf_host;          // [n0 + n1 + n2 grid values | n0*n1*n2 table values]
    int32_t leaf_n[3] = {0, 0, 0};
    int32_t leaf_proj = 0;
    float *dev_leaf[OCD_MAX_DEVICES];
    int32_t n_cus[OCD_MAX_DEVICES];
    int32_t two_sided = 0;                 // fence_shape * fence_width < 1/80: generic kernels, both fence sides (ocd_kernels.h)
    mutable int32_t last_launch[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ocd_scenario_last_launch (guarded by mu)
    // ocd_rollout_indexed with a DEVICE-memory index: the kernels report an out-of-range row here (pinned host memory the
    // device writes; 1 + the row's position); sticky until ocd_scenario_index_error reads it
    mutable int32_t *index_error = nullptr;
};

#ifdef OCD_STAMPS
// diagnostic build only (make stamps): where the kernels drop their per-wavefront cycle totals
static unsigned long long *g_stamp_buf = nullptr;
extern "C" void ocd_debug_set_stamp_buffer(void *dev_ptr) { g_stamp_buf = (unsigned long long *)dev_ptr; }
#endif

namespace {

thread_local char g_err[512] = "";

int32_t fail(int32_t status, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return status;
}

int32_t hip_fail(hipError_t e, const char *what)
{
    return fail(OCD_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

int32_t validate(const ocd_scenario_desc *d)
{
    if (!d) return fail(OCD_ERR_INVALID_ARG, "descriptor is NULL");
    if (d->abi_version != OCD_ABI_VERSION)
        return fail(OCD_ERR_INVALID_ARG, "descriptor abi_version %d != %d", d->abi_version, OCD_ABI_VERSION);
    if (d->reward_kind != OCD_REWARD_LANE_FEATURES && d->reward_kind != OCD_REWARD_TARGET_SPEED &&
        d->reward_kind != OCD_REWARD_LINEAR_TARGET_SPEED)
        return fail(OCD_ERR_INVALID_ARG, "unknown reward_kind %d", d->reward_kind);
    if (d->n_cars < 1 || d->n_cars > OCD_MAX_CARS) return fail(OCD_ERR_INVALID_ARG, "n_cars %d out of [1,%d]", d->n_cars, OCD_MAX_CARS);
    if (d->n_lanes < 0 || d->n_lanes > OCD_MAX_LANES) return fail(OCD_ERR_INVALID_ARG, "n_lanes %d out of [0,%d]", d->n_lanes, OCD_MAX_LANES);
    if (d->reward_kind == OCD_REWARD_LANE_FEATURES && (d->n_lanes < 1 || d->n_cars < 2))
        return fail(OCD_ERR_INVALID_ARG, "lane-feature reward needs >=1 lane and >=1 scripted car");
    if (d->reward_kind == OCD_REWARD_LANE_FEATURES && !(d->lane_normal_y == 0.0f))
        return fail(OCD_ERR_UNSUPPORTED, "lane_normal_y = %g: lanes run along y (StraightLane.n = (-1, 0), world.py:150-158); "
                                         "the planner's lane offset has no y-term", (double)d->lane_normal_y);
    if (d->horizon < 1 || d->horizon > OCD_MAX_HORIZON) return fail(OCD_ERR_INVALID_ARG, "horizon %d out of [1,%d]", d->horizon, OCD_MAX_HORIZON);
    if (d->n_iter < 0) return fail(OCD_ERR_INVALID_ARG, "n_iter %d < 0", d->n_iter);
    if (d->episode_len < 0) return fail(OCD_ERR_INVALID_ARG, "episode_len %d < 0", d->episode_len);
    if (d->n_samples < 1 || d->n_samples > OCD_MAX_SAMPLES) return fail(OCD_ERR_INVALID_ARG, "n_samples %d out of [1,%d]", d->n_samples, OCD_MAX_SAMPLES);
    for (int j = 0; j < d->n_cars - 1; ++j)
        if (d->other_plan_len[j] < 0 || d->other_plan_len[j] > OCD_MAX_PLAN)
            return fail(OCD_ERR_INVALID_ARG, "other_plan_len[%d] = %d out of [0,%d]", j, d->other_plan_len[j], OCD_MAX_PLAN);
    if (d->teleport_period < 0 || d->teleport_period > OCD_MAX_SAMPLES)
        return fail(OCD_ERR_INVALID_ARG, "teleport_period %d out of [0,%d]", d->teleport_period, OCD_MAX_SAMPLES);
    if (d->teleport_step > 0)
        for (int s = 0; s < (d->teleport_period > 0 ? d->teleport_period : d->n_samples); ++s)
            if (d->teleport_car[s] >= d->n_cars)
                return fail(OCD_ERR_INVALID_ARG, "teleport_car[%d] = %d >= n_cars", s, d->teleport_car[s]);
    if (!(d->dt > 0.0f)) return fail(OCD_ERR_INVALID_ARG, "dt must be > 0");
    // smooth_threshold(x) = F1 / (F1 + F2), F = exp(-1 / (shape * .)): on the road side of the fence F1 = 0 and
    // F2 = exp(-1 / (shape * (width - xd))) with width - xd >= width.  The reference always has shape * width = c = 5
    // (math_utils.py:88-95, merging.py:80).  Below 1/87 F2 flushes to 0 as well and the feature is 0/0 = NaN on a band of the
    // road in the reference itself; the specialised kernels (ocd_device.h: reward_state and the forms derived from it) evaluate
    // ONE side of the fence per lane and take the other as exactly 0, which that NaN would break.  Round 4 refused such a
    // descriptor; since round 5 it is accepted and its handle runs the generic kernels with both sides of the fence
    // evaluated as merging.py:80-81 writes them (ocd_scenario::two_sided, set in ocd_scenario_create).
    if (d->reward_kind == OCD_REWARD_LANE_FEATURES && !(d->fence_lo >= 0.0f && d->fence_width > 0.0f))
        return fail(OCD_ERR_INVALID_ARG, "fence_lo must be >= 0 and fence_width > 0 (0.05*num_lanes - 0.05, 0.05)");
    return OCD_OK;
}

// the kernels' template parameter L (ocd_device.h: feat_dim)
int kernel_L(const ocd_scenario_desc &d)
{
    if (d.reward_kind == OCD_REWARD_LANE_FEATURES) return d.n_lanes;
    return d.reward_kind == OCD_REWARD_LINEAR_TARGET_SPEED ? -1 : 0;
}

int32_t need_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(OCD_ERR_NO_DEVICE, "no HIP device visible (the planner has no CPU fallback)");
    }
    return OCD_OK;
}

// device-side constants of the handle on the current device: compute-unit count, terminal-value table
int32_t device_state(const ocd_scenario *scn_c, hipStream_t st, ocd::KernelParams &p)
{
    ocd_scenario *scn = const_cast<ocd_scenario *>(scn_c);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
    if (dev < 0 || dev >= OCD_MAX_DEVICES) return fail(OCD_ERR_UNSUPPORTED, "device ordinal %d >= %d", dev, OCD_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(scn->mu);
    if (scn->n_cus[dev] == 0) {
        int n = 0;
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess || n <= 0) return hip_fail(e, "hipDeviceGetAttribute(multiprocessor count)");
        scn->n_cus[dev] = n;
    }
    // "concurrent_launches" G: the launch rules plan this launch for 1/G of the compute units, so that G launches on G
    // streams -- each at most one wavefront per SIMD of ITS share -- sit side by side on the chip
    p.n_cus = scn->opt_concurrent > 1 ? (scn->n_cus[dev] / scn->opt_concurrent > 0 ? scn->n_cus[dev] / scn->opt_concurrent : 1)
                                      : scn->n_cus[dev];
    if (!scn->leaf_host.empty()) {
        if (!scn->dev_leaf[dev]) {
            const size_t bytes = scn->leaf_host.size() * sizeof(float);
            float *buf = nullptr;
            e = hipMalloc(&buf, bytes);
            if (e != hipSuccess) return hip_fail(e, "hipMalloc(terminal-value table)");
            e = hipMemcpyAsync(buf, scn->leaf_host.data(), bytes, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { (void)hipFree(buf); return hip_fail(e, "hipMemcpy(terminal-value table)"); }
            scn->dev_leaf[dev] = buf;
        }
        const int ng = scn->leaf_n[0] + scn->leaf_n[1] + scn->leaf_n[2];
        p.leaf.grid = scn->dev_leaf[dev];
        p.leaf.values = scn->dev_leaf[dev] + ng;
        for (int k = 0; k < 3; ++k) p.leaf.n[k] = scn->leaf_n[k];
        p.leaf.proj_kind = scn->leaf_proj;
    }
    return OCD_OK;
}

void base_params(const ocd_scenario *scn, ocd::KernelParams &p)
{
    std::memset(&p, 0, sizeof(p));
    p.d = scn->desc;
    p.K = scn->K;
    p.S = scn->desc.n_samples;
    p.segs_used = scn->opt_segs;
    p.no_skips = scn->opt_no_skips;
    p.scan_mode = scn->opt_scan_mode;
    p.no_unify = scn->opt_no_unify;
    p.force_full = scn->opt_no_skips ? ~0ull : 0ull;
    p.force_full_any = scn->opt_no_unify ? ~0ull : 0ull;
    p.no_latency_build = scn->opt_no_lat;
    p.reset_phase = scn->opt_reset_phase;
    p.chunk_size = scn->opt_chunk;
    if (scn->two_sided) {                  // every feature of every lane, the generic LDS-window kernel (no specialised build)
        p.two_sided = 1;
        p.no_skips = 1;
        p.force_full = ~0ull;
        p.scan_mode = 1;
        p.no_latency_build = 1;
    }
}

int32_t launch(const ocd_scenario *scn, ocd::KernelParams &p, void *hip_stream)
{
    int32_t ds = device_state(scn, (hipStream_t)hip_stream, p);
    if (ds != OCD_OK) return ds;
#ifdef OCD_STAMPS
    p.debug = g_stamp_buf;
#endif
    bool supported = false;
    const int L = kernel_L(scn->desc);
    int32_t info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    p.launch_info = info;
    hipError_t e = ocd::launch_mpc_dispatch(scn->desc.horizon, scn->desc.n_cars - 1, L, p, (hipStream_t)hip_stream, &supported);
    {
        std::lock_guard<std::mutex> lock(const_cast<ocd_scenario *>(scn)->mu);
        std::memcpy(scn->last_launch, info, sizeof(info));
    }
    if (!supported)
        return fail(OCD_ERR_UNSUPPORTED, "no compiled kernel for horizon %d with %d scripted cars and %d lanes (see OCD_PAIR_TABLE)",
                    scn->desc.horizon, scn->desc.n_cars - 1, L);
    if (e != hipSuccess) return hip_fail(e, "mpc_kernel launch");
    return OCD_OK;
}

} // namespace

extern "C" {

int32_t ocd_abi_version(void) { return OCD_ABI_VERSION; }

int32_t ocd_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(OCD_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return n;
}

const char *ocd_last_error(void) { return g_err; }

int32_t ocd_scenario_set_option(ocd_scenario *scn, const char *name, int32_t value)
{
    if (!scn) return fail(OCD_ERR_INVALID_ARG, "scenario is NULL");
    if (!name) return fail(OCD_ERR_INVALID_ARG, "option name is NULL");
    if (std::strcmp(name, "segs_per_wave") == 0) {
        if (value < 0 || value > 64) return fail(OCD_ERR_INVALID_ARG, "segs_per_wave %d out of [0,64]", value);
        scn->opt_segs = value;
        return OCD_OK;
    }
    if (std::strcmp(name, "scan_mode") == 0) {
        if (value < 0 || value > 4) return fail(OCD_ERR_INVALID_ARG, "scan_mode %d out of [0,4]", value);
        scn->opt_scan_mode = value;
        return OCD_OK;
    }
    if (std::strcmp(name, "chunk_size") == 0) {
        if (value < 0 || value > OCD_MAX_HORIZON) return fail(OCD_ERR_INVALID_ARG, "chunk_size %d out of [0,%d]", value, OCD_MAX_HORIZON);
        scn->opt_chunk = value;
        return OCD_OK;
    }
    if (std::strcmp(name, "concurrent_launches") == 0) {
        if (value < 0 || value > 16) return fail(OCD_ERR_INVALID_ARG, "concurrent_launches %d out of [0,16]", value);
        scn->opt_concurrent = value > 1 ? value : 1;
        return OCD_OK;
    }
    if (std::strcmp(name, "no_unified_features") == 0) { scn->opt_no_unify = value ? 1 : 0; return OCD_OK; }
    if (std::strcmp(name, "no_latency_build") == 0) { scn->opt_no_lat = value ? 1 : 0; return OCD_OK; }
    if (std::strcmp(name, "no_feature_skips") == 0) { scn->opt_no_skips = value ? 1 : 0; return OCD_OK; }
    if (std::strcmp(name, "reset_phase") == 0) {
        if (value < 0) return fail(OCD_ERR_INVALID_ARG, "reset_phase %d < 0", value);
        scn->opt_reset_phase = value;
        return OCD_OK;
    }
    return fail(OCD_ERR_INVALID_ARG, "unknown option '%s'", name);
}

int32_t ocd_scenario_last_launch(const ocd_scenario *scn, int32_t info[8])
{
    if (!scn || !info) return fail(OCD_ERR_INVALID_ARG, "scenario or info is NULL");
    std::lock_guard<std::mutex> lock(const_cast<ocd_scenario *>(scn)->mu);
    std::memcpy(info, scn->last_launch, sizeof(scn->last_launch));
    return OCD_OK;
}

int32_t ocd_scenario_plan_launch(const ocd_scenario *scn, int64_t n_problems, int32_t n_cus, int32_t info[8])
{
    if (!scn || !info) return fail(OCD_ERR_INVALID_ARG, "scenario or info is NULL");
    if (n_problems < 1) return fail(OCD_ERR_INVALID_ARG, "n_problems %lld < 1", (long long)n_problems);
    if (n_cus < 0) return fail(OCD_ERR_INVALID_ARG, "n_cus %d < 0", n_cus);
    ocd::KernelParams p;
    base_params(scn, p);
    p.mode = ocd::OCD_MODE_PLAN;
    p.n_problems = n_problems;
    p.n_cus = n_cus > 0 ? n_cus : 256;
    p.dry_run = 1;
    if (!scn->leaf_host.empty()) {                       // (never dereferenced in a dry run)
        const int ng = scn->leaf_n[0] + scn->leaf_n[1] + scn->leaf_n[2];
        p.leaf.grid = scn->leaf_host.data();
        p.leaf.values = scn->leaf_host.data() + ng;
        for (int k = 0; k < 3; ++k) p.leaf.n[k] = scn->leaf_n[k];
        p.leaf.proj_kind = scn->leaf_proj;
    }
    int32_t rec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    p.launch_info = rec;
    bool supported = false;
    (void)ocd::launch_mpc_dispatch(scn->desc.horizon, scn->desc.n_cars - 1, kernel_L(scn->desc), p, nullptr, &supported);
    if (!supported)
        return fail(OCD_ERR_UNSUPPORTED, "no compiled kernel for horizon %d with %d scripted cars (see OCD_PAIR_TABLE)",
                    scn->desc.horizon, scn->desc.n_cars - 1);
    std::memcpy(info, rec, sizeof(rec));
    return OCD_OK;
}

int32_t ocd_scenario_set_leaf_value(ocd_scenario *scn, const float *grid0, int32_t n0,
                                    const float *grid1, int32_t n1, const float *grid2, int32_t n2,
                                    const float *values, int32_t proj_kind)
{
    if (!scn) return fail(OCD_ERR_INVALID_ARG, "scenario is NULL");
    std::lock_guard<std::mutex> lock(scn->mu);
    for (int i = 0; i < OCD_MAX_DEVICES; ++i)
        if (scn->dev_leaf[i]) { (void)hipFree(scn->dev_leaf[i]); scn->dev_leaf[i] = nullptr; }
    scn->leaf_host.clear();
    if (!values) return OCD_OK;
    if (!grid0 || !grid1 || !grid2) return fail(OCD_ERR_INVALID_ARG, "a grid pointer is NULL");
    if (proj_kind != 0 && proj_kind != 1) return fail(OCD_ERR_INVALID_ARG, "proj_kind %d not in {0,1}", proj_kind);
    const int32_t n[3] = {n0, n1, n2};
    const float *gr[3] = {grid0, grid1, grid2};
    for (int k = 0; k < 3; ++k) {
        if (n[k] < 2 || n[k] > 4096) return fail(OCD_ERR_INVALID_ARG, "grid %d has %d points (need 2..4096)", k, n[k]);
        for (int i = 1; i < n[k]; ++i)
            if (!(gr[k][i] > gr[k][i - 1])) return fail(OCD_ERR_INVALID_ARG, "grid %d is not strictly ascending at %d", k, i);
    }
    const size_t nv = (size_t)n0 * n1 * n2;
    scn->leaf_host.reserve((size_t)n0 + n1 + n2 + nv);
    for (int k = 0; k < 3; ++k) scn->leaf_host.insert(scn->leaf_host.end(), gr[k], gr[k] + n[k]);
    scn->leaf_host.insert(scn->leaf_host.end(), values, values + nv);
    for (int k = 0; k < 3; ++k) scn->leaf_n[k] = n[k];
    scn->leaf_proj = proj_kind;
    return OCD_OK;
}

int32_t ocd_scenario_create(const ocd_scenario_desc *desc, ocd_scenario **out)
{
    if (!out) return fail(OCD_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    int32_t st = validate(desc);
    if (st != OCD_OK) return st;
    ocd_scenario *s = new (std::nothrow) ocd_scenario;
    if (!s) return fail(OCD_ERR_INVALID_ARG, "out of memory");
    s->desc = *desc;
    s->K = desc->extra_inits ? 6 : 3;
    s->D = OCD_N_FEATURES(desc->reward_kind, desc->n_lanes);
    s->two_sided = (desc->reward_kind == OCD_REWARD_LANE_FEATURES && !(desc->fence_shape * desc->fence_width >= 0.0125f)) ? 1 : 0;
    for (int i = 0; i < OCD_MAX_DEVICES; ++i) { s->dev_plans[i] = nullptr; s->dev_leaf[i] = nullptr; s->n_cus[i] = 0; }
    *out = s;
    return OCD_OK;
}

void ocd_scenario_destroy(ocd_scenario *scn)
{
    if (!scn) return;
    for (int i = 0; i < OCD_MAX_DEVICES; ++i)
        if (scn->dev_plans[i]) (void)hipFree(scn->dev_plans[i]);
    for (int i = 0; i < OCD_MAX_DEVICES; ++i)
        if (scn->dev_leaf[i]) (void)hipFree(scn->dev_leaf[i]);
    if (scn->index_error) (void)hipHostFree(scn->index_error);
    delete scn;
}

int32_t ocd_plan_batch_from(const ocd_scenario *scn, const float *world_state, const float *init_speed,
                            const float *weights, int32_t weights_per_problem, const float *other_plans,
                            float *plans_out, float *best_loss_out, int32_t *best_init_out,
                            float *all_plans_out, float *all_losses_out, int64_t B, void *hip_stream)
{
    if (!scn) return fail(OCD_ERR_INVALID_ARG, "scenario is NULL");
    if (B < 0) return fail(OCD_ERR_INVALID_ARG, "B = %lld < 0", (long long)B);
    if (B == 0) return OCD_OK;
    if (!world_state || !plans_out) return fail(OCD_ERR_INVALID_ARG, "world_state / plans_out is NULL");
    if (scn->desc.reward_kind != OCD_REWARD_TARGET_SPEED && !weights)
        return fail(OCD_ERR_INVALID_ARG, "weights is NULL for a weighted-feature reward");
    int32_t st = need_device();
    if (st != OCD_OK) return st;
    ocd::KernelParams p;
    base_params(scn, p);
    p.mode = ocd::OCD_MODE_PLAN;
    p.T = 1;
    p.ego_states = world_state;
    p.init_speed = init_speed;
    p.weights = weights;
    p.weights_per_problem = weights_per_problem;
    p.other_plans = other_plans;
    p.plans_out = plans_out;
    p.best_loss_out = best_loss_out;
    p.best_init_out = best_init_out;
    p.all_plans_out = all_plans_out;
    p.all_losses_out = all_losses_out;
    p.n_problems = B;
    return launch(scn, p, hip_stream);
}

int32_t ocd_plan_batch(const ocd_scenario *scn, const float *world_state,
                       const float *weights, int32_t weights_per_problem, const float *other_plans,
                       float *plans_out, float *best_loss_out, int32_t *best_init_out,
                       float *all_plans_out, float *all_losses_out, int64_t B, void *hip_stream)
{
    return ocd_plan_batch_from(scn, world_state, nullptr, weights, weights_per_problem, other_plans, plans_out,
                               best_loss_out, best_init_out, all_plans_out, all_losses_out, B, hip_stream);
}

static int32_t rollout_params(const ocd_scenario *scn, const float *init_states, const float *cand_weights,
                              int64_t P, int64_t N, int64_t ep_begin, int64_t ep_end,
                              float *returns_out, float *traj_out, float *ctrl_out,
                              const float *other_plans_dev, ocd::KernelParams &p)
{
    if (!scn) return fail(OCD_ERR_INVALID_ARG, "scenario is NULL");
    if (P < 1 || N < 1) return fail(OCD_ERR_INVALID_ARG, "P = %lld, N = %lld must be >= 1", (long long)P, (long long)N);
    const int64_t E = P * N * scn->desc.n_samples;
    if (ep_begin < 0 || ep_end < ep_begin || ep_end > E)
        return fail(OCD_ERR_INVALID_ARG, "episode range [%lld, %lld) outside [0, %lld)", (long long)ep_begin, (long long)ep_end, (long long)E);
    if (!init_states || !returns_out) return fail(OCD_ERR_INVALID_ARG, "init_states / returns_out is NULL");
    if (scn->desc.reward_kind != OCD_REWARD_TARGET_SPEED && !cand_weights)
        return fail(OCD_ERR_INVALID_ARG, "cand_weights is NULL for a weighted-feature reward");
    base_params(scn, p);
    p.mode = ocd::OCD_MODE_ROLLOUT;
    p.ego_states = init_states;
    p.weights = cand_weights;
    p.other_plans = other_plans_dev;
    p.returns_out = returns_out;
    p.traj_out = traj_out;
    p.ctrl_out = ctrl_out;
    p.n_problems = ep_end - ep_begin;
    p.ep_begin = ep_begin;
    p.N = N;
    p.T = scn->desc.episode_len;
    return OCD_OK;
}

} // extern "C"

namespace {

int32_t scripted_plans_device(const ocd_scenario *scn_c, hipStream_t st, const float **out)
{
    *out = nullptr;
    ocd_scenario *scn = const_cast<ocd_scenario *>(scn_c);
    const ocd_scenario_desc &d = scn->desc;
    if (!d.check_plans || d.n_cars < 2) return OCD_OK;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
    if (dev < 0 || dev >= OCD_MAX_DEVICES) return fail(OCD_ERR_UNSUPPORTED, "device ordinal %d >= %d", dev, OCD_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(scn->mu);
    if (scn->dev_plans[dev]) { *out = scn->dev_plans[dev]; return OCD_OK; }
    const int H = d.horizon, NO = d.n_cars - 1;
    float host[OCD_MAX_OTHERS * OCD_MAX_HORIZON * 2];
    for (int j = 0; j < NO; ++j)
        for (int t = 0; t < H; ++t) {
            const float *src = (t < d.other_plan_len[j]) ? d.other_plan[j][t] : d.other_assumed_default[j];
            host[(j * H + t) * 2] = src[0];
            host[(j * H + t) * 2 + 1] = src[1];
        }
    const size_t bytes = sizeof(float) * NO * H * 2;
    float *buf = nullptr;
    e = hipMalloc(&buf, bytes);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(scripted plans)");
    e = hipMemcpyAsync(buf, host, bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);   // host[] is a stack buffer
    if (e != hipSuccess) { (void)hipFree(buf); return hip_fail(e, "hipMemcpy(scripted plans)"); }
    scn->dev_plans[dev] = buf;
    *out = buf;
    return OCD_OK;
}

} // namespace

extern "C" {

int32_t ocd_rollout_episodes(const ocd_scenario *scn, const float *init_states, const float *cand_weights,
                             int64_t P, int64_t N, int64_t ep_begin, int64_t ep_end,
                             float *returns_out, float *traj_out, float *ctrl_out, void *hip_stream)
{
    if (!scn) return fail(OCD_ERR_INVALID_ARG, "scenario is NULL");
    if (ep_begin == ep_end && ep_begin >= 0) return OCD_OK;
    ocd::KernelParams p;
    int32_t st = rollout_params(scn, init_states, cand_weights, P, N, ep_begin, ep_end, returns_out, traj_out, ctrl_out, nullptr, p);
    if (st != OCD_OK) return st;                          // argument errors first, device errors after
    st = need_device();
    if (st != OCD_OK) return st;
    st = scripted_plans_device(scn, (hipStream_t)hip_stream, &p.other_plans);
    if (st != OCD_OK) return st;
    return launch(scn, p, hip_stream);
}

int32_t ocd_rollout_indexed(const ocd_scenario *scn, const float *init_states, int64_t N_rows,
                            const float *cand_weights, int64_t P_rows, const int32_t *episode_index, int64_t E,
                            float *returns_out, float *traj_out, float *ctrl_out, void *hip_stream)
{
    if (!scn) return fail(OCD_ERR_INVALID_ARG, "scenario is NULL");
    if (E < 0) return fail(OCD_ERR_INVALID_ARG, "E = %lld < 0", (long long)E);
    if (E == 0) return OCD_OK;
    if (!episode_index) return fail(OCD_ERR_INVALID_ARG, "episode_index is NULL");
    if (P_rows > INT32_MAX || N_rows > INT32_MAX)
        return fail(OCD_ERR_INVALID_ARG, "P_rows / N_rows beyond what an int32 index row can name");
    ocd::KernelParams p;
    // (the flat-range checks of rollout_params on a population of P_rows x N_rows; the launch then covers E index rows)
    int32_t st = rollout_params(scn, init_states, cand_weights, P_rows, N_rows, 0, 0, returns_out, traj_out, ctrl_out, nullptr, p);
    if (st != OCD_OK) return st;
    p.n_problems = E;
    p.ep_index = episode_index;
    p.P_rows = P_rows;
    st = need_device();
    if (st != OCD_OK) return st;
    // An index row outside the tables is a caller bug, never a clamp (round 5 clamped: a wrong-but-plausible fitness).
    // Host memory (pinned / registered: what the native lockstep loop builds its index in) is checked here, before anything
    // is launched; device memory is checked by the kernels, which report through the handle's pinned error word.
    hipPointerAttribute_t attr;
    const hipError_t pe = hipPointerGetAttributes(&attr, episode_index);
    if (pe != hipSuccess) (void)hipGetLastError();               // (plain host memory: not device-addressable, caught below)
    const bool host_index = pe == hipSuccess && attr.type == hipMemoryTypeHost;
    if (pe != hipSuccess || attr.type == hipMemoryTypeUnregistered)
        return fail(OCD_ERR_INVALID_ARG, "episode_index is not device-addressable memory (device, or pinned / registered host)");
    if (host_index) {
        for (int64_t e = 0; e < E; ++e) {
            const int32_t *ix = episode_index + 3 * e;
            if (ix[0] < 0 || ix[0] >= P_rows || ix[1] < 0 || ix[1] >= N_rows || ix[2] < 0)
                return fail(OCD_ERR_INVALID_ARG, "episode_index row %lld = (%d, %d, %d): candidate row outside [0, %lld), init row "
                            "outside [0, %lld) or negative reset number", (long long)e, ix[0], ix[1], ix[2], (long long)P_rows, (long long)N_rows);
        }
    } else {
        std::lock_guard<std::mutex> lock(const_cast<ocd_scenario *>(scn)->mu);
        if (!scn->index_error) {
            int32_t *w = nullptr;
            const hipError_t he = hipHostMalloc((void **)&w, 2 * sizeof(int32_t), hipHostMallocMapped);
            if (he != hipSuccess) return hip_fail(he, "hipHostMalloc(index error word)");
            w[0] = 0; w[1] = 0;
            scn->index_error = w;
        }
        if (scn->index_error[0] != 0) {
            const int32_t row = scn->index_error[0] - 1;
            scn->index_error[0] = 0;
            return fail(OCD_ERR_INVALID_ARG, "an earlier ocd_rollout_indexed launch on this handle found episode_index row %d out of "
                                             "range (its return is NaN); nothing was launched now", row);
        }
        p.index_error = scn->index_error;
    }
    st = scripted_plans_device(scn, (hipStream_t)hip_stream, &p.other_plans);
    if (st != OCD_OK) return st;
    return launch(scn, p, hip_stream);
}

int32_t ocd_scenario_index_error(const ocd_scenario *scn, int64_t *bad_row)
{
    if (!scn) return fail(OCD_ERR_INVALID_ARG, "scenario is NULL");
    std::lock_guard<std::mutex> lock(const_cast<ocd_scenario *>(scn)->mu);
    if (bad_row) *bad_row = -1;
    if (!scn->index_error || scn->index_error[0] == 0) return OCD_OK;
    const int32_t row = scn->index_error[0] - 1;
    scn->index_error[0] = 0;
    if (bad_row) *bad_row = row;
    return fail(OCD_ERR_INVALID_ARG, "ocd_rollout_indexed: episode_index row %d was out of range (that episode's return is NaN)", row);
}

int32_t ocd_rollout_from_state(const ocd_scenario *scn, const float *world_state,
                               const float *weights, int32_t weights_per_problem,
                               int32_t first_step, int32_t n_steps, int32_t sample,
                               float *returns_out, float *traj_out, float *ctrl_out,
                               int64_t B, void *hip_stream)
{
    if (!scn) return fail(OCD_ERR_INVALID_ARG, "scenario is NULL");
    if (B < 0 || n_steps < 0 || first_step < 0) return fail(OCD_ERR_INVALID_ARG, "negative B / n_steps / first_step");
    if (sample < 0 || sample >= OCD_MAX_SAMPLES) return fail(OCD_ERR_INVALID_ARG, "sample %d out of [0,%d)", sample, OCD_MAX_SAMPLES);
    if (B == 0) return OCD_OK;
    if (!world_state || !returns_out) return fail(OCD_ERR_INVALID_ARG, "world_state / returns_out is NULL");
    if (scn->desc.reward_kind != OCD_REWARD_TARGET_SPEED && !weights)
        return fail(OCD_ERR_INVALID_ARG, "weights is NULL for a weighted-feature reward");
    int32_t st = need_device();
    if (st != OCD_OK) return st;
    const float *plans = nullptr;
    st = scripted_plans_device(scn, (hipStream_t)hip_stream, &plans);
    if (st != OCD_OK) return st;
    ocd::KernelParams p;
    base_params(scn, p);
    p.mode = ocd::OCD_MODE_ROLLOUT;
    p.from_state = 1;
    p.ego_states = world_state;
    p.weights = weights;
    p.weights_per_problem = weights_per_problem;
    p.other_plans = plans;
    p.returns_out = returns_out;
    p.traj_out = traj_out;
    p.ctrl_out = ctrl_out;
    p.n_problems = B;
    p.N = 1;
    p.T = n_steps;
    p.t0 = first_step;
    p.sample_fixed = sample;
    return launch(scn, p, hip_stream);
}

int32_t ocd_mpc_reward_batch(const ocd_scenario *scn, const float *world_state,
                             const float *weights, int32_t weights_per_problem,
                             const float *controls, const float *other_plans,
                             float *reward_out, float *grad_out, float *traj_out,
                             int64_t B, void *hip_stream)
{
    if (!scn) return fail(OCD_ERR_INVALID_ARG, "scenario is NULL");
    if (B < 0) return fail(OCD_ERR_INVALID_ARG, "B = %lld < 0", (long long)B);
    if (B == 0) return OCD_OK;
    if (!world_state || !controls) return fail(OCD_ERR_INVALID_ARG, "world_state / controls is NULL");
    if (scn->desc.reward_kind != OCD_REWARD_TARGET_SPEED && !weights)
        return fail(OCD_ERR_INVALID_ARG, "weights is NULL for a weighted-feature reward");
    int32_t st = need_device();
    if (st != OCD_OK) return st;
    ocd::KernelParams p;
    base_params(scn, p);
    st = device_state(scn, (hipStream_t)hip_stream, p);
    if (st != OCD_OK) return st;
    p.ego_states = world_state;
    p.weights = weights;
    p.weights_per_problem = weights_per_problem;
    p.other_plans = other_plans;
    p.n_problems = B;
    bool supported = false;
    const int L = kernel_L(scn->desc);
    hipError_t e = ocd::launch_objective(scn->desc.n_cars - 1, L, p, controls, reward_out, grad_out, traj_out,
                                         (hipStream_t)hip_stream, &supported);
    if (!supported) return fail(OCD_ERR_UNSUPPORTED, "objective kernel: %d scripted cars, %d lanes", scn->desc.n_cars - 1, L);
    if (e != hipSuccess) return hip_fail(e, "objective_kernel launch");
    return OCD_OK;
}

int32_t ocd_dynamics_batch(const float *states, const float *controls, float dt, float dt_sq, float friction,
                           float *next_out, int64_t B, void *hip_stream)
{
    if (B < 0) return fail(OCD_ERR_INVALID_ARG, "B < 0");
    if (B == 0) return OCD_OK;
    if (!states || !controls || !next_out) return fail(OCD_ERR_INVALID_ARG, "NULL pointer");
    int32_t st = need_device();
    if (st != OCD_OK) return st;
    hipError_t e = ocd::launch_dynamics(states, controls, dt, dt_sq, friction, next_out, B, (hipStream_t)hip_stream);
    if (e != hipSuccess) return hip_fail(e, "dynamics_kernel launch");
    return OCD_OK;
}

int32_t ocd_reward_batch(const ocd_scenario *scn, const float *world_state, const float *weights,
                         float *feats_out, float *reward_out, int64_t B, void *hip_stream)
{
    if (!scn) return fail(OCD_ERR_INVALID_ARG, "scenario is NULL");
    if (B < 0) return fail(OCD_ERR_INVALID_ARG, "B < 0");
    if (B == 0) return OCD_OK;
    if (!world_state) return fail(OCD_ERR_INVALID_ARG, "world_state is NULL");
    if (scn->desc.reward_kind != OCD_REWARD_TARGET_SPEED && !weights)
        return fail(OCD_ERR_INVALID_ARG, "weights is NULL");
    int32_t st = need_device();
    if (st != OCD_OK) return st;
    ocd::KernelParams p;
    base_params(scn, p);
    p.ego_states = world_state;
    p.weights = weights;
    p.n_problems = B;
    bool supported = false;
    const int L = kernel_L(scn->desc);
    hipError_t e = ocd::launch_reward(scn->desc.n_cars - 1, L, p, feats_out, reward_out, (hipStream_t)hip_stream, &supported);
    if (!supported) return fail(OCD_ERR_UNSUPPORTED, "reward kernel: %d scripted cars, %d lanes", scn->desc.n_cars - 1, L);
    if (e != hipSuccess) return hip_fail(e, "reward_kernel launch");
    return OCD_OK;
}

int32_t ocd_debug_math(const float *in, float *exp_out, float *sin_out, float *cos_out, int64_t n, void *hip_stream)
{
    if (n < 0 || (n > 0 && !in)) return fail(OCD_ERR_INVALID_ARG, "bad arguments");
    if (n == 0) return OCD_OK;
    int32_t st = need_device();
    if (st != OCD_OK) return st;
    hipError_t e = ocd::launch_math(in, exp_out, sin_out, cos_out, n, (hipStream_t)hip_stream);
    if (e != hipSuccess) return hip_fail(e, "math_kernel launch");
    return OCD_OK;
}

int32_t ocd_stream_synchronize(void *hip_stream)
{
    int32_t st = need_device();
    if (st != OCD_OK) return st;
    hipError_t e = hipStreamSynchronize((hipStream_t)hip_stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    return OCD_OK;
}

int32_t ocd_debug_packed_math(const float *num, const float *den, const float *x, float *div_scalar_out,
                              float *div_packed_out, float *exp_scalar_out, float *exp_packed_out, int64_t n_pairs,
                              void *hip_stream)
{
    if (n_pairs < 0 || (n_pairs > 0 && !((num && den) || x)) || ((num == nullptr) != (den == nullptr)))
        return fail(OCD_ERR_INVALID_ARG, "bad arguments");
    if (n_pairs == 0) return OCD_OK;
    int32_t st = need_device();
    if (st != OCD_OK) return st;
    hipError_t e = ocd::launch_packed_math(num, den, x, div_scalar_out, div_packed_out, exp_scalar_out, exp_packed_out,
                                           n_pairs, (hipStream_t)hip_stream);
    if (e != hipSuccess) return hip_fail(e, "packed_math_kernel launch");
    return OCD_OK;
}

int32_t ocd_debug_guarded_division(const float *u, const float *n, const float *w, float *m_out, float *k_out,
                                   float *q_out, int64_t n_pairs, void *hip_stream)
{
    if (n_pairs < 0 || (n_pairs > 0 && !(u || (n && w))) || ((n == nullptr) != (w == nullptr)))
        return fail(OCD_ERR_INVALID_ARG, "bad arguments");
    if (n_pairs == 0) return OCD_OK;
    int32_t st = need_device();
    if (st != OCD_OK) return st;
    hipError_t e = ocd::launch_guarded_division(u, n, w, m_out, k_out, q_out, n_pairs, (hipStream_t)hip_stream);
    if (e != hipSuccess) return hip_fail(e, "guarded_division_kernel launch");
    return OCD_OK;
}

int32_t ocd_debug_feature_variants(const ocd_scenario *scn, const float *world_state, const float *weights, float *out,
                                   int32_t *valid, int64_t B, void *hip_stream)
{
    if (!scn) return fail(OCD_ERR_INVALID_ARG, "scenario is NULL");
    if (B < 0) return fail(OCD_ERR_INVALID_ARG, "B = %lld < 0", (long long)B);
    if (B == 0) return OCD_OK;
    if (!world_state || !weights || !out || !valid) return fail(OCD_ERR_INVALID_ARG, "world_state / weights / out / valid is NULL");
    if (scn->desc.reward_kind != OCD_REWARD_LANE_FEATURES) return fail(OCD_ERR_UNSUPPORTED, "lane-feature rewards only");
    int32_t st = need_device();
    if (st != OCD_OK) return st;
    ocd::KernelParams p;
    base_params(scn, p);
    p.ego_states = world_state;
    p.weights = weights;
    p.n_problems = B;
    bool supported = false;
    hipError_t e = ocd::launch_feature_variants(scn->desc.n_cars - 1, scn->desc.n_lanes, p, out, valid, (hipStream_t)hip_stream, &supported);
    if (!supported) return fail(OCD_ERR_UNSUPPORTED, "feature variants: %d scripted cars, %d lanes (compiled: (1,2) (1,3) (2,2) (2,3) (3,3))",
                                scn->desc.n_cars - 1, scn->desc.n_lanes);
    if (e != hipSuccess) return hip_fail(e, "feature_variants_kernel launch");
    return OCD_OK;
}

int32_t ocd_time_rollout(const ocd_scenario *scn, const float *init_states, const float *cand_weights,
                         int64_t P, int64_t N, int64_t ep_begin, int64_t ep_end,
                         float *returns_out, int32_t reps, float *ms_out, void *hip_stream)
{
    if (!ms_out || reps < 1) return fail(OCD_ERR_INVALID_ARG, "ms_out is NULL or reps < 1");
    if (!scn) return fail(OCD_ERR_INVALID_ARG, "scenario is NULL");
    hipStream_t stream = (hipStream_t)hip_stream;
    ocd::KernelParams p;
    int32_t st = rollout_params(scn, init_states, cand_weights, P, N, ep_begin, ep_end, returns_out, nullptr, nullptr, nullptr, p);
    if (st != OCD_OK) return st;
    st = need_device();
    if (st != OCD_OK) return st;
    st = scripted_plans_device(scn, stream, &p.other_plans);
    if (st != OCD_OK) return st;
    hipEvent_t e0, e1;
    hipError_t e = hipEventCreate(&e0);
    if (e != hipSuccess) return hip_fail(e, "hipEventCreate");
    e = hipEventCreate(&e1);
    if (e != hipSuccess) { (void)hipEventDestroy(e0); return hip_fail(e, "hipEventCreate"); }
    (void)hipEventRecord(e0, stream);
    for (int i = 0; i < reps && st == OCD_OK; ++i) st = launch(scn, p, hip_stream);
    (void)hipEventRecord(e1, stream);
    e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (st != OCD_OK) return st;
    if (e != hipSuccess) return hip_fail(e, "event timing");
    *ms_out = ms / (float)reps;
    return OCD_OK;
}

} // extern "C"
