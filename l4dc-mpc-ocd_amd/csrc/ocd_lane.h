// ocd_lane.h -- cross-lane primitives and diagnostics shared by the planner kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace ocd {

__device__ __forceinline__ bool finite_(float v)
{
    return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u;
}

// DPP moves.  row_shr:1 / row_shl:1 stay inside a 16-lane row (the first / last lane of the row has no
// source); wave_shr:1 / wave_shl:1 shift across the whole wavefront.  BC = bound_ctrl: a lane without a
// source reads 0; otherwise it keeps `old`.
constexpr int DPP_ROW_SHL1 = 0x101, DPP_ROW_SHR1 = 0x111, DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138;

template <int CTRL, bool BC>
__device__ __forceinline__ float dpp_move(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, 0xf, BC));
}

// V_ROW: lane t of a row receives lane t-1 / t+1; the boundary lane keeps `old`
__device__ __forceinline__ float row_below(float old, float src) { return dpp_move<DPP_ROW_SHR1, false>(old, src); }
__device__ __forceinline__ float row_above(float old, float src) { return dpp_move<DPP_ROW_SHL1, false>(old, src); }
// V_SEG: lane l receives lane l-1 / l+1 of the wavefront; the caller selects at segment boundaries
__device__ __forceinline__ float wave_below(float src) { return dpp_move<DPP_WAVE_SHR1, true>(0.0f, src); }
__device__ __forceinline__ float wave_above(float src) { return dpp_move<DPP_WAVE_SHL1, true>(0.0f, src); }

__device__ __forceinline__ float lane_read(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

template <bool B> using bool_c = std::integral_constant<bool, B>;

// In-kernel cycle stamps (diagnostic build only: make STAMPS=1; never shipped).  Section totals of
// wavefront (block, wave) go to p.debug[(block * K + wave) * 16 + section]; no output depends on them.
#ifdef OCD_STAMPS
#define OCD_STAMP_DECL unsigned long long st_acc[16] = {0}, st_last = __builtin_amdgcn_s_memtime();
#define OCD_STAMP_NOW(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long st_now = __builtin_amdgcn_s_memtime(); \
                              __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_sched_barrier(0); st_acc[i] += st_now - st_last; st_last = st_now; } while (0)
#ifdef OCD_STAMPS_LIGHT
// (make stamps_light: only the wavefront's total and its placement -- two s_memtime per wavefront, the product's speed)
#define OCD_STAMP(i) do { } while (0)
#define OCD_STAMP_COUNT(i) do { } while (0)
#else
#define OCD_STAMP(i) OCD_STAMP_NOW(i)
#define OCD_STAMP_COUNT(i) do { st_acc[i] += 1; } while (0)
#endif
#define OCD_STAMP_LAST OCD_STAMP_NOW(0)
#else
#define OCD_STAMP_DECL
#define OCD_STAMP(i) do { } while (0)
#define OCD_STAMP_COUNT(i) do { } while (0)
#define OCD_STAMP_LAST do { } while (0)
#endif

} // namespace ocd
