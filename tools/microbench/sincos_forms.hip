// Lone-wavefront cost (one wavefront per SIMD, whole chip) of the sincos forms of ocd_devmath.h and of the single
// instructions the round-4 form introduced.  cycles per call / per instruction from s_memtime around a dependent loop.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -I../../l4dc-mpc-ocd_amd/csrc sincos_forms.hip -o sincos_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "ocd_devmath.h"

#define REP4(S) S S S S
#define REP16(S) REP4(REP4(S))
#define REP64(S) REP4(REP16(S))

template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, long long *cyc, int iters, float seed)
{
    float a = seed + threadIdx.x * 1e-3f, b = 0.5f;
    int q = threadIdx.x, m = 1;
    float tmp = 0.0f, tmp2 = 0.0f, c = 0.25f;
    const ocd::ScConsts scc = ocd::sc_consts();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { float s_, c_; ocd::sincos_(a, s_, c_); a = s_ + c_; }
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { float s_, c_; ocd::sincos_pk(a, s_, c_, scc); a = s_ + c_; }
        } else if constexpr (MODE == 2) {
            asm volatile(REP64("v_bfe_i32 %0, %0, 0, 1\n") : "+v"(q));
        } else if constexpr (MODE == 3) {
            asm volatile(REP64("v_bfi_b32 %0, %1, %0, %2\n") : "+v"(a) : "v"(m), "v"(b));
        } else if constexpr (MODE == 4) {
            asm volatile(REP64("v_bitop3_b32 %0, %1, %0, %2 bitop3:0x6c\n") : "+v"(a) : "v"(q), "s"(0x80000000u));
        } else if constexpr (MODE == 5) {
            asm volatile(REP64("v_rndne_f32 %0, %0\n") : "+v"(a));
        } else if constexpr (MODE == 6) {
            ocd::v2f p = {a, b}, z = {b, a};
            asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]\n") : "+v"(p) : "v"(z), "v"(scc.k0));
            a = p.x + p.y;
        } else if constexpr (MODE == 7) {
            asm volatile(REP64("v_mul_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        } else if constexpr (MODE == 8) {       // the fused tail's pattern (DPP reads >= 2 instructions after their source's write)
            asm volatile("s_mov_b64 vcc, 0x401\n"
                         REP4(REP4("v_cndmask_b32_dpp %[T], %[A], %[s], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                   "v_mul_f32 %[B], %[T], %[s]\n"
                                   "v_cndmask_b32_dpp %[U], %[C], %[s], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                   "v_add_f32 %[A], %[B], %[s]\n"
                                   "v_mul_f32 %[C], %[U], %[s]\n"
                                   "v_add_f32 %[B], %[C], %[s]\n"))
                         : [A] "+v"(a), [B] "+v"(b), [C] "+v"(c), [T] "+v"(tmp), [U] "+v"(tmp2) : [s] "v"(seed) : "vcc");
        } else if constexpr (MODE == 9) {       // the compiler's form of the same work: mov_dpp + cndmask on an SGPR pair
            asm volatile("s_mov_b64 s[20:21], 0x401\n"
                         REP4(REP4("v_mov_b32_dpp %[T], %[A] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                   "v_mov_b32_dpp %[U], %[C] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                   "v_cndmask_b32 %[T], %[T], %[s], s[20:21]\n"
                                   "v_cndmask_b32 %[U], %[U], %[s], s[20:21]\n"
                                   "v_mul_f32 %[B], %[T], %[s]\n"
                                   "v_add_f32 %[A], %[B], %[s]\n"
                                   "v_mul_f32 %[C], %[U], %[s]\n"
                                   "v_add_f32 %[B], %[C], %[s]\n"))
                         : [A] "+v"(a), [B] "+v"(b), [C] "+v"(c), [T] "+v"(tmp), [U] "+v"(tmp2) : [s] "v"(seed) : "s20", "s21");
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a + b + q + tmp + tmp2 + c;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int ops)
{
    const int blocks = 1024, iters = 2000;
    float *out; long long *cyc;
    hipMalloc(&out, blocks * 64 * sizeof(float)); hipMalloc(&cyc, blocks * sizeof(long long));
    k<MODE><<<blocks, 64>>>(out, cyc, iters / 4, 1.0f);
    k<MODE><<<blocks, 64>>>(out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-64s %8.2f cycles per op (median wavefront), %8.2f (slowest)\n", name, (double)h[blocks / 2] / iters / ops, (double)h[blocks - 1] / iters / ops);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<7>("v_mul_f32 (reference: one issue slot)", 64);
    run<0>("sincos_ (scalar chains, compare + 2 v_cndmask quadrant swap)", 16);
    run<1>("sincos_pk (two-wide chains, bit-select quadrant swap)", 16);
    run<2>("v_bfe_i32", 64);
    run<3>("v_bfi_b32", 64);
    run<4>("v_bitop3_b32 (SGPR mask)", 64);
    run<5>("v_rndne_f32", 64);
    run<6>("v_pk_fma_f32 op_sel_hi:[1,0,1]", 64);
    run<8>("fused select: 2 x [cndmask_dpp(VCC), mul, add]      [cycles per GROUP of 6]", 16);
    run<9>("compiler's:   2 x [mov_dpp, cndmask(SGPR), mul, add] [cycles per GROUP of 8]", 16);
    return 0;
}
