// cu_spread.hip -- does every CU run the same instruction stream at the same speed?  (gfx950)
//
// One wavefront per SIMD (n_cus workgroups of 256 threads, registers claimed so that no second wavefront fits), each
// running the same dependent fp32 chain; per wavefront: shader cycles (s_memtime) and where it ran (HW_ID, XCC_ID).
// Two bodies: SMALL = a loop of 64 instructions (lives in the instruction buffer / one cache line set),
// BIG = a loop body of BODY_KB kilobytes of straight-line code (streams through the instruction cache like the
// planner kernels' passes do).  Prints the per-CU distribution and the slowest CUs.
//   hipcc -O3 --offload-arch=gfx950 cu_spread.hip -o cu_spread && ./cu_spread
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

#ifndef BODY_REPS
#define BODY_REPS 4096          // x 8 bytes (VOP3 v_fma_f32) = 32 KB per loop body
#endif

template <int REPS>
__global__ void __launch_bounds__(256, 1) chain(float *out, long long *cyc, unsigned *hwid, int iters, float seed)
{
    asm volatile("" ::: "a255");                       // > half of the register file: one wavefront per SIMD
    float a = seed + threadIdx.x * 1e-6f, b = 0.999f, c = 1e-3f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REPS; ++r) a = __builtin_fmaf(a, b, c);
        asm volatile("" : "+v"(a));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) {
        const size_t wv = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        cyc[wv] = t1 - t0;
        hwid[2 * wv] = hw;
        hwid[2 * wv + 1] = xcc;
    }
}

template <int REPS>
static void run(const char *name, int cus, long long total_ops)
{
    const int blocks = cus, threads = 256, waves = blocks * 4;
    const int iters = (int)(total_ops / REPS);
    float *out; long long *cyc; unsigned *hw;
    hipMalloc(&out, (size_t)blocks * threads * sizeof(float));
    hipMalloc(&cyc, waves * sizeof(long long));
    hipMalloc(&hw, 2 * waves * sizeof(unsigned));
    chain<REPS><<<blocks, threads>>>(out, cyc, hw, iters / 8 + 1, 1.0f);
    chain<REPS><<<blocks, threads>>>(out, cyc, hw, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<long long> h(waves); hipMemcpy(h.data(), cyc, waves * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<unsigned> id(2 * waves); hipMemcpy(id.data(), hw, 2 * waves * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<long long>> per_cu;
    for (int i = 0; i < waves; ++i) {
        const unsigned v = id[2 * i], x = id[2 * i + 1] & 0xf;
        per_cu[(x << 12) | (((v >> 13) & 7) << 8) | (((v >> 12) & 1) << 4) | ((v >> 8) & 0xf)].push_back(h[i]);
    }
    std::vector<std::pair<double, unsigned>> m;
    for (auto &kv : per_cu) { double s = 0; for (long long c : kv.second) s += (double)c; m.push_back({s / kv.second.size(), kv.first}); }
    std::sort(m.begin(), m.end());
    const double med = m[m.size() / 2].first;
    printf("%s: %d wavefronts on %zu CUs, %d instructions per loop body (%d KB), %.2f cycles per instruction (median CU)\n", name, waves,
           m.size(), REPS, REPS * 8 / 1024, med / ((double)iters * REPS));
    printf("  per-CU mean / median CU: min %.4f  10%% %.4f  90%% %.4f  99%% %.4f  max %.4f\n", m[0].first / med,
           m[m.size() / 10].first / med, m[m.size() * 9 / 10].first / med, m[m.size() * 99 / 100].first / med, m.back().first / med);
    printf("  slowest CUs (xcc, se, sh, cu):");
    for (size_t i = m.size() - 1; i + 8 >= m.size() && i < m.size(); --i)
        printf(" %.4f (%u,%u,%u,%u)", m[i].first / med, m[i].second >> 12, (m[i].second >> 8) & 7, (m[i].second >> 4) & 1, m[i].second & 0xf);
    printf("\n");
    hipFree(out); hipFree(cyc); hipFree(hw);
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    run<64>("SMALL", cus, 40000000ll);
    run<BODY_REPS>("BIG", cus, 40000000ll);
    run<2 * BODY_REPS>("BIG2", cus, 40000000ll);
    run<64>("SMALL again", cus, 40000000ll);
    return 0;
}
