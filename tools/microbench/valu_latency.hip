// Microbenchmark: issue / dependency cost of the VALU patterns the planner kernel is made of, for ONE
// wavefront alone on its SIMD (the small-batch regime) and for 2 wavefronts on one SIMD.
//   build: hipcc -O3 --offload-arch=gfx950 valu_latency.hip -o valu_latency ; run: ./valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(S) S S S S S S S S
#define REP64(S) REP8(REP8(S))

template <int MODE>
__global__ void bench(float *out, long long *cyc, int iters)
{
    float a = threadIdx.x * 1e-3f + 1.0f, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;
    const float k = 0.999f, m = 1e-4f;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v pa = {a, b}, pk = {k, k}, pm = {m, m};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 64 dependent fma
            asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(k), "v"(m));
        } else if (MODE == 1) {   // 2 independent chains, 32 each
            asm volatile(REP8(REP8("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n")) : "+v"(a), "+v"(b) : "v"(k), "v"(m));
        } else if (MODE == 2) {   // 4 independent chains
            asm volatile(REP8(REP8("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"))
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k), "v"(m));
        } else if (MODE == 3) {   // dependent add_dpp row_shr chain with the required s_nop 1
            asm volatile(REP64("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n") : "+v"(a) : "v"(m));
        } else if (MODE == 4) {   // two interleaved add_dpp chains + s_nop 0 (the x/y recurrence)
            asm volatile(REP64("v_add_f32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 0\n")
                         : "+v"(a), "+v"(b) : "v"(m));
        } else if (MODE == 5) {   // dependent v_rcp chain
            asm volatile(REP64("v_rcp_f32 %0, %0\n") : "+v"(a));
        } else if (MODE == 6) {   // dependent mul chain
            asm volatile(REP64("v_mul_f32 %0, %0, %1\n") : "+v"(a) : "v"(k));
        } else if (MODE == 7) {   // wave_shr add chain
            asm volatile(REP64("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n") : "+v"(a) : "v"(m));
        } else if (MODE == 8) {   // s_nop 0 only
            asm volatile(REP64("s_nop 0\n"));
        } else if (MODE == 9) {   // full IEEE division, dependent chain (compiler sequence)
#pragma unroll
            for (int j = 0; j < 64; ++j) a = b / a;
        } else if (MODE == 10) {  // cndmask_dpp select chain (V_SEG boundary form)
            asm volatile(REP64("v_add_f32 %1, %0, %2\n s_nop 1\n v_cndmask_b32_dpp %0, %1, %2, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(a), "+v"(b) : "v"(m) : "vcc");
        } else if (MODE == 11) {  // dependent v_cndmask
            asm volatile(REP64("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(a) : "v"(m) : "vcc");
        } else if (MODE == 12) {  // SALU chain
            int s = iters;
            asm volatile(REP64("s_add_u32 %0, %0, 1\n") : "+s"(s));
            if (s == 12345) a += 1.0f;
        } else if (MODE == 14) {  // dependent v_pk_fma_f32 chain (2 floats per lane per instruction)
            asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(pa) : "v"(pk), "v"(pm));
        } else if (MODE == 15) {  // dependent v_pk_mul_f32
            asm volatile(REP64("v_pk_mul_f32 %0, %0, %1\n") : "+v"(pa) : "v"(pk));
        } else if (MODE == 16) {  // dependent v_pk_add_f32
            asm volatile(REP64("v_pk_add_f32 %0, %0, %1\n") : "+v"(pa) : "v"(pm));
        } else if (MODE == 17) {  // pk_fma alternating with a plain fma on other registers
            asm volatile(REP64("v_pk_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %4, %5\n") : "+v"(pa), "+v"(b) : "v"(pk), "v"(pm), "v"(k), "v"(m));
        } else if (MODE == 18) {  // fma + taken s_branch to the next instruction (per pair)
            asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n s_branch 1f\n1:\n") : "+v"(a) : "v"(k), "v"(m));
        } else if (MODE == 19) {  // fma + never-taken conditional branch (per pair)
            asm volatile("s_cmp_eq_u32 0, 1\n" REP64("v_fma_f32 %0, %0, %1, %2\n s_cbranch_scc1 1f\n1:\n") : "+v"(a) : "v"(k), "v"(m) : "scc");
        } else if (MODE == 20) {  // 4 fma + taken branch (per group of 5)
            asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n s_branch 1f\n1:\n") : "+v"(a) : "v"(k), "v"(m));
        } else if (MODE == 21) {  // fma + taken conditional branch over 8 skipped instructions (per pair)
            asm volatile("s_cmp_eq_u32 0, 0\n" REP64("v_fma_f32 %0, %0, %1, %2\n s_cbranch_scc1 1f\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n1:\n") : "+v"(a) : "v"(k), "v"(m) : "scc");
        } else if (MODE == 22) {  // v_cmp -> SGPR pair, s_nop, v_cndmask on it (per triple)
            asm volatile(REP64("v_cmp_gt_f32 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %2, s[20:21]\n") : "+v"(a) : "v"(m), "v"(k) : "s20", "s21");
        } else if (MODE == 23) {  // v_cmp -> vcc, v_cndmask on vcc (per pair)
            asm volatile(REP64("v_cmp_gt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %2, vcc\n") : "+v"(a) : "v"(m), "v"(k) : "vcc");
        } else if (MODE == 24) {  // v_exp_f32 dependent
            asm volatile(REP64("v_exp_f32 %0, %0\n") : "+v"(a));
        } else if (MODE == 25) {  // v_readlane + v_writelane round trip
            int sreg;
            asm volatile(REP64("v_readlane_b32 %1, %0, 3\n s_nop 3\n v_writelane_b32 %0, %1, 5\n") : "+v"(a), "=s"(sreg));
        } else if (MODE == 26) {  // v_cmp -> SGPR, dependent s_and on it, v_cndmask on the s_and result (per triple)
            asm volatile(REP64("v_cmp_gt_f32 s[20:21], %0, %1\n s_and_b64 s[22:23], s[20:21], exec\n s_nop 0\n v_cndmask_b32 %0, %0, %2, s[22:23]\n") : "+v"(a) : "v"(m), "v"(k) : "s20", "s21", "s22", "s23");
        } else if (MODE == 27) {  // same instruction mix, the s_and independent of the v_cmp
            asm volatile(REP64("v_cmp_gt_f32 s[20:21], %0, %1\n s_and_b64 s[22:23], s[24:25], exec\n s_nop 0\n v_cndmask_b32 %0, %0, %2, s[20:21]\n") : "+v"(a) : "v"(m), "v"(k) : "s20", "s21", "s22", "s23", "s24", "s25");
        } else if (MODE == 28) {  // div_scale flag -> SGPR, s_mov to vcc, div_fmas (per triple + nops)
            asm volatile(REP64("v_div_scale_f32 %1, s[20:21], %0, %2, %0\n s_mov_b64 vcc, s[20:21]\n s_nop 3\n v_div_fmas_f32 %0, %1, %2, %0\n") : "+v"(a), "+v"(b) : "v"(k) : "s20", "s21", "vcc");
        } else if (MODE == 29) {  // div_scale flag -> vcc directly, div_fmas
            asm volatile(REP64("v_div_scale_f32 %1, vcc, %0, %2, %0\n s_nop 3\n v_div_fmas_f32 %0, %1, %2, %0\n") : "+v"(a), "+v"(b) : "v"(k) : "vcc");
        } else if (MODE == 30) {  // s_mov vcc then cndmask_dpp on it (SALU -> VALU)
            asm volatile(REP64("s_mov_b64 vcc, s[20:21]\n v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(a) : "v"(m) : "s20", "s21", "vcc");
        } else if (MODE == 31) {  // v_cmp -> vcc, s_cbranch_vccz untaken (per pair)
            asm volatile(REP64("v_cmp_gt_f32 vcc, %0, %0\n s_cbranch_vccnz 1f\n1:\n") : "+v"(a) : : "vcc");
        } else if (MODE == 32) {  // v_cmp -> sgpr, s_cmp on it, s_cbranch_scc untaken (per triple)
            asm volatile(REP64("v_cmp_gt_f32 s[20:21], %0, %0\n s_cmp_lg_u64 s[20:21], 0\n s_cbranch_scc1 1f\n1:\n") : "+v"(a) : : "s20", "s21", "scc");
        } else if (MODE == 13) {  // ds_bpermute dependent chain
#pragma unroll
            for (int j = 0; j < 64; ++j) a = __int_as_float(__builtin_amdgcn_ds_bpermute(((threadIdx.x + 1) & 63) << 2, __float_as_int(a)));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + pa.x + pa.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// whole-chip load: `blocks` single-wavefront workgroups; cycles per op (s_memtime) and ns per op (events)
template <int MODE>
void run_grid(const char *name, int ops_per_iter, int blocks, int threads)
{
    float *out; long long *cyc;
    hipMalloc(&out, (size_t)blocks * threads * sizeof(float)); hipMalloc(&cyc, blocks * sizeof(long long));
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    bench<MODE><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e0);
    bench<MODE><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= blocks;
    const double ops = (double)iters * ops_per_iter;
    printf("%-30s grid %5d x %3d thr: %.2f memtime-cycles/op, %.3f ns/op wall -> %.2f GHz-equivalent\n", name, blocks, threads,
           mean / ops, ms * 1e6 / ops, (mean / ops) / (ms * 1e6 / ops));
    hipFree(out); hipFree(cyc);
}

template <int MODE>
void run(const char *name, int ops_per_iter, int waves)
{
    float *out; long long *cyc;
    hipMalloc(&out, 1024 * sizeof(float)); hipMalloc(&cyc, 16 * sizeof(long long));
    const int iters = 2000;
    bench<MODE><<<1, 64 * waves>>>(out, cyc, iters);
    bench<MODE><<<1, 64 * waves>>>(out, cyc, iters);
    hipDeviceSynchronize();
    long long h = 0; hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-46s waves/CU %d: %.2f cycles per op\n", name, waves, (double)h / ((double)iters * ops_per_iter));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int blocks : {1, 256, 1024, 2048, 4096}) run_grid<0>("dependent v_fma_f32", 64, blocks, 64);
    for (int blocks : {256, 1024, 2048}) run_grid<9>("dependent IEEE division", 64, blocks, 64);
    for (int blocks : {256, 1024, 2048}) run_grid<4>("x/y round (2 add_dpp + nop)", 64, blocks, 64);
    for (int blocks : {256, 512}) run_grid<0>("dependent v_fma_f32", 64, blocks, 192);

    for (int waves : {1, 8}) {   // 1 wave; 4 waves = 1 per SIMD; 8 = 2 per SIMD
        run<0>("dependent v_fma_f32", 64, waves);
        run<1>("2 independent fma chains (per instr)", 128, waves);
        run<2>("4 independent fma chains (per instr)", 256, waves);
        run<6>("dependent v_mul_f32", 64, waves);
        run<3>("dep. v_add_f32_dpp row_shr + s_nop 1 (per pair)", 64, waves);
        run<7>("dep. v_add_f32_dpp wave_shr + s_nop 1 (per pair)", 64, waves);
        run<4>("x/y round: 2 add_dpp + s_nop 0 (per round)", 64, waves);
        run<10>("add + s_nop 1 + cndmask_dpp wave_shr (per round)", 64, waves);
        run<11>("dependent v_cndmask_b32", 64, waves);
        run<5>("dependent v_rcp_f32", 64, waves);
        run<9>("dependent IEEE division a = b / a", 64, waves);
        run<8>("s_nop 0", 64, waves);
        run<12>("dependent s_add_u32", 64, waves);
        run<13>("dependent ds_bpermute_b32", 64, waves);
        run<14>("dependent v_pk_fma_f32", 64, waves);
        run<15>("dependent v_pk_mul_f32", 64, waves);
        run<16>("dependent v_pk_add_f32", 64, waves);
        run<17>("pk_fma + fma alternating (per pair)", 64, waves);
        run<18>("fma + taken s_branch (per pair)", 64, waves);
        run<19>("fma + untaken s_cbranch (per pair)", 64, waves);
        run<20>("4 fma + taken s_branch (per group)", 64, waves);
        run<21>("fma + taken s_cbranch over 8 instrs (per pair)", 64, waves);
        run<22>("v_cmp->sgpr, s_nop 1, v_cndmask (per triple)", 64, waves);
        run<23>("v_cmp->vcc, s_nop 1, v_cndmask (per triple)", 64, waves);
        run<24>("dependent v_exp_f32", 64, waves);
        run<25>("readlane, s_nop 3, writelane (per triple)", 64, waves);
        run<26>("v_cmp->sgpr, dep. s_and, nop, cndmask (per 4)", 64, waves);
        run<27>("v_cmp->sgpr, indep. s_and, nop, cndmask (per 4)", 64, waves);
        run<28>("div_scale->sgpr, s_mov vcc, nop3, div_fmas (per 4)", 64, waves);
        run<29>("div_scale->vcc, nop3, div_fmas (per 3)", 64, waves);
        run<30>("s_mov vcc, v_cndmask (per pair)", 64, waves);
        run<31>("v_cmp->vcc, untaken s_cbranch_vccnz (per pair)", 64, waves);
        run<32>("v_cmp->sgpr, s_cmp, untaken s_cbranch_scc1 (per 3)", 64, waves);
    }
    return 0;
}
