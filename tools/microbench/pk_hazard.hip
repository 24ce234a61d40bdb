// Does a dependent chain of packed fp32 instructions need the s_nop hipcc inserts between them?
// Runs the chain with and without the nops (inline asm: hipcc pads nothing inside) next to the same chain
// in scalar instructions and compares all three bit for bit on every lane, many wavefronts, many rounds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#define REP8(S) S S S S S S S S
#define REP64(S) REP8(REP8(S))
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void k(const float *in, float *out, int rounds)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a0 = in[4 * i], a1 = in[4 * i + 1];
    const float k0 = in[4 * i + 2], m0 = in[4 * i + 3];
    v2f pa = {a0, a1}, pb = {a0, a1};
    const v2f pk = {k0, k0 * 0.75f}, pm = {m0, -m0};
    float s0 = a0, s1 = a1;
    const float k1 = k0 * 0.75f, m1 = -m0;
    for (int r = 0; r < rounds; ++r) {
        asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2\n") : "+v"(pa) : "v"(pk), "v"(pm));
        asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n v_pk_mul_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %2\n s_nop 0\n") : "+v"(pb) : "v"(pk), "v"(pm));
        asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n") : "+v"(s0) : "v"(k0), "v"(m0));
        asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n") : "+v"(s1) : "v"(k1), "v"(m1));
    }
    out[6 * i] = pa.x; out[6 * i + 1] = pa.y; out[6 * i + 2] = pb.x; out[6 * i + 3] = pb.y; out[6 * i + 4] = s0; out[6 * i + 5] = s1;
}

int main()
{
    const int blocks = 2048, threads = 64, n = blocks * threads;
    std::vector<float> h(4 * n);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = 0.5f + (s >> 8) * (1.0f / 16777216.0f) * 0.01f; }
    for (int i = 0; i < n; ++i) { h[4 * i + 2] = 0.99f + h[4 * i + 2] * 0.001f; h[4 * i + 3] = (h[4 * i + 3] - 0.5f) * 0.1f; }
    float *din, *dout;
    hipMalloc(&din, h.size() * sizeof(float)); hipMalloc(&dout, 6 * n * sizeof(float));
    hipMemcpy(din, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    k<<<blocks, threads>>>(din, dout, 50);
    hipDeviceSynchronize();
    std::vector<float> o(6 * n);
    hipMemcpy(o.data(), dout, o.size() * sizeof(float), hipMemcpyDeviceToHost);
    long bad_nonop = 0, bad_nop = 0;
    for (int i = 0; i < n; ++i) {
        if (memcmp(&o[6 * i], &o[6 * i + 4], 8)) ++bad_nonop;
        if (memcmp(&o[6 * i + 2], &o[6 * i + 4], 8)) ++bad_nop;
    }
    printf("packed chain WITHOUT s_nop differs from scalar on %ld of %d lanes; WITH s_nop on %ld; sample %g %g\n", bad_nonop, n, bad_nop, o[0], o[4]);
    return 0;
}
