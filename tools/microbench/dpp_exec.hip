// Does a DPP operand read the register of a lane that EXEC disables?  (gfx9 ISA: a disabled source lane is
// "invalid": bound_ctrl:0 substitutes 0, otherwise the destination lane keeps its old value.)
//   build: hipcc -O3 --offload-arch=gfx950 dpp_exec.hip -o dpp_exec ; run: ./dpp_exec
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(float *out)
{
    const int lane = threadIdx.x;
    float src = 100.0f + lane, a = -1.0f, b = -1.0f, c = -1.0f;
    // EXEC without lanes 0, 10, 20, ...: their neighbours 1, 11, 21, ... read a disabled lane
    asm volatile("s_mov_b64 s[20:21], exec\n"
                 "s_mov_b64 exec, %3\n"
                 "s_nop 4\n"
                 "v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"   // bound_ctrl:0 in the old syntax
                 "v_mov_b32_dpp %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %2, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                 "s_nop 4\n"
                 "s_mov_b64 exec, s[20:21]\n"
                 : "+v"(a), "+v"(b), "+v"(c) : "s"(0xfffbfeffbfeffbfeull), "v"(src) : "s20", "s21");
    out[lane] = a; out[64 + lane] = b; out[128 + lane] = c;
}

int main()
{
    float *d; hipMalloc(&d, 192 * sizeof(float));
    k<<<1, 64>>>(d);
    float h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 2, 9, 10, 11, 12, 16, 17, 20, 21}) printf("lane %2d: bound_ctrl %6.1f   keep-old %6.1f   add row_shr %6.1f\n", l, h[l], h[64 + l], h[128 + l]);
    return 0;
}
