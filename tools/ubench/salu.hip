// Micro-benchmark: scalar-ALU / branch issue capacity on gfx950 relative to VALU (is the scalar pipe shared per CU?).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/salu.hip -o salu
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: 32 VALU per iteration; 1: 32 SALU; 2: 32 VALU + 32 SALU interleaved; 3: 32 VALU + 8 exec-mask branches
// (v_cmp + s_and_saveexec + s_cbranch_execz + s_or exec), the shape of a guarded fallback.
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float a = seed + threadIdx.x * 0.001f;
    unsigned s = (unsigned)iters;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 32; i++) {
                asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a));
                if (MODE == 2) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc");
                if (MODE == 3 && (i & 3) == 3) {
                    asm volatile(
                        "v_cmp_lt_f32 vcc, %0, %0\n\t"
                        "s_and_saveexec_b64 s[30:31], vcc\n\t"
                        "s_cbranch_execz 1f\n\t"
                        "v_add_f32 %0, 2.0, %0\n"
                        "1:\n\t"
                        "s_or_b64 exec, exec, s[30:31]"
                        : "+v"(a) : : "vcc", "scc", "s30", "s31");  // s_and_saveexec / s_or write SCC
                }
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 32; i++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + (float)s;
}
template <int MODE>
double run(int blocks, int iters, float* d_out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 10, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    float* d_out;
    (void)hipMalloc(&d_out, sizeof(float) * 256 * 4096);
    const int iters = 20000;
    for (int wpc = 4; wpc <= 32; wpc *= 2) {
        int blocks = 256 * wpc / 4;
        fflush(stdout);
        printf("waves/CU %2d: 32 VALU %.3f ms | 32 SALU %.3f ms | 32 VALU + 32 SALU %.3f ms | 32 VALU + 8 guarded branches %.3f ms\n",
               wpc, run<0>(blocks, iters, d_out), run<1>(blocks, iters, d_out), run<2>(blocks, iters, d_out),
               run<3>(blocks, iters, d_out));
    }
    return 0;
}
