// Micro-benchmark: VALU wave-instruction throughput on gfx950 for the instruction mix K1 uses.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/valu.hip -o valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float fdiv_guarded(float a, float b)
{
    unsigned ea = (__float_as_uint(a) >> 23) & 0xffu, eb = (__float_as_uint(b) >> 23) & 0xffu;
    if (__builtin_expect(!(((ea - 67u) < 120u) & ((eb - 67u) < 120u)), 0)) return a / b;
    float r = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = a * r;
    float e2 = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e2, r, q);
    float e3 = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e3, r, q);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 0.001f + i;
    double d = seed;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) a[i] = a[i] * 1.0001f + 0.5f;              // mul + add (no contraction): 2 instr
            if (MODE == 1) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f); // fma: 1 instr
            if (MODE == 2) a[i] = a[i] / (1.0001f + a[(i + 1) & 7] * 1e-9f);  // IEEE division
            if (MODE == 6) a[i] = fdiv_guarded(a[i], 1.0001f + a[(i + 1) & 7] * 1e-9f);
            if (MODE == 3) a[i] = (a[i] > 3.0f) ? a[i] - 1.5f : a[i] + 0.25f;  // cmp + cndmask + 2 add
            if (MODE == 4) { d = d * 1.0000001 + 0.5; a[i] += (float)d; }      // f64 mul+add, cvt, add
        }
        if (MODE == 5) {  // dependent chain, 1 accumulator
            a[0] = a[0] * 1.0001f + 0.5f;
            a[0] = a[0] * 1.0001f + 0.5f;
            a[0] = a[0] * 1.0001f + 0.5f;
            a[0] = a[0] * 1.0001f + 0.5f;
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)d;
}

template <int MODE>
double run(int blocks, int iters, float* d_out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float* d_out;
    hipMalloc(&d_out, sizeof(float) * 256 * 8192);
    const int iters = 20000;
    const char* names[7] = {"mul+add x8 (16 valu/iter)", "fma x8 (8 valu/iter)", "fdiv x8", "cmp+cndmask+2add x8",
                            "f64 mul+add,cvt,add x8", "dependent mul+add chain x4", "guarded rcp+7fma division x8"};
    for (int wpc = 4; wpc <= 32; wpc *= 2) {  // waves per CU: blocks of 4 waves
        int blocks = 256 * wpc / 4;
        double ms[7] = {run<0>(blocks, iters, d_out), run<1>(blocks, iters, d_out), run<2>(blocks, iters / 10, d_out),
                        run<3>(blocks, iters, d_out), run<4>(blocks, iters, d_out), run<5>(blocks, iters, d_out),
                        run<6>(blocks, iters / 10, d_out)};
        printf("waves/CU %2d:", wpc);
        for (int m = 0; m < 7; m++) printf("  [%d] %.3f ms", m, ms[m]);
        printf("\n");
    }
    // derived: mode 0 issues 16 VALU per iteration per wave
    for (int m = 0; m < 7; m++) printf("mode %d = %s\n", m, names[m]);
    return 0;
}
