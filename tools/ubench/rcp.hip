// Experiment: is rcp + one FMA residual step a correctly rounded 1.0f/b on gfx950, and how fast is it?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/rcp.hip -o rcp
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float rcp1(float b)
{
    float r = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp2(float b)
{
    float r = rcp1(b);
    float e = __builtin_fmaf(-b, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ bool rcp_ok(float b)
{
    unsigned u = __float_as_uint(b);
    unsigned e = (u >> 23) & 0xffu;
    return ((e - 2u) < 250u);  // 2..251: |b| in [2^-125, 2^125): result normal
}
__device__ __forceinline__ float rcp_guarded(float b)
{
    if (__builtin_expect(!rcp_ok(b), 0)) return 1.0f / b;
    return rcp1(b);
}

// counters: [0] mismatches of rcp1 in range, [1] of those with mantissa all ones, [2] mismatches rcp2, [3] tested
__global__ void k_check(unsigned long long* cnt, unsigned* examples)
{
    unsigned long long base = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 256ull;
    unsigned long long m1 = 0, m1a = 0, m2 = 0, n = 0;
    for (int i = 0; i < 256; i++) {
        unsigned u = (unsigned)(base + i);
        float b = __uint_as_float(u);
        if (!rcp_ok(b)) continue;
        n++;
        float want = 1.0f / b;
        float g1 = rcp1(b), g2 = rcp2(b);
        if (__float_as_uint(g1) != __float_as_uint(want)) {
            m1++;
            if ((u & 0x7FFFFFu) == 0x7FFFFFu) m1a++;
            else {
                unsigned long long k = atomicAdd(&cnt[4], 1ull);
                if (k < 16) examples[k] = u;
            }
        }
        if (__float_as_uint(g2) != __float_as_uint(want)) m2++;
    }
    atomicAdd(&cnt[0], m1);
    atomicAdd(&cnt[1], m1a);
    atomicAdd(&cnt[2], m2);
    atomicAdd(&cnt[3], n);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float b = 1.0001f + a[(i + 1) & 7] * 1e-3f;
            if (MODE == 0) a[i] = 1.0f / b + a[i] * 0.25f;
            if (MODE == 1) a[i] = rcp_guarded(b) + a[i] * 0.25f;
            if (MODE == 2) a[i] = __builtin_amdgcn_rcpf(b) + a[i] * 0.25f;
            if (MODE == 3) a[i] = b * 0.7f + a[i] * 0.25f;
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
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
    unsigned long long* d_cnt;
    unsigned* d_ex;
    hipMalloc(&d_cnt, 8 * sizeof(unsigned long long));
    hipMalloc(&d_ex, 16 * sizeof(unsigned));
    hipMemset(d_cnt, 0, 8 * sizeof(unsigned long long));
    hipMemset(d_ex, 0, 16 * sizeof(unsigned));
    hipLaunchKernelGGL(k_check, dim3(65536), dim3(256), 0, 0, d_cnt, d_ex);  // 65536*256*256 = 2^32
    unsigned long long c[8];
    unsigned ex[16];
    hipMemcpy(c, d_cnt, sizeof(c), hipMemcpyDeviceToHost);
    hipMemcpy(ex, d_ex, sizeof(ex), hipMemcpyDeviceToHost);
    printf("tested %llu  rcp1 mismatches %llu (mantissa all-ones: %llu, other: %llu)  rcp2 mismatches %llu\n", c[3], c[0],
           c[1], c[4], c[2]);
    for (int i = 0; i < 16 && i < (int)c[4]; i++) printf("  other example b=0x%08x\n", ex[i]);
    float* d_out;
    hipMalloc(&d_out, sizeof(float) * 256 * 2048);
    const int iters = 4000;
    int blocks = 256 * 8;  // 32 waves/CU
    printf("32 waves/CU, %d iters x8:  1/b+mad %.3f ms   guarded rcp1+mad %.3f ms   raw rcp+mad %.3f ms   mul+mad %.3f ms\n",
           iters, run<0>(blocks, iters, d_out), run<1>(blocks, iters, d_out), run<2>(blocks, iters, d_out),
           run<3>(blocks, iters, d_out));
    return 0;
}
