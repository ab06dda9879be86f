// Exhaustive checks of candidate "exact" instruction sequences on gfx950 (results decide what the engine may use):
//   A. sqrtf(x) over EVERY positive float: v_rsq_f32 + one / two FMA residual steps, v_sqrt_f32 + the +-1 ulp test without
//      the denormal scaling -- against the compiler's correctly rounded sqrtf
//   B. quot_fast(a, b, rcp_fast(b)) (sdm_device.h) for divisors whose significand is ALL ONES (quot_window_ok excludes
//      them): every such b with 2^-40 <= |b| < 2^41, a over all significands x 8 exponents x both signs
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/exact_ops.hip -o exact_ops
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float sqrt_rsq1(float x)
{
    const float r = __builtin_amdgcn_rsqf(x);
    const float y0 = x * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-y0, y0, x);
    return __builtin_fmaf(e, h, y0);
}
__device__ __forceinline__ float sqrt_rsq2(float x)
{
    const float r = __builtin_amdgcn_rsqf(x);
    const float y0 = x * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-y0, y0, x);
    const float y1 = __builtin_fmaf(e, h, y0);
    const float e1 = __builtin_fmaf(-y1, y1, x);
    return __builtin_fmaf(e1, h, y1);
}
__device__ __forceinline__ float sqrt_pm1(float x)
{
    const float y = __builtin_amdgcn_sqrtf(x);
    const float ym = __uint_as_float(__float_as_uint(y) - 1u), yp = __uint_as_float(__float_as_uint(y) + 1u);
    float r = y;
    if (__builtin_fmaf(-ym, y, x) <= 0.0f) r = ym;
    if (__builtin_fmaf(-yp, y, x) > 0.0f) r = yp;
    return r;
}

// cnt: [0..2] mismatches of the three forms over normal positive x, [3] tested, [4..5] lowest / highest exponent field
// with a mismatch of form 0, [6..7] the same for form 1
__global__ void k_sqrt(unsigned long long* cnt)
{
    const unsigned long long base = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 256ull;
    unsigned long long m0 = 0, m1 = 0, m2 = 0, n = 0;
    for (int i = 0; i < 256; i++) {
        const unsigned u = (unsigned)(base + i);
        const unsigned ex = u >> 23;
        if (ex < 1u || ex > 254u) continue;  // positive normal
        const float x = __uint_as_float(u);
        const float want = sqrtf(x);
        n++;
        if (__float_as_uint(sqrt_rsq1(x)) != __float_as_uint(want)) {
            m0++;
            atomicMin(&cnt[4], (unsigned long long)ex);
            atomicMax(&cnt[5], (unsigned long long)ex);
        }
        if (__float_as_uint(sqrt_rsq2(x)) != __float_as_uint(want)) {
            m1++;
            atomicMin(&cnt[6], (unsigned long long)ex);
            atomicMax(&cnt[7], (unsigned long long)ex);
        }
        if (__float_as_uint(sqrt_pm1(x)) != __float_as_uint(want)) {
            m2++;
            atomicMin(&cnt[8], (unsigned long long)ex);
            atomicMax(&cnt[9], (unsigned long long)ex);
        }
    }
    atomicAdd(&cnt[0], m0);
    atomicAdd(&cnt[1], m1);
    atomicAdd(&cnt[2], m2);
    atomicAdd(&cnt[3], n);
}

__device__ __forceinline__ float rcp_fast(float b)
{
    const float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float quot_fast(float a, float b, float r)
{
    float q0 = a * r;
    float e0 = __builtin_fmaf(-q0, b, a);
    float q1 = __builtin_fmaf(e0, r, q0);
    float e1 = __builtin_fmaf(-q1, b, a);
    return __builtin_fmaf(e1, r, q1);
}
__device__ __forceinline__ float quot_one(float a, float b, float r)
{
    float q0 = a * r;
    float e0 = __builtin_fmaf(-q0, b, a);
    return __builtin_fmaf(e0, r, q0);
}
// grid.y = divisor exponent field 87 .. 167 (81 values), both signs inside; a: every significand, exponent fields
// eb-3 .. eb+4 around the divisor's (quotients around 1 -- the window's edges are selftest 5's business), both signs.
// cnt: [0] two-step mismatches, [1] one-step mismatches, [2] tested
__global__ void k_quot_ones(unsigned long long* cnt)
{
    const unsigned eb = 87u + blockIdx.y;
    unsigned long long m2 = 0, m1 = 0, n = 0;
    for (unsigned sb = 0; sb < 2; sb++) {
        const float b = __uint_as_float((sb << 31) | (eb << 23) | 0x7FFFFFu);
        const float r = rcp_fast(b);
        for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 27);
             i += (unsigned long long)gridDim.x * blockDim.x) {
            const unsigned man = (unsigned)i & 0x7FFFFFu, de = ((unsigned)i >> 23) & 7u, sa = (unsigned)(i >> 26) & 1u;
            const unsigned ea = eb + de - 3u;
            if (ea < 87u || ea > 167u) continue;
            const float a = __uint_as_float((sa << 31) | (ea << 23) | man);
            const float want = a / b;
            n++;
            if (__float_as_uint(quot_fast(a, b, r)) != __float_as_uint(want)) m2++;
            if (__float_as_uint(quot_one(a, b, r)) != __float_as_uint(want)) m1++;
        }
    }
    atomicAdd(&cnt[0], m2);
    atomicAdd(&cnt[1], m1);
    atomicAdd(&cnt[2], n);
}

int main()
{
    unsigned long long *d, h[16];
    hipMalloc(&d, sizeof(h));
    for (int i = 0; i < 16; i++) h[i] = 0;
    h[4] = h[6] = h[8] = 1000;
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sqrt, dim3((1u << 31) / 256 / 256), dim3(256), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("sqrt over %llu positive normal floats: rsq+1 step %llu mismatches (exponent fields %llu..%llu), rsq+2 steps %llu (%llu..%llu), "
           "sqrt +-1ulp without scaling %llu (%llu..%llu)\n", h[3], h[0], h[4], h[5], h[1], h[6], h[7], h[2], h[8], h[9]);
    for (int i = 0; i < 16; i++) h[i] = 0;
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_quot_ones, dim3(1024, 81), dim3(256), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("quotients with an all-ones divisor significand: %llu tested, two-step form %llu mismatches, one-step form %llu\n", h[2], h[0],
           h[1]);
    return 0;
}
