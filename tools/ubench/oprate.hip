// Micro-benchmark: issue cost of individual VALU opcodes on gfx950 (cycles per wave64 instruction per SIMD at
// 8 waves/SIMD, i.e. throughput).  Each kernel runs 64 independent-ish instances of ONE opcode per iteration over
// 4 accumulators (ILP 4), so the number is issue rate, not latency.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/oprate.hip -o oprate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP16(X) X X X X X X X X X X X X X X X X

#define KERNEL(NAME, ASM4)                                                                      \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed)             \
    {                                                                                           \
        float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x, c = seed * 3, d = seed * 5;   \
        for (int it = 0; it < iters; it++) {                                                    \
            REP16(asm volatile(ASM4 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");)           \
        }                                                                                       \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;                                    \
    }

#define KERNEL64(NAME, ASM4)                                                                    \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed)             \
    {                                                                                           \
        float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x, c = seed * 3, d = seed * 5;   \
        asm volatile("v_cvt_f64_f32 v[60:61], %0\n v_cvt_f64_f32 v[62:63], %1" : : "v"(a), "v"(b) : "v60", "v61", "v62", "v63"); \
        for (int it = 0; it < iters; it++) {                                                    \
            REP16(asm volatile(ASM4 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc", "v60", "v61", "v62", "v63");) \
        }                                                                                       \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;                                    \
    }

// four instructions per asm statement, one per accumulator
KERNEL(k_add_f32, "v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %1, %1, %2, %1\n v_fma_f32 %2, %2, %3, %2\n v_fma_f32 %3, %3, %0, %3")
KERNEL(k_and_b32, "v_and_b32 %0, 0x7fffffff, %0\n v_and_b32 %1, 0x7fffffff, %1\n v_and_b32 %2, 0x7fffffff, %2\n v_and_b32 %3, 0x7fffffff, %3")
KERNEL(k_add_u32, "v_add_u32 %0, 1, %0\n v_add_u32 %1, 1, %1\n v_add_u32 %2, 1, %2\n v_add_u32 %3, 1, %3")
KERNEL(k_max_u32, "v_max_u32 %0, %0, %1\n v_max_u32 %1, %1, %2\n v_max_u32 %2, %2, %3\n v_max_u32 %3, %3, %0")
KERNEL(k_max3_u32, "v_max3_u32 %0, %0, %1, %2\n v_max3_u32 %1, %1, %2, %3\n v_max3_u32 %2, %2, %3, %0\n v_max3_u32 %3, %3, %0, %1")
KERNEL(k_med3_f32, "v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %0\n v_med3_f32 %3, %3, %0, %1")
KERNEL(k_max_f32, "v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %0")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc")
KERNEL(k_cmp_f32, "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0")
KERNEL(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0")
KERNEL(k_cvt_i2f, "v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3")
KERNEL(k_cvt_f2i, "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3")
KERNEL(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0")
KERNEL(k_rcp_f32, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3")
KERNEL(k_sqrt_f32, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3")
KERNEL(k_floor_f32, "v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3")
KERNEL(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0")
KERNEL(k_divscale, "v_div_scale_f32 %0, vcc, %0, %1, %0\n v_div_scale_f32 %1, vcc, %1, %2, %1\n v_div_scale_f32 %2, vcc, %2, %3, %2\n v_div_scale_f32 %3, vcc, %3, %0, %3")
KERNEL(k_divfmas, "v_div_fmas_f32 %0, %0, %1, %2\n v_div_fmas_f32 %1, %1, %2, %3\n v_div_fmas_f32 %2, %2, %3, %0\n v_div_fmas_f32 %3, %3, %0, %1")
KERNEL(k_divfixup, "v_div_fixup_f32 %0, %0, %1, %2\n v_div_fixup_f32 %1, %1, %2, %3\n v_div_fixup_f32 %2, %2, %3, %0\n v_div_fixup_f32 %3, %3, %0, %1")
KERNEL64(k_cvt_f64, "v_cvt_f64_f32 v[60:61], %0\n v_cvt_f64_f32 v[62:63], %1\n v_cvt_f32_f64 %2, v[60:61]\n v_cvt_f32_f64 %3, v[62:63]")
KERNEL64(k_mul_f64, "v_mul_f64 v[60:61], v[60:61], v[62:63]\n v_mul_f64 v[62:63], v[62:63], v[60:61]\n v_add_f64 v[60:61], v[60:61], v[62:63]\n v_add_f64 v[62:63], v[62:63], v[60:61]")

// SGPR operands / lane masks (s[40:41] and s42 are scratch scalars set up by the macro below)
#define KERNELS(NAME, ASM4)                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed)             \
    {                                                                                           \
        float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x, c = seed * 3, d = seed * 5;   \
        asm volatile("s_mov_b64 s[40:41], 0x55\n s_mov_b32 s42, 0x3f800000" : : : "s40", "s41", "s42"); \
        for (int it = 0; it < iters; it++) {                                                    \
            REP16(asm volatile(ASM4 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc", "s40", "s41", "s42", "s44", "s45");) \
        }                                                                                       \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;                                    \
    }
KERNELS(k_add_sgpr, "v_add_f32 %0, s42, %0\n v_add_f32 %1, s42, %1\n v_add_f32 %2, s42, %2\n v_add_f32 %3, s42, %3")
KERNELS(k_cndmask_s, "v_cndmask_b32_e64 %0, %0, %1, s[40:41]\n v_cndmask_b32_e64 %1, %1, %2, s[40:41]\n v_cndmask_b32_e64 %2, %2, %3, s[40:41]\n v_cndmask_b32_e64 %3, %3, %0, s[40:41]")
KERNELS(k_cmp_s, "v_cmp_lt_f32_e64 s[44:45], %0, %1\n v_cmp_lt_f32_e64 s[44:45], %1, %2\n v_cmp_lt_f32_e64 s[44:45], %2, %3\n v_cmp_lt_f32_e64 s[44:45], %3, %0")
KERNELS(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc")
KERNELS(k_cmp_cnd_s, "v_cmp_lt_f32_e64 s[44:45], %0, %1\n v_cndmask_b32_e64 %2, %2, %3, s[44:45]\n v_cmp_lt_f32_e64 s[44:45], %2, %3\n v_cndmask_b32_e64 %0, %0, %1, s[44:45]")
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0")
KERNEL(k_sub_f32, "v_sub_f32 %0, %0, %1\n v_sub_f32 %1, %1, %2\n v_sub_f32 %2, %2, %3\n v_sub_f32 %3, %3, %0")
KERNEL(k_or_b32, "v_or_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_or_b32 %2, %2, %3\n v_or_b32 %3, %3, %0")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1\n v_lshl_or_b32 %1, %1, 1, %2\n v_lshl_or_b32 %2, %2, 1, %3\n v_lshl_or_b32 %3, %3, 1, %0")
KERNEL(k_add_abs, "v_add_f32_e64 %0, |%0|, %1\n v_add_f32_e64 %1, |%1|, %2\n v_add_f32_e64 %2, |%2|, %3\n v_add_f32_e64 %3, |%3|, %0")

KERNEL(k_add_lit, "v_add_f32 %0, 0xc2b40000, %0\n v_add_f32 %1, 0xc2b40000, %1\n v_add_f32 %2, 0xc2b40000, %2\n v_add_f32 %3, 0xc2b40000, %3")
KERNEL(k_and_lit, "v_and_b32 %0, 0x43b40000, %0\n v_and_b32 %1, 0x43b40000, %1\n v_and_b32 %2, 0x43b40000, %2\n v_and_b32 %3, 0x43b40000, %3")
KERNEL(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 4\n v_add_lshl_u32 %1, %1, %2, 4\n v_add_lshl_u32 %2, %2, %3, 4\n v_add_lshl_u32 %3, %3, %0, 4")
KERNEL(k_min_i32, "v_min_i32 %0, %0, %1\n v_min_i32 %1, %1, %2\n v_min_i32 %2, %2, %3\n v_min_i32 %3, %3, %0")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 31, %0\n v_ashrrev_i32 %1, 31, %1\n v_ashrrev_i32 %2, 31, %2\n v_ashrrev_i32 %3, 31, %3")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0")
KERNEL(k_ubyte, "v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte0 %2, %2\n v_cvt_f32_ubyte1 %3, %3")
KERNEL(k_sub_u32, "v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_sub_u32 %3, %3, %0")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %1, %1, %2, %3\n v_or3_b32 %2, %2, %3, %0\n v_or3_b32 %3, %3, %0, %1")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1")
KERNEL(k_fma_neg, "v_fma_f32 %0, -%0, %1, %2\n v_fma_f32 %1, -%1, %2, %3\n v_fma_f32 %2, -%2, %3, %0\n v_fma_f32 %3, -%3, %0, %1")

KERNEL(k_mix_fs, "v_add_f32 %0, 1.0, %0\n v_cvt_f32_i32 %1, %1\n v_add_f32 %2, 1.0, %2\n v_cvt_f32_i32 %3, %3")
KERNEL(k_mix_ft, "v_add_f32 %0, 1.0, %0\n v_rcp_f32 %1, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3")
KERNEL(k_mix_st, "v_cvt_f32_i32 %0, %0\n v_rcp_f32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3")

// pairing rules: two dependent chains A (%0) and B (%1), ordered AABB vs ABAB; and fast ops whose neighbours are slow ops of the SAME chain
KERNEL(k_pair_aabb, "v_add_f32 %0, 1.0, %0\n v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %1, 1.0, %1")
KERNEL(k_pair_abab, "v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1")
KERNEL(k_pair_dep_fs, "v_add_f32 %0, 1.0, %0\n v_cvt_f32_i32 %0, %0\n v_add_f32 %1, 1.0, %1\n v_cvt_f32_i32 %1, %1")
KERNEL(k_pair_cmp_add, "v_cmp_lt_f32 vcc, %0, %1\n v_add_f32 %2, 1.0, %2\n v_cmp_lt_f32 vcc, %1, %0\n v_add_f32 %3, 1.0, %3")
KERNELS(k_pair_cnd_add, "v_cndmask_b32_e64 %0, %0, %1, s[40:41]\n v_add_f32 %2, 1.0, %2\n v_cndmask_b32_e64 %1, %1, %0, s[40:41]\n v_add_f32 %3, 1.0, %3")
KERNEL64(k_pair_f64_add, "v_mul_f64 v[60:61], v[60:61], v[62:63]\n v_add_f32 %2, 1.0, %2\n v_add_f64 v[62:63], v[62:63], v[60:61]\n v_add_f32 %3, 1.0, %3")
KERNELS(k_pair_sgpr_add, "v_add_f32 %0, s42, %0\n v_add_f32 %2, 1.0, %2\n v_add_f32 %1, s42, %1\n v_add_f32 %3, 1.0, %3")

// packed fp32 (two results per lane and instruction; operands are aligned VGPR pairs): does one v_pk_* issue like one or like two?
#define KERNELPK(NAME, ASM4)                                                                    \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed)             \
    {                                                                                           \
        float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x, c = seed * 3, d = seed * 5;   \
        asm volatile("v_mov_b32 v56, %0\n v_mov_b32 v57, %1\n v_mov_b32 v58, %0\n v_mov_b32 v59, %1\n v_mov_b32 v60, %0\n v_mov_b32 v61, %1\n v_mov_b32 v62, %1\n v_mov_b32 v63, %0" \
                     : : "v"(a), "v"(b) : "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"); \
        for (int it = 0; it < iters; it++) {                                                    \
            REP16(asm volatile(ASM4 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");) \
        }                                                                                       \
        asm volatile("v_add_f32 %0, v56, %0\n v_add_f32 %0, v58, %0\n v_add_f32 %0, v60, %0\n v_add_f32 %0, v62, %0" : "+v"(a) : : "v56", "v58", "v60", "v62"); \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;                                    \
    }
KERNELPK(k_pk_fma, "v_pk_fma_f32 v[56:57], v[56:57], v[58:59], v[56:57]\n v_pk_fma_f32 v[58:59], v[58:59], v[60:61], v[58:59]\n v_pk_fma_f32 v[60:61], v[60:61], v[62:63], v[60:61]\n v_pk_fma_f32 v[62:63], v[62:63], v[56:57], v[62:63]")
KERNELPK(k_pk_mul, "v_pk_mul_f32 v[56:57], v[56:57], v[58:59]\n v_pk_mul_f32 v[58:59], v[58:59], v[60:61]\n v_pk_mul_f32 v[60:61], v[60:61], v[62:63]\n v_pk_mul_f32 v[62:63], v[62:63], v[56:57]")
KERNELPK(k_pk_add, "v_pk_add_f32 v[56:57], v[56:57], v[58:59]\n v_pk_add_f32 v[58:59], v[58:59], v[60:61]\n v_pk_add_f32 v[60:61], v[60:61], v[62:63]\n v_pk_add_f32 v[62:63], v[62:63], v[56:57]")
KERNELPK(k_pk_fma_add, "v_pk_fma_f32 v[56:57], v[56:57], v[58:59], v[56:57]\n v_add_f32 %2, 1.0, %2\n v_pk_fma_f32 v[60:61], v[60:61], v[62:63], v[60:61]\n v_add_f32 %3, 1.0, %3")
KERNELPK(k_pk_fma_cvt, "v_pk_fma_f32 v[56:57], v[56:57], v[58:59], v[56:57]\n v_cvt_f32_i32 %2, %2\n v_pk_fma_f32 v[60:61], v[60:61], v[62:63], v[60:61]\n v_cvt_f32_i32 %3, %3")

typedef void (*kern_t)(float*, int, float);
double run(kern_t k, int blocks, int iters, float* d_out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 10, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main(int argc, char** argv)
{
    float* d_out;
    (void)hipMalloc(&d_out, sizeof(float) * 256 * 4096);
    const int wps = argc > 1 ? atoi(argv[1]) : 8;  // waves per SIMD (1, 2, 4 or 8)
    const int iters = 4000, blocks = 256 * wps;    // 4 waves per workgroup, 256 CUs
    struct { const char* name; kern_t k; } tab[] = {
        {"v_add_f32", k_add_f32}, {"v_fma_f32", k_fma_f32}, {"v_and_b32", k_and_b32}, {"v_add_u32", k_add_u32},
        {"v_max_u32", k_max_u32}, {"v_max3_u32", k_max3_u32}, {"v_med3_f32", k_med3_f32}, {"v_max_f32", k_max_f32},
        {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_f32", k_cmp_f32}, {"v_cmp_lt_u32", k_cmp_u32},
        {"v_cvt_f32_i32", k_cvt_i2f}, {"v_cvt_i32_f32", k_cvt_f2i}, {"v_bfe_u32", k_bfe_u32}, {"v_lshlrev_b32", k_lshl},
        {"v_mul_u32_u24", k_mul_u24}, {"v_mul_lo_u32", k_mul_lo}, {"v_rcp_f32", k_rcp_f32}, {"v_sqrt_f32", k_sqrt_f32},
        {"v_floor_f32", k_floor_f32}, {"v_mov_b32", k_mov}, {"v_div_scale_f32", k_divscale}, {"v_div_fmas_f32", k_divfmas},
        {"v_div_fixup_f32", k_divfixup}, {"v_cvt f32<->f64 (2+2)", k_cvt_f64}, {"v_mul_f64/v_add_f64 (2+2)", k_mul_f64},
        {"v_add_f32 sgpr src", k_add_sgpr}, {"v_cndmask_b32 sgpr-pair mask", k_cndmask_s}, {"v_cmp -> sgpr pair", k_cmp_s},
        {"v_cmp vcc + v_cndmask (2+2)", k_cmp_cnd}, {"v_cmp sgpr + v_cndmask (2+2)", k_cmp_cnd_s}, {"v_mul_f32", k_mul_f32},
        {"v_sub_f32", k_sub_f32}, {"v_or_b32", k_or_b32}, {"v_lshl_or_b32", k_lshl_or}, {"v_add_f32 |abs| (VOP3)", k_add_abs},
        {"v_add_f32 literal", k_add_lit}, {"v_and_b32 literal", k_and_lit}, {"v_add_lshl_u32", k_add_lshl}, {"v_min_i32", k_min_i32},
        {"v_ashrrev_i32", k_ashr}, {"v_xor_b32", k_xor}, {"v_cvt_f32_ubyte0/1", k_ubyte}, {"v_sub_u32", k_sub_u32},
        {"v_or3_b32", k_or3}, {"v_fmac_f32", k_fmac}, {"v_fma_f32 neg", k_fma_neg},
        {"mix: 2 fast + 2 slow (add, cvt)", k_mix_fs}, {"mix: 3 fast + 1 rcp", k_mix_ft}, {"mix: 3 slow + 1 rcp", k_mix_st},
        {"2 dependent chains, order AABB", k_pair_aabb}, {"2 dependent chains, order ABAB", k_pair_abab},
        {"add->cvt dependent, 2 chains", k_pair_dep_fs}, {"cmp + independent add", k_pair_cmp_add},
        {"cndmask(sgpr mask) + independent add", k_pair_cnd_add}, {"f64 mul/add + independent add", k_pair_f64_add},
        {"add(sgpr src) + independent add", k_pair_sgpr_add},
        {"v_pk_fma_f32 (2 results each)", k_pk_fma}, {"v_pk_mul_f32", k_pk_mul}, {"v_pk_add_f32", k_pk_add},
        {"pk_fma + independent add (2+2)", k_pk_fma_add}, {"pk_fma + independent cvt (2+2)", k_pk_fma_cvt}};
    // instructions per SIMD: 8 waves * iters * 16 * 4
    const double n_per_simd = (double)wps * iters * 64.0;
    double base = 0;
    for (auto& t : tab) {
        double ms = run(t.k, blocks, iters, d_out);
        if (base == 0) base = ms;
        printf("%-28s %8.3f ms  %6.2f ns/instr/SIMD  x%.2f of v_add_f32\n", t.name, ms, ms * 1e6 / n_per_simd, ms / base);
        fflush(stdout);
    }
    return 0;
}
