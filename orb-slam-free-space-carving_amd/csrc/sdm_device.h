// sdm_device.h -- device-side arithmetic of the ProbabilityMapping path for gfx950.
//
// Every function here is the GPU statement of a piece of
// /root/reference/src/Modeler/ProbabilityMapping.cc ("PM.cc"); the cited lines give the float /
// double promotion pattern that must be kept for the support masks to stay bit-exact
// (SURVEY.md App. A.0).  Built with -ffp-contract=off: no mul-add may be fused, and hipcc's
// default correctly-rounded f32 divide/sqrt is relied upon.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <stddef.h>
#include <stdint.h>

#include "sdm_c.h"

namespace sdm {

// ---- device-resident tables ------------------------------------------------------------------
struct KfMeta {  // per keyframe slot; what PM.cc reads through KeyFrame accessors
    float fx, fy, cx, cy;  // include/KeyFrame.h:162
    float Tcw[12];         // src/KeyFrame.cc:70-121
    float I_stddev;        // PM.cc:457
    int uploaded;
    int pad[2];
};

struct RefConst {  // per reference keyframe of a batch
    int slot;
    float fx, fy, cx, cy;
    float mind, maxd;  // min_depth / max_depth as named at PM.cc:381-382
    int act_count;     // entries in the keyframe's active-pixel list (PM.cc:201 gate)
};

struct alignas(16) PairConst {  // per (reference, neighbour): hoisted out of the per-pixel code
    // -- the first PCV_FLOATS (20) dwords are what K1's searches read; k_search_fuse stages exactly those in LDS --
    float F[9];                 // F12, PM.cc:972-986
    float Rx[3];                // R21 row 0, PM.cc:859 == 890 == 643
    float Rz[3];                // R21 row 2
    float tx, tz;               // t21 x and z, PM.cc:860 == 891 == 644
    float rot;                  // median in-plane rotation, PM.cc:170-179
    float istd;                 // neighbour's I_stddev, PM.cc:457
    int clean;                  // bit 0: every angle the scan's two gates see lies in [0,360]: both keyframes' GradTheta planes
                                // (checked when the records are packed) and rot in [-360,360] -- the closed-form gates
                                // then hold for every candidate and the scan drops their per-candidate precondition;
                                // bit 1: line_quot_safe(F) -- the line's quotients need no per-lane guard
                                // bit 2: the search range at the principal point spans at least MASK_HINT_L columns (or cannot be
                                // told): K1's waves then ask themselves whether the mask scan pays (scan_masked)
    // -- the rest (K4, set-up) --
    float Ry[3];                // R21 row 1
    float ty;
    int nbr_slot;
    float nfx, nfy, ncx, ncy;  // neighbour intrinsics, PM.cc:675
    float pb[3];               // K4's approximate projection: error-bound constants (k4_proj_bounds, sdm_kernels.h)
};
static_assert(sizeof(PairConst) == 128, "PairConst must stay 128 B");

struct DevParams {  // sdm_params + host-precomputed (1/THETA), PM.cc:455-456
    float lambdaG, lambdaL, lambdaTheta;
    int lambdaN;
    double theta_var;
    double inv_theta;
    int fast_theta_div;  // x/theta_var via reciprocal + FMA correction (exhaustively verified for 0.23)
    int default_gates;   // lambdaL == 80 && lambdaTheta == 45: the closed-form angle gates apply
    // the scan's closed-form gates and approximate arg-min cost for thresholds OTHER than the defaults (the default ones are
    // literals in the code): constants derived on the host, each form checked on the device against the reference statement
    // for exactly these values when the parameters are set (sdm_set_params -> validate_params)
    float g2c;            // gate 2 fails  <=>  ||x-180| - 90| < g2c          (g2c = 90 - lambdaL rounded up; 10 for the default)
    unsigned g3lo, g3span;  // gate 3 fails  <=>  bits(x) - g3lo < g3span       (bits in (lambdaTheta, 360 - lambdaTheta))
    float inv_theta_f;    // (float)(1 / theta_var)
    int closed_ok;        // the two closed forms agree with PM.cc:416-431 for every d in [-400, 360): use them
    int approx_ok;        // fma(ge2, inv_theta_f, pe2) stays within COST_BAND / 4 steps of PM.cc:436: the arg-min may compare it
    int bins_ok;          // 0 <= lambdaTheta <= MASK_MAX_LAMBDA_THETA: the mask scan's six orientation planes hold the window
    int scan_mode;       // K1's scan per (wave, neighbour): 0 = chosen from the wave's range lengths and line slopes,
                         // 1 = always the batched scan, 2 = always the gradient-mask scan (tests; SDM_SCAN_MODE)
};

// x / d for a double x that was widened from a non-negative float, d = theta_var, r = RN(1/d).
// Two FMA residual corrections (Markstein): q2 == RN(x/d).  tests/test_gpu_arith.py checks it
// against the hardware-division sequence for ALL 2^32 float inputs at d = 0.23; for any other
// theta_var the engine keeps the plain division.
__device__ __forceinline__ double div_theta(double x, const double d, const double r, const bool fast)
{
    if (!fast) return x / d;
    double q0 = x * r;
    double e0 = __builtin_fma(-q0, d, x);
    double q1 = __builtin_fma(e0, r, q0);
    double e1 = __builtin_fma(-q1, d, x);
    double q2 = __builtin_fma(e1, r, q1);
    return (x < 1.0e300) ? q2 : q0;  // +inf -> +inf, NaN -> NaN (no residuals through inf - inf)
}

// ---- cv::fastAtan2 (PM.cc:414): OpenCV 3.x atan_f32 polynomial, degrees [0,360) ------------------
__host__ __device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    // the two branches of atan_f32 differ in which magnitude is divided by which and in the final 90 - p: one quotient and
    // one polynomial on selected operands (the same operations on the same values as either branch)
    const bool flat = ax >= ay;
    const float c = (flat ? ay : ax) / ((flat ? ax : ay) + (float)DBL_EPSILON);
    const float c2 = c * c;
    const float p = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    float a = flat ? p : 90.f - p;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---- 3x3 float algebra, left-to-right accumulation ----------------------------------------------
__host__ __device__ inline void mat3_mul(const float* A, const float* B, float* C)
{
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) {
            float p0 = A[i * 3 + 0] * B[0 * 3 + k];
            float p1 = A[i * 3 + 1] * B[1 * 3 + k];
            float p2 = A[i * 3 + 2] * B[2 * 3 + k];
            C[i * 3 + k] = (p0 + p1) + p2;
        }
}
__host__ __device__ inline void mat3_mul_bt(const float* A, const float* B, float* C)
{
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) {
            float p0 = A[i * 3 + 0] * B[k * 3 + 0];
            float p1 = A[i * 3 + 1] * B[k * 3 + 1];
            float p2 = A[i * 3 + 2] * B[k * 3 + 2];
            C[i * 3 + k] = (p0 + p1) + p2;
        }
}
__host__ __device__ inline void mat3_vec(const float* A, const float* v, float* o)
{
    for (int i = 0; i < 3; i++) {
        float p0 = A[i * 3 + 0] * v[0];
        float p1 = A[i * 3 + 1] * v[1];
        float p2 = A[i * 3 + 2] * v[2];
        o[i] = (p0 + p1) + p2;
    }
}

// R21/t21 (PM.cc:859-860) and F12 (PM.cc:972-986; skew matrix as src/LocalMapping.cc:711-716,
// since PM's GetSkewSymmetricMatrix is declared but never defined).  K^-1 in closed form.
__host__ __device__ inline void pair_geometry(const KfMeta& k1, const KfMeta& k2, float* F12,
                                              float* R21, float* t21)
{
    float R1[9], t1[3], R2[9], t2[3], tmp[3];
    for (int i = 0; i < 3; i++) {
        for (int k = 0; k < 3; k++) {
            R1[i * 3 + k] = k1.Tcw[i * 4 + k];
            R2[i * 3 + k] = k2.Tcw[i * 4 + k];
        }
        t1[i] = k1.Tcw[i * 4 + 3];
        t2[i] = k2.Tcw[i * 4 + 3];
    }
    mat3_mul_bt(R2, R1, R21);
    mat3_vec(R21, t1, tmp);
    for (int i = 0; i < 3; i++) t21[i] = (-tmp[i]) + t2[i];

    float R12[9], t12[3];
    mat3_mul_bt(R1, R2, R12);
    mat3_vec(R12, t2, tmp);
    for (int i = 0; i < 3; i++) t12[i] = (-tmp[i]) + t1[i];

    float t12x[9] = {0.f, -t12[2], t12[1], t12[2], 0.f, -t12[0], -t12[1], t12[0], 0.f};
    float K1ti[9] = {1.0f / k1.fx, 0.f, 0.f, 0.f, 1.0f / k1.fy, 0.f,
                     -k1.cx / k1.fx, -k1.cy / k1.fy, 1.f};
    float K2i[9] = {1.0f / k2.fx, 0.f, -k2.cx / k2.fx, 0.f, 1.0f / k2.fy, -k2.cy / k2.fy,
                    0.f, 0.f, 1.f};
    float A[9], B[9];
    mat3_mul(K1ti, t12x, A);
    mat3_mul(A, R12, B);
    mat3_mul(B, K2i, F12);
}

// R21.row(i) * xp, xp = (xp0, xp1, 1): PM.cc:866,868,894
__device__ __forceinline__ float row_dot_xp(const float* r, float xp0, float xp1)
{
    float p0 = r[0] * xp0;
    float p1 = r[1] * xp1;
    float p2 = r[2] * 1.0f;
    return (p0 + p1) + p2;
}

// ---- search record: everything one scan candidate needs in ONE 16-byte load ----------------------
//   .x = GradImg(y,x)   .y = GradTheta(y,x)   .z = GradImg(y+1,x)
//   .w = bits: im(y,x) | im(y+1,x) << 8
// bilinear<T>(img, yf, uj) of PM.cc:40-59 at integer uj only ever touches rows floor(yf) and
// floor(yf)+1 of column uj, which is exactly one record.
// y1f = (float)(y0 + 1), passed as floorf(yf) + 1.0f (the same value for 0 <= yf < 2^23; v_floor + v_add is
// cheaper than the int round trip: conversions issue at half the rate of add/mul, tools/ubench/oprate.hip)
__device__ __forceinline__ float rec_lerp_im(const float4& r, float y1f, float yf)
{
    unsigned w = __float_as_uint(r.w);
    float y0w = y1f - yf;
    float y1w = 1.0f - y0w;
    float v0 = (float)(int)(w & 0xffu), v1 = (float)(int)((w >> 8) & 0xffu);
    return v0 * y0w + v1 * y1w;
}
__device__ __forceinline__ float rec_lerp_grad(const float4& r, float y1f, float yf)
{
    float y0w = y1f - yf;
    float y1w = 1.0f - y0w;
    return r.x * y0w + r.z * y1w;
}

// GetPixelDepth, PM.cc:845-875 (Eq. 8) with the per-(pixel,pair) dot products hoisted.
__device__ __forceinline__ float pixel_depth(float uj, float fx, float cx, float rzxp, float rxxp,
                                             float tx, float tz)
{
    float ucx = uj - cx;
    float num1 = rzxp * ucx;
    float num2 = fx * rxxp;
    float denom1 = -tz * ucx;
    float denom2 = fx * tx;
    return (num1 - num2) / (denom1 + denom2);
}

// GetSearchRange, PM.cc:877-910
__device__ __forceinline__ void search_range(float fx, float cx, float rxxp, float rzxp, float tx,
                                             float tz, float mind, float maxd, int W, float& umin,
                                             float& umax)
{
    float x_min = rxxp * mind + tx, z_min = rzxp * mind + tz;
    float x_max = rxxp * maxd + tx, z_max = rzxp * maxd + tz;
    umin = fx * x_min / z_min + cx;
    umax = fx * x_max / z_max + cx;
    if (umin > umax) {
        float t = umax;
        umax = umin;
        umin = t;
    }
    float cols = (float)W;
    if (umin < 0) umin = 0;
    if (umax < 0) umax = 0;
    if (umin > cols) umin = cols;
    if (umax > cols) umax = cols;
}

// The scan only needs the integer ends ceil(umin), floor(umax) of that range (PM.cc:405).  For operands that are not
// NaN the swap of PM.cc:900-904 is (min, max) and the four clamps of PM.cc:906-909 are a median with 0 and cols;
// a -0 that min/max/median may return where the reference keeps +0 (or the other way round) becomes the same
// integer 0.  A NaN end means "no hypothesis" (N5) and is reported before the medians could swallow it.
__device__ __forceinline__ bool search_range_int(float fx, float cx, float rxxp, float rzxp, float tx, float tz,
                                                 float mind, float maxd, int W, int& lo, int& hi)
{
    float x_min = rxxp * mind + tx, z_min = rzxp * mind + tz;
    float x_max = rxxp * maxd + tx, z_max = rzxp * maxd + tz;
    const float u1 = fx * x_min / z_min + cx;
    const float u2 = fx * x_max / z_max + cx;
    const float cols = (float)W;
    const float umin = __builtin_amdgcn_fmed3f(fminf(u1, u2), 0.0f, cols);
    const float umax = __builtin_amdgcn_fmed3f(fmaxf(u1, u2), 0.0f, cols);
    lo = (int)ceilf(umin);
    hi = (int)floorf(umax);
    return !__builtin_isunordered(u1, u2);
}

// ---- the two angle gates of the scan, PM.cc:414-431 -----------------------------------------------------
// Reference statement (every step in float):
//   d = theta2 - ref;  if (d >= 360) d -= 360;  if (d < 0) d += 360;  if (d > 180) d = 360 - d;
//   gate 2 (ref = epipolar-line angle):  if (d > 90) d = 180 - d;  skip if d > lambdaL
//   gate 3 (ref = theta_pi + rot):       skip if d > lambdaTheta
__device__ __forceinline__ bool gate2_fails_ref(float d, float lambdaL)
{
    if (d >= 360) d -= 360;
    if (d < 0) d += 360;
    if (d > 180) d = 360 - d;
    if (d > 90) d = 180 - d;
    return d > lambdaL;
}
__device__ __forceinline__ bool gate3_fails_ref(float d, float lambdaTheta)
{
    if (d >= 360) d -= 360;
    if (d < 0) d += 360;
    if (d > 180) d = 360 - d;
    return d > lambdaTheta;
}
// Closed forms, valid when d < 360 (any d below that, including -Inf; NaN excluded by the caller's
// guard) and the thresholds are the defaults (80, 45).  With x = d + (d < 0 ? 360 : 0) -- the same
// rounding as the reference's "d += 360" -- every later step is exact (Sterbenz), so
//   gate 2 fails  <=>  x in (80,100) or (260,280)  <=>  |x-90| < 10  or  |x-270| < 10
//   gate 3 fails  <=>  45 < x < 315
// (x-90 and x-270 are exact wherever their magnitude is near 10).  sdm_selftest(3) compares both
// forms with the reference statement over dense and boundary inputs.
__device__ __forceinline__ float wrap_neg360(float d)
{
    unsigned neg = (unsigned)((int)__float_as_uint(d) >> 31);  // all ones iff the sign bit is set
    return d + __uint_as_float(neg & 0x43B40000u);            // + 360.0f or + 0.0f
}
__device__ __forceinline__ bool gate2_fails_fast(float d)
{
    float x = wrap_neg360(d);
    return (fabsf(x - 90.0f) < 10.0f) | (fabsf(x - 270.0f) < 10.0f);
}
__device__ __forceinline__ bool gate3_fails_fast(float d)
{
    float x = wrap_neg360(d);
    return (x > 45.0f) & (x < 315.0f);
}
// The same two decisions with ONE comparison each (comparisons, min/max and conversions issue at half the rate of
// add/sub/and on this part, tools/ubench/oprate.hip).
//   gate 2: x in (80,100) or (260,280)  <=>  ||x-180| - 90| < 10.  x-180 is exact wherever it matters (x >= 64: x and 180
//           are multiples of ulp(x) >= 2^-17 and |x-180| < 256... representable; below 64 the rounded difference stays
//           above 116, far from [80,100]), and |x-180| - 90 is exact for |x-180| in [45,180] (Sterbenz).
//   gate 3: 45 < x < 315 as an unsigned range test on the bit pattern (x is never negative after the wrap when d >= -360;
//           a negative or NaN x lands above the range: not "fails", like the float comparisons).
// sdm_selftest(3) compares these forms, too, with the reference statement over every float in [-400,400].
// the same forms with run-time constants (DevParams::g2c, g3lo, g3span): thresholds other than the defaults
__device__ __forceinline__ bool gate2_fails_k(float d, float g2c)
{
    const float x = wrap_neg360(d);
    return fabsf(fabsf(x - 180.0f) - 90.0f) < g2c;
}
__device__ __forceinline__ bool gate3_fails_k(float d, unsigned g3lo, unsigned g3span)
{
    const float x = wrap_neg360(d);
    return (__float_as_uint(x) - g3lo) < g3span;
}
__device__ __forceinline__ bool gate2_fails_fast1(float d)
{
    const float x = wrap_neg360(d);
    return fabsf(fabsf(x - 180.0f) - 90.0f) < 10.0f;
}
__device__ __forceinline__ bool gate3_fails_fast1(float d)
{
    const float x = wrap_neg360(d);
    return (__float_as_uint(x) - 0x42340001u) < (0x439D8000u - 0x42340001u);  // bits in (45.0f, 315.0f)
}

// ---- matching cost, PM.cc:436:  err = (float)((double)pe2 + (double)ge2 / THETA) ------------------------
// Fast path: s = pe2 + ge2 * (1/THETA) in double differs from the reference's double sum by at most
// a few ulp(double), so (float)s is the reference's value unless s lies within 2^-45 relative of a
// float rounding midpoint (low 29 mantissa bits == 0x10000000 +- 256) or err is outside the normal
// float range; those rare cases take the exact division.  sdm_selftest(2) checks it.
__device__ __forceinline__ float match_cost(float pe2, float ge2, const DevParams& prm)
{
    double s = (double)pe2 + (double)ge2 * prm.inv_theta;
    unsigned lo = (unsigned)__double2loint(s);
    bool risky = (((lo & 0x1FFFFFFFu) - 0x0FFFFF00u) <= 0x200u) | !(s > 1.0e-30) | !(s < 1.0e30);
    if (__builtin_expect(risky, 0)) s = (double)pe2 + (double)ge2 / prm.theta_var;
    return (float)s;
}

// (build-time knobs: SDM_ABLATE = n compiles one part of the search out -- diagnostic builds for the time
// attribution in DESIGN.md §5, never shipped; SDM_SCAN_UNROLL = records prefetched per batch)
#ifndef SDM_ABLATE
#define SDM_ABLATE 0
#endif
#ifndef SDM_SCAN_UNROLL
#define SDM_SCAN_UNROLL 4
#endif
constexpr int SCAN_UNROLL = SDM_SCAN_UNROLL;
typedef float v4f __attribute__((ext_vector_type(4)));  // one 128-bit VGPR tuple

struct SearchStats {
    unsigned long long searches, candidates, gate_pass;
};

// ---- correctly rounded reciprocal --------------------------------------------------------------------------
// 1.0f/b as v_rcp_f32 (<= 1 ulp) plus one FMA residual step.  On gfx950 this is bit-identical to the
// IEEE quotient for EVERY b with 2^-125 <= |b| < 2^125 (sdm_selftest(6) walks all 2^32 bit patterns);
// zero, denormal, huge, Inf and NaN operands take the plain division.  ~2.3x cheaper than the
// v_div_scale/v_div_fmas/v_div_fixup sequence (tools/ubench/rcp.hip).
__device__ __forceinline__ float rcp_exact(float b)
{
    const float ab = fabsf(b);
    if (__builtin_expect(!((ab >= 0x1p-125f) & (ab < 0x1p125f)), 0)) return 1.0f / b;
    const float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

#ifndef SDM_K1_OPT_ATAN_RCP
#define SDM_K1_OPT_ATAN_RCP 1  // the steep-line branch's 1/|y| through rcp_exact
#endif
// cv::fastAtan2(y, 1.0f) (PM.cc:414): fast_atan2_deg specialised for x == 1.  With ax = 1 the
// first branch divides by 1.0f + (float)DBL_EPSILON == 1.0f, i.e. c == ay exactly, so the common
// |a/b| <= 1 case needs no division; results are bit-identical to fast_atan2_deg(y, 1.0f).
__device__ __forceinline__ float fast_atan2_deg_x1(float y)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    float ay = fabsf(y);
    float a, c, c2;
    if (1.0f >= ay) {
        c = ay;
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
#if SDM_K1_OPT_ATAN_RCP
        c = rcp_exact(ay + (float)DBL_EPSILON);  // == 1.0f / (...) for every operand (sdm_selftest(6))
#else
        c = 1.0f / (ay + (float)DBL_EPSILON);
#endif
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (y < 0) a = 360.f - a;
    return a;
}

// ---- build-time switches of the K1 scan (A/B builds; every form is bit-identical, tests/test_gpu_arith.py) ----
// bit 0  no column clamp on the prefetch addresses (they stay inside the plane, see scan_segment)
// bit 1  two instantiations of the scan: pairs whose angles are all in [0,360] (PairConst::clean) skip the
//        per-candidate precondition of the closed-form gates
// bit 2  one-comparison forms of the two angle gates (gate2_fails_fast1 / gate3_fails_fast1)
// bit 3  match_cost: the magnitude guard as one unsigned range test on the high word
// bit 4  lerp weight from v_fract instead of floor + add + sub
// bit 5  search range: min/max through v_med3 (no canonicalising v_max), clamps on the integers
// bit 6  the in-plane-rotation wrap of PM.cc:425-426 with integer masks instead of compare + select
// bit 7  (lost: +4 %, tools/experiments/k1_buffer_loads.patch) records through a buffer descriptor: hardware range check
//        instead of the row clamp
// bit 8  (lost: +3 %, tools/experiments/k1_lds_constants.patch) per-pair constants read from an LDS copy into vector registers
// bit 9  GetFusion's 1/sigma^2 (double) from a float seed and three FMA steps instead of the IEEE double division
// bits 10, 11 (lost: +2 % / neutral) one exec-mask region per candidate; gate 2's |.| as an AND with a literal
// bit 12 (lost: +1.2 %, EXPERIMENTS.md) wave-uniform scan plan: the wave's longest range is covered by batches of 4 and 3
//        records (exact for every length >= 6) instead of by batches of 4 until the last lane is done -- a padding slot only
//        costs ~13 vector instructions (the wave branches past the gates), less than the plan's wave-wide maximum
// bit 13 arg-min of the scan from an APPROXIMATE matching cost in float (one FMA), the exact cost only where two costs
//        are too close to call (scan_batch; sdm_selftest(2) bounds the approximation) -- clean pairs with the default theta
// bit 14 the arg-min's four state updates as exec-masked moves (a real branch region) instead of four v_cndmask
// bit 15 the line's two quotients a/b, c/b (PM.cc:393, 407) in reciprocal form (one v_rcp_f32 + FMA steps, shared) for pairs
//        whose F12 entries are +0 or of moderate magnitude (PairConst::clean bit 1, line_quot_safe): no per-lane guard
// bit 16 sqrtf(ustar_var) (PM.cc:818) as v_rsq_f32 + one FMA residual step, exact for every x in [2^-100, 2^127)
//        (tools/ubench/exact_ops.hip walks all positive floats; sdm_selftest(6) repeats it)
// bit 17 the refinement's two records as one 16-byte gather each
#ifndef SDM_K1_OPT
#define SDM_K1_OPT 0x3a27f
#endif

// what one search reads of its PairConst, as float indices into the block `cv` points at: the PairConst itself
// (global memory -> scalar loads), or K1's compact LDS copy (k_search_fuse stages PCV_FLOATS floats per pair)
constexpr int PCV_FLOATS = 20;  // leading dwords of PairConst (5 x 16 bytes)
constexpr int CV_RX = 9, CV_RZ = 12, CV_TX = 15, CV_TZ = 16, CV_ROT = 17, CV_ISTD = 18;
static_assert(offsetof(PairConst, Rx) == 4 * CV_RX && offsetof(PairConst, Rz) == 4 * CV_RZ && offsetof(PairConst, tx) == 4 * CV_TX &&
                  offsetof(PairConst, tz) == 4 * CV_TZ && offsetof(PairConst, rot) == 4 * CV_ROT &&
                  offsetof(PairConst, istd) == 4 * CV_ISTD && offsetof(PairConst, Ry) == 4 * PCV_FLOATS,
              "the CV_* indices follow PairConst");

// matching cost, PM.cc:436, with the guard of match_cost() as ONE unsigned comparison: the fast double sum is used
// when 2^-99 <= s < 2^99 (inside the normal float range; negative, NaN and Inf patterns land outside) and its low 29
// bits are not within 256 of a float rounding midpoint
__device__ __forceinline__ float match_cost1(float pe2, float ge2, const DevParams& prm)
{
    double s = (double)pe2 + (double)ge2 * prm.inv_theta;
    const unsigned lo = (unsigned)__double2loint(s), hi = (unsigned)__double2hiint(s);
    const bool risky = (((lo & 0x1FFFFFFFu) - 0x0FFFFF00u) <= 0x200u) | ((hi - 0x39C00000u) >= (0x46200000u - 0x39C00000u));
    if (__builtin_expect(risky, 0)) s = (double)pe2 + (double)ge2 / prm.theta_var;
    return (float)s;
}

// ---- arg-min of the scan without the exact cost (SDM_K1_OPT bit 13) ------------------------------------------------------
// The scan only ever COMPARES matching costs (PM.cc:437 "err < old_err", PM.cc:446 "old_err < 1000000"): the value itself
// feeds nothing.  e~ = fma(ge2, (float)(1/THETA), pe2) is within 4 float steps of the reference's
// err = (float)((double)pe2 + (double)ge2 / THETA) for every non-negative pe2, ge2 (one rounding of the exact sum with a
// constant that is off by 2^-24 relative, against the reference's float rounding of a sum that is exact to 2^-52: at most
// 1.5 ulp + 1 ulp apart where ulps differ by a factor of two; sdm_selftest(2) measures the distance).  Non-negative floats
// order like their bit patterns, so when bits(e~_new) and bits(e~_best) differ by more than COST_BAND steps the exact
// costs compare the same way (strictly); inside the band both exact costs are evaluated and compared as the reference
// does -- ties and near-ties only.  NaN costs (NaN records) are never "better" under either statement; an Inf cost lies
// 2^23 steps above any best (<= 1e6).  The literal keeps the FMA in the full-rate issue class (no scalar-register operand).
constexpr float INV_THETA_DEFAULT_F = (float)(1.0 / 0.23);
constexpr unsigned COST_BAND = 16u;
__device__ __forceinline__ float match_cost_ref(float pe2, float ge2, const DevParams& prm)  // PM.cc:436 as written
{
    return (float)((double)pe2 + (double)ge2 / prm.theta_var);
}
__device__ __forceinline__ float match_cost_approx(float pe2, float ge2) { return __builtin_fmaf(ge2, INV_THETA_DEFAULT_F, pe2); }

// lerp weights of bilinear<T> at integer x (PM.cc:40-59): y0w = (floor(yf)+1) - yf, y1w = 1 - y0w.
// (floor(yf)+1) - yf = 1 - (yf - floor(yf)) in real arithmetic, yf - floor(yf) is exactly representable (what v_fract
// returns for yf >= 0), so both statements round the same real number: bit-identical (sdm_selftest(8)).
__device__ __forceinline__ float lerp_w0(float yf)
{
#if SDM_K1_OPT & 0x10
    return 1.0f - __builtin_amdgcn_fractf(yf);
#else
    return (floorf(yf) + 1.0f) - yf;
#endif
}
__device__ __forceinline__ float rec_lerp_im_w(const float4& r, float y0w)
{
    unsigned w = __float_as_uint(r.w);
    float y1w = 1.0f - y0w;
    float v0 = (float)(int)(w & 0xffu), v1 = (float)(int)((w >> 8) & 0xffu);
    return v0 * y0w + v1 * y1w;
}
__device__ __forceinline__ float rec_lerp_grad_w(const float4& r, float y0w)
{
    float y1w = 1.0f - y0w;
    return r.x * y0w + r.z * y1w;
}

// search range as integers, with fewer half-rate operations than search_range_int: min/max as v_med3 against -/+Inf
// (fminf/fmaxf cost a canonicalising v_max each), and the clamps of PM.cc:906-909 moved behind the float->int
// conversion: ceil/floor and a clamp to integer bounds commute, and v_cvt_i32_f32 saturates (+-Inf ends clamp like
// the float statement).  hi comes back clamped to W-1 (the scan never visits column W, N3).
__device__ __forceinline__ int cvt_i32_sat(float x)
{
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));  // saturating, NaN -> 0 (the C cast is undefined out of range)
    return r;
}
__device__ __forceinline__ bool search_range_int1(float fx, float cx, float rxxp, float rzxp, float tx, float tz,
                                                  float mind, float maxd, int W, int& lo, int& hi)
{
    float x_min = rxxp * mind + tx, z_min = rzxp * mind + tz;
    float x_max = rxxp * maxd + tx, z_max = rzxp * maxd + tz;
    const float u1 = fx * x_min / z_min + cx;
    const float u2 = fx * x_max / z_max + cx;
    const float mn = __builtin_amdgcn_fmed3f(u1, u2, -__builtin_inff());
    const float mx = __builtin_amdgcn_fmed3f(u1, u2, __builtin_inff());
    lo = min(max(cvt_i32_sat(ceilf(mn)), 0), W);
    hi = min(max(cvt_i32_sat(floorf(mx)), 0), W - 1);
    return !__builtin_isunordered(u1, u2);
}

// PM.cc:424-426: ang = th_pi + rot;  if (ang >= 360) ang -= 360;  if (ang < 0) ang += 360;
// Integer-mask form (add / sub / shift / and issue at the full rate, compare + select at half of it):
//   * a + 0.0f first turns a -0 into +0 (the reference would keep -0; the only consumer is d3 = theta2 - ang, where the
//     two differ only in the sign of a zero difference, and both signs pass gate 3 under either statement);
//   * "a >= 360": a non-negative float orders like its bit pattern; the sign mask excludes negative values; a positive
//     NaN subtracts 360 and stays NaN;
//   * "a < 0" is then the sign bit (no -0 left), wrap_neg360.
// sdm_selftest(8) compares it with the reference statement over every float in [-800, 800] and the special values.
__device__ __forceinline__ float wrap_once_360_ref(float a)
{
    if (a >= 360) a -= 360;
    if (a < 0) a += 360;
    return a;
}
__device__ __forceinline__ float wrap_once_360(float a)
{
#if SDM_K1_OPT & 0x40
    a = a + 0.0f;
    const unsigned bits = __float_as_uint(a);
    const unsigned lt360 = (unsigned)((int)(bits - 0x43B40000u) >> 31);  // all ones iff bits < bits(360.0f) (for a >= +0)
    const unsigned neg = (unsigned)((int)bits >> 31);
    unsigned keep = lt360 | neg;
    asm("" : "+v"(keep));  // keeps hipcc from turning the mask back into a compare + select
    a = a + __uint_as_float(~keep & 0xC3B40000u);  // - 360.0f or + 0.0f
    return wrap_neg360(a);
#else
    return wrap_once_360_ref(a);
#endif
}

// ---- float quotients in reciprocal form (K4) ------------------------------------------------------------------
// a/b as q = a*r with two FMA residual corrections (Markstein), r = 1/b correctly rounded (rcp_fast: v_rcp_f32 +
// one FMA step, see rcp_exact).  Bit-identical to the IEEE quotient whenever |a| and |b| lie in [2^-40, 2^41) --
// quot_window_ok states exactly that; K4 folds the same test over all operands of a neighbour into running integer
// min/max (NaN and Inf land above the window) instead of testing per quotient.  (Rounds 1-3 also excluded divisors whose
// significand is all ones -- the case where a Newton iteration cannot deliver the correctly rounded reciprocal.  rcp_fast IS
// correctly rounded for every divisor (sdm_selftest(6)), and with that Markstein's theorem has no exception:
// tools/ubench/exact_ops.hip finds no mismatch over 2 * 10^10 quotients by all-ones divisors, and the scale invariance of
// the sequence makes that sweep complete for the window.)
// sdm_selftest(5) compares quot_fast with the division over 2^33 operand pairs in and around the window.
constexpr unsigned QUOT_MAG_LO = 87u << 23;          // 2^-40
constexpr unsigned QUOT_MAG_HI = (168u << 23) - 1u;  // just below 2^41
__device__ __forceinline__ unsigned absbits(float x) { return __float_as_uint(x) & 0x7FFFFFFFu; }
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
__device__ __forceinline__ bool quot_window_ok(float a, float b)
{
    const unsigned ua = absbits(a), ub = absbits(b);
    return (ua >= QUOT_MAG_LO) & (ua <= QUOT_MAG_HI) & (ub >= QUOT_MAG_LO) & (ub <= QUOT_MAG_HI);
}

// ---- correctly rounded sqrtf without the compiler's scaling / +-1 ulp selection (SDM_K1_OPT bit 16) --------------------------
// y0 = x * rsq(x), one residual step y0 + (x - y0^2) * (rsq(x)/2): bit-identical to the IEEE square root for EVERY float in
// [2^-102, 2^128) on gfx950 (exhaustive: tools/ubench/exact_ops.hip, profiles/r04_exact_ops.txt; the 1.8 M mismatches all
// lie below 2^-102, where the residual underflows).  5 instructions (one transcendental) against 17.  Anything outside
// [2^-100, 2^127) -- zero, denormal, negative, Inf, NaN -- takes sqrtf.
__device__ __forceinline__ float sqrt_exact(float x)
{
#if SDM_K1_OPT & 0x10000
    if (__builtin_expect((__float_as_uint(x) - (27u << 23)) >= ((254u - 27u) << 23), 0)) return sqrtf(x);
    const float r = __builtin_amdgcn_rsqf(x);
    const float y0 = x * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-y0, y0, x);
    return __builtin_fmaf(e, h, y0);
#else
    return sqrtf(x);
#endif
}

// ---- the epipolar line's quotients without a per-lane guard (SDM_K1_OPT bit 15) ----------------------------------------------
// a, b, c = (x*F[0+k] + y*F[3+k]) + F[6+k] with integer pixel coordinates 0 <= x, y < 2^16.  If every entry of F12 is +0 or
// has a magnitude in [2^-36, 2^20] (line_quot_safe, decided once per pair), then
//   * each of a, b, c is a multiple of 2^-59 (products of an integer and a float keep the float's grid; sums and
//     roundings keep the coarser grid) and at most 3 * 2^36 in magnitude: zero, or inside [2^-59, 2^38];
//   * none of them is -0: the last addend F[6+k] is +0 or non-zero, and a sum that cancels rounds to +0;
//   * for b != 0 the reciprocal (v_rcp_f32 + one FMA step: correctly rounded, sdm_selftest(6)) and every quotient are
//     normal, all residuals stay above 2^-90: quot_fast is the IEEE quotient (two Markstein steps from a correctly rounded
//     reciprocal; sdm_selftest(5) covers the operand range, all-ones divisor significands included -- the exception
//     quot_window_ok makes for them is not needed: tools/ubench/exact_ops.hip, 2 * 10^10 quotients, no mismatch);
//     a zero numerator gives the quotient's zero with the IEEE sign (+0 numerator);
//   * b == 0 makes both forms produce Inf or NaN for a/b: the search is rejected by PM.cc:393 either way, c/b is not used.
__device__ __forceinline__ bool line_quot_safe(const float* F)
{
    bool ok = true;
    for (int i = 0; i < 9; i++) {
        const unsigned u = __float_as_uint(F[i]), mag = u & 0x7FFFFFFFu;
        ok = ok && (u == 0u || (mag >= ((127u - 36u) << 23) && mag <= ((127u + 20u) << 23)));
    }
    return ok;
}

struct ScanState {
    float old_err, best_pe, best_ge;
    int best_pixel;
};

// PM.cc:405-443: the scan over uj = lo..hi of one search.  Candidates are visited in increasing uj exactly as the
// reference does (the strict '<' at PM.cc:437 makes the lowest uj win ties), but their records are fetched NB at a
// time so that NB independent 16-byte gathers are in flight per lane.  Fetch rows are clamped into [1, H-2]
// (validity is decided separately from the unclamped value); fetch columns run to hi+3 <= W+2 at most (a lane only
// enters a batch while u0 <= hi), and (H-2)*W + W+2 < H*W, so every address stays inside the neighbour's plane without
// a column clamp.
// CLEAN: every angle is in [0,360] (PairConst::clean), so d2, d3 are in [-360,360] and the closed-form gates hold for
// every candidate; otherwise a candidate with d >= gate_lim (or NaN) takes the reference statement.
struct ScanConst {
    const char* nbase;
    unsigned W16;  // record pitch in bytes (< 2^24)
    float hlim_b;  // largest float below H-1: clamping yf to [1, hlim_b] leaves exactly the valid rows 1 <= yf < H-1 unchanged
                   // (and its integer part is at most H-2), so "clamped == original" is the row test of PM.cc:408 + N3
    float ab, cb, pixel, grad1, th_line, ang_pi_rot, gate_lim;
    int hi;
};
// One candidate that passed the row test (PM.cc:408 + N3): the gradient gate, the two angle gates, the matching cost and
// the arg-min update of PM.cc:411-443.  r = the candidate's record, yf = -((a/b)*uj + c/b) as the reference computes it.
template <bool STATS, bool CLEAN>
__device__ __forceinline__ void scan_candidate(const ScanConst& q, int uj, float yf, const float4& r, const DevParams& prm,
                                               ScanState& S, SearchStats* st)
{
    if (r.x < prm.lambdaG) return;                              // PM.cc:411
#if SDM_ABLATE == 4
    if (r.z > S.old_err) { S.best_pixel = uj; S.old_err = r.z; }
    return;
#endif
    const float d2 = r.y - q.th_line;     // PM.cc:415-416
    const float d3 = r.y - q.ang_pi_rot;  // PM.cc:427
    bool fail;
    if (CLEAN) {
#if SDM_K1_OPT & 0x04
        fail = gate2_fails_fast1(d2) | gate3_fails_fast1(d3);
#else
        fail = gate2_fails_fast(d2) | gate3_fails_fast(d3);
#endif
    } else {
        // the constants of the thresholds in force; gate_lim = 360 where they were validated (else -Inf: every candidate
        // takes the reference statement), and an angle pair outside the closed forms' domain takes it too
        fail = gate2_fails_k(d2, prm.g2c) | gate3_fails_k(d3, prm.g3lo, prm.g3span);
        if (__builtin_expect(!((d2 < q.gate_lim) & (d3 < q.gate_lim)), 0))
            fail = gate2_fails_ref(d2, prm.lambdaL) || gate3_fails_ref(d3, prm.lambdaTheta);
    }
    if (fail) return;  // PM.cc:421,431
    if (STATS) st->gate_pass++;
#if SDM_ABLATE == 3 || SDM_ABLATE == 4
    if (r.z > S.old_err) { S.best_pixel = uj; S.old_err = r.z; }
    return;
#endif
    const float y0w = lerp_w0(yf);                     // yf is in [1, H-1) here
    float pe = q.pixel - rec_lerp_im_w(r, y0w);        // PM.cc:433
    float ge = q.grad1 - rec_lerp_grad_w(r, y0w);      // PM.cc:434
#if SDM_K1_OPT & 0x2000
    if (CLEAN || prm.approx_ok) {  // (CLEAN implies theta_var == 0.23: the literal; otherwise wave-uniform)
        // S.old_err holds the APPROXIMATE cost of the best candidate so far (1e6 exactly before the first)
        const float e = CLEAN ? match_cost_approx(pe * pe, ge * ge) : __builtin_fmaf(ge * ge, prm.inv_theta_f, pe * pe);
        bool better = e < S.old_err;
        if (__builtin_expect((__float_as_uint(e) - __float_as_uint(S.old_err) + COST_BAND) <= 2u * COST_BAND, 0))
            better = match_cost_ref(pe * pe, ge * ge, prm) < match_cost_ref(S.best_pe * S.best_pe, S.best_ge * S.best_ge, prm);
        if (better) {
#if SDM_K1_OPT & 0x4000
            asm volatile("" ::: "memory");  // not if-convertible: the four moves run under the exec mask
#endif
            S.best_pixel = uj;
            S.old_err = e;
            S.best_pe = pe;
            S.best_ge = ge;
        }
        return;
    }
#endif
#if SDM_K1_OPT & 0x08
    float err = match_cost1(pe * pe, ge * ge, prm);    // PM.cc:436
#else
    float err = match_cost(pe * pe, ge * ge, prm);
#endif
    if (err < S.old_err) {  // PM.cc:437 strict: lowest uj wins ties
        S.best_pixel = uj;
        S.old_err = err;
        S.best_pe = pe;
        S.best_ge = ge;
    }
}

template <bool STATS, bool CLEAN, int NB>
__device__ __forceinline__ void scan_batch(const ScanConst& q, int u0, float u0f, const DevParams& prm, ScanState& S,
                                           SearchStats* st)
{
    float yfs[NB];
    unsigned long long rowok[NB];  // lane masks taken before the loads: scalar registers, not VGPRs
    v4f rs[NB];
    const unsigned c0 = (unsigned)u0 << 4;
#if !(SDM_K1_OPT & 0x01)
    const unsigned hi16 = (unsigned)max(q.hi, 0) << 4;
#endif
#pragma unroll
    for (int k = 0; k < NB; k++) {
        float yf = -(q.ab * (u0f + (float)k) + q.cb);  // PM.cc:407,433
        float yc = __builtin_amdgcn_fmed3f(yf, 1.0f, q.hlim_b);
#if SDM_K1_OPT & 0x01
        const unsigned off = __umul24((unsigned)(int)yc, q.W16) + c0;  // one v_mad_u32_u24; 16*k rides in the instruction
        const char* __restrict__ nb_k = q.nbase + 16 * k;
#else
        unsigned uc16 = min(c0 + 16u * k, hi16);                    // min(uj, hi) * 16
        unsigned off = __umul24((unsigned)(int)yc, q.W16) + uc16;     // one v_mad_u32_u24
        const char* __restrict__ nb_k = q.nbase;
#endif
        yfs[k] = yf;
        rowok[k] = __builtin_amdgcn_fcmpf(yc, yf, 1 /* ordered == */);
#if SDM_ABLATE == 5
        rs[k] = v4f{20.0f + (float)(off & 15u), 10.0f, 21.0f, __uint_as_float(0x6040u)};
#else
        rs[k] = *reinterpret_cast<const v4f*>(nb_k + off);
#endif
    }
    // keep each record one 16-byte gather issued here: without this hipcc splits the first record
    // into a 4-byte load plus a dependent 12-byte load behind the gradient gate (a second round trip)
#pragma unroll
    for (int k = 0; k < NB; k++)
        asm volatile("" : "+v"(rs[k]));
#pragma unroll
    for (int k = 0; k < NB; k++) {
        const int uj = u0 + k;
        if (STATS && uj <= q.hi) st->candidates++;
        const float4 r = make_float4(rs[k].x, rs[k].y, rs[k].z, rs[k].w);
        if (!((uj <= q.hi) & __builtin_amdgcn_inverse_ballot_w64(rowok[k]))) continue;  // PM.cc:408 + N3 (NaN rows fail)
        scan_candidate<STATS, CLEAN>(q, uj, yfs[k], r, prm, S, st);
    }
}

// every lane walks its own range in batches of SCAN_UNROLL (the wave loops until its longest lane is done)
template <bool STATS, bool CLEAN>
__device__ __forceinline__ void scan_segment(const ScanConst& q, int lo, const DevParams& prm, ScanState& S, SearchStats* st)
{
    float u0f = (float)lo;  // (float)uj without a conversion per candidate: exact below 2^24
    for (int u0 = lo; u0 <= q.hi; u0 += SCAN_UNROLL, u0f += (float)SCAN_UNROLL)
        scan_batch<STATS, CLEAN, SCAN_UNROLL>(q, u0, u0f, prm, S, st);
}

// ---- the scan over gradient-gate bit planes (long ranges) -------------------------------------------------------------------
// Far from the true match most candidates of a long range fail the gradient gate (PM.cc:411: ~80 % of an image does) or the
// orientation gate against the reference pixel (PM.cc:427-431: a random edge passes with probability 1/4), and the batched
// scan above still pays a 16-byte gather and the row / range tests for each of them.  Every keyframe slot therefore carries
// bit planes of those gates, written with its records (sdm_ingest.h): MASK_PLANES dwords per (row, 32-column word), the planes
// of one word side by side --
//   plane b < 16:  bit x = !(GradImg(y,x) < lambdaG)  and  (GradTheta(y,x) in [22.5 b, 22.5 (b+1))  or  GradTheta(y,x) outside [0,360))
//   planes 16-20:  bins 0-4 once more (a window that wraps past 360 is still six consecutive planes)
//   plane 21:      bit x = !(GradImg(y,x) < lambdaG)                                (the union; NaN angles are in every plane)
// A lane ORs the planes of the bins its orientation window [ang - lambdaTheta, ang + lambdaTheta] can touch (ang = the
// reference pixel's GradTheta + rot, wrapped, PM.cc:424-426) -- 8-byte loads that answer both gates for up to 33 consecutive
// columns of one image row -- and only the candidates whose bit is set are visited, in increasing uj, with the reference's
// statements (scan_candidate: the exact gates run again), so the arg-min is the reference's.  The listing only has to be a
// SUPERSET of the candidates that pass the gates:
//   * bins: a candidate passes gate 3 iff the angular distance of its GradTheta to ang is <= lambdaTheta (both in [0,360): the
//     reference's wrap, PM.cc:428-431; its float roundings move d by < 1e-4).  bin(t) = min(floor(t * 16/360), 15) in float is
//     weakly monotone in t, so the bins bin(lo) .. bin(hi) of a window [lo, hi] hold every angle inside it, whatever the
//     rounding at bin edges; the window is widened by MASK_ANG_MARGIN on both sides: at most MASK_WINDOW = 6 bins for the
//     default lambdaTheta = 45 and for any threshold up to MASK_MAX_LAMBDA_THETA.  Pairs that are not "clean" (an angle
//     outside [0,360], PairConst::clean) and wider thresholds use the union plane; pixels whose angle is outside [0,360) (caller-supplied planes only) sit in every plane.
//   * rows: the candidates uj = u .. u+n of a lane lie in ONE row when n is small enough:  yf(uj) = -((a/b)*uj + c/b) as computed
//     in float is weakly monotone in uj (a product and a sum by constants, each rounded to nearest: rounding is monotone) and
//     stays within E of the real line through the float values a/b, c/b -- E < 2^-11 + 2^-9 for |a/b| <= 4, uj < 2^14,
//     |yf| < 2^14 (one rounding of the product, one of the sum).  So with f = yf(u) - floor(yf(u)), moving AWAY from the row
//     boundary on the monotone side costs nothing, and towards the other boundary every uj with |a/b| * (uj - u) < room - 2E,
//     room = f (falling line) or 1 - f (rising line), has floor(yf(uj)) = floor(yf(u)).  MASK_EPS = 2^-6 > 2E; the column count
//     comes from an approximate reciprocal shortened by 2^-9, i.e. is never too large.
//   * a plane built under a smaller lambdaG only lists more.
// Candidates whose row test fails (PM.cc:408 + N3) are stepped over one at a time -- unless the LAST candidate of the range
// lies outside the image on the same side: by monotonicity so does everything in between, and the scan ends.
#ifndef SDM_MASK_MIN_L
#define SDM_MASK_MIN_L 24  // a wave takes the mask scan when at least half of its searching lanes have this many candidates ...
#endif
#ifndef SDM_MASK_MIN_L_UNION
#define SDM_MASK_MIN_L_UNION 48  // (without the orientation window: pairs that are not clean, non-default thresholds)
#endif
#ifndef SDM_MASK_MAX_SLOPE
#define SDM_MASK_MAX_SLOPE 0.25f  // ... and no lane's line is steeper than this (a row run is ~1/slope columns)
#endif
#ifndef SDM_MASK_NB
#define SDM_MASK_NB 2  // listed candidates evaluated per iteration of phase 2 (their gathers are in flight together)
#endif
#ifndef SDM_MASK_BINS
#define SDM_MASK_BINS 1  // 0: the union plane only (gradient gate)
#endif
constexpr int MASK_BINS = 16;
constexpr int MASK_WINDOW = 6;                               // bins a window of 2 * (45 + margin) degrees can touch
constexpr int MASK_UNION = MASK_BINS + MASK_WINDOW - 1;      // planes 16 .. 20 repeat bins 0 .. 4: a window is 6 CONSECUTIVE planes
constexpr int MASK_PLANES = MASK_UNION + 1;                  // 22 dwords per (row, 32-column word): 88 bytes
constexpr unsigned MASK_WORD_BYTES = 4u * MASK_PLANES;
constexpr float MASK_EPS = 0x1p-6f;
constexpr int MASK_HINT_L = SDM_MASK_MIN_L * 3 / 4;  // PairConst::clean bit 2
#ifndef SDM_MASK_CALL_MEAN_L
#define SDM_MASK_CALL_MEAN_L 32  // K1's mask-scan instantiation runs for calls whose mean range (principal point, all pairs) is at least this
#endif
constexpr int MASK_CALL_MEAN_L = SDM_MASK_CALL_MEAN_L;
constexpr float MASK_ANG_MARGIN = 0.01f;
// six consecutive bins from the one the window starts in cover at least 5 * 22.5 degrees from the window's start: windows of
// 2 * (lambdaTheta + margin) <= 112.5 degrees
constexpr float MASK_MAX_LAMBDA_THETA = 56.0f;
constexpr float MASK_BIN_SCALE = (float)MASK_BINS / 360.0f;
// the bin of an angle in [0,360): weakly monotone in t (product by a constant, floor, min)
__host__ __device__ __forceinline__ int mask_bin(float t)
{
    const int b = (int)(t * MASK_BIN_SCALE);
    return b < MASK_BINS - 1 ? b : MASK_BINS - 1;
}
struct MaskStats {
    unsigned long long waves, steps, row_mismatch;
};
struct MaskView {  // one keyframe slot's planes
    const char* base;    // row 0, word 0, plane 0
    unsigned row_pitch;  // bytes per image row: 32-bit words per row x MASK_WORD_BYTES
};
// BINNED: the lane's orientation window (six planes); otherwise the union plane
template <bool STATS, bool CLEAN, bool BINNED>
__device__ __forceinline__ void scan_masked(const ScanConst& q, const MaskView& mv, int lo, const DevParams& prm, ScanState& S,
                                            SearchStats* st, MaskStats* ms)
{
    const float yh = -(q.ab * (float)q.hi + q.cb);  // the last candidate's row coordinate
    const float inv_s = __builtin_amdgcn_rcpf(fabsf(q.ab)) * 0.998046875f;  // columns per unit of row room, shortened by 2^-9
    const bool falling = q.ab > 0.0f;  // yf decreases with uj
    // the first plane of the lane's window as a byte offset inside a word's planes (BINNED: ang is in [0,360))
    unsigned pl0 = 4u * (unsigned)MASK_UNION;
    if (BINNED) {
        float w0 = q.ang_pi_rot - (prm.lambdaTheta + MASK_ANG_MARGIN);
        if (w0 < 0.0f) w0 += 360.0f;
        pl0 = 4u * (unsigned)mask_bin(w0);
    }
    // Two phases per chunk of 64 candidates, so that the lanes of a wave -- whose row runs start and end at different
    // columns -- stay together: (1) the listed candidates of the chunk as ONE lane-private bit set, cand bit k = candidate
    // base + k passes the row test and its bit is set in one of the lane's planes; (2) SDM_MASK_NB listed candidates per lane
    // and iteration.
    for (int base = lo; base <= q.hi; base += 64) {
        const int chi = min(q.hi, base + 63);
        unsigned long long cand = 0ull;
        bool ended = false;
        int u = base;
        while (u <= chi) {
            const float yf = -(q.ab * (float)u + q.cb);  // PM.cc:407,433
            const float yc = __builtin_amdgcn_fmed3f(yf, 1.0f, q.hlim_b);
            if (!(yc == yf)) {  // PM.cc:408 + N3: this candidate's row is outside [1, H-2] (or NaN)
                const bool out_lo = (yf < 1.0f) & (yh < 1.0f), out_hi = (yf > q.hlim_b) & (yh > q.hlim_b);
                if (!(yf == yf) | out_lo | out_hi) {  // ... and so is every later one: the range ends here
                    if (STATS) st->candidates += (unsigned long long)(q.hi - u + 1);
                    ended = true;
                    break;
                }
                if (STATS) st->candidates++;
                u++;
                continue;
            }
            const float fr = __builtin_amdgcn_fractf(yf);
            const float room = (falling ? fr : 1.0f - fr) - MASK_EPS;
            const int n = max(cvt_i32_sat(room * inv_s), 0);  // (0 * Inf = NaN -> 0; a negative room -> 0)
            const int cover = min(min(n, chi - u), 32);       // columns u .. u+cover share the row; the two words hold >= 33
            const char* __restrict__ mp = mv.base + (__umul24((unsigned)(int)yc, mv.row_pitch) + __umul24((unsigned)u >> 5, MASK_WORD_BYTES) + pl0);
            unsigned mlo, mhi;
            if (BINNED) {  // six planes of this word and of the next one: 24 contiguous bytes each
                typedef unsigned mword4 __attribute__((ext_vector_type(4), aligned(4)));
                typedef unsigned mword2 __attribute__((ext_vector_type(2), aligned(4)));
                const mword4 a0 = *reinterpret_cast<const mword4*>(mp);
                const mword2 a1 = *reinterpret_cast<const mword2*>(mp + 16);
                const mword4 b0 = *reinterpret_cast<const mword4*>(mp + MASK_WORD_BYTES);
                const mword2 b1 = *reinterpret_cast<const mword2*>(mp + MASK_WORD_BYTES + 16);
                mlo = (a0.x | a0.y | a0.z) | (a0.w | a1.x | a1.y);
                mhi = (b0.x | b0.y | b0.z) | (b0.w | b1.x | b1.y);
            } else {
                mlo = *reinterpret_cast<const unsigned*>(mp);
                mhi = *reinterpret_cast<const unsigned*>(mp + MASK_WORD_BYTES);
            }
            unsigned long long bits = (((unsigned long long)mhi << 32) | (unsigned long long)mlo) >> ((unsigned)u & 31u);
            bits &= (2ull << cover) - 1ull;
            cand |= bits << (unsigned)(u - base);
            if (STATS) {
                st->candidates += (unsigned long long)(cover + 1);
                ms->steps++;
            }
            u += cover + 1;
        }
        if (STATS) {  // self-check of the row runs and the bins: no candidate that passes the three gates may be missing
            for (int k = 0; base + k <= (ended ? u - 1 : chi); k++) {
                const float yk = -(q.ab * (float)(base + k) + q.cb);
                if (!(__builtin_amdgcn_fmed3f(yk, 1.0f, q.hlim_b) == yk)) continue;
                const v4f r = *reinterpret_cast<const v4f*>(q.nbase + (__umul24((unsigned)(int)yk, q.W16) + ((unsigned)(base + k) << 4)));
                if (r.x < prm.lambdaG) continue;
                const bool fail = gate2_fails_ref(r.y - q.th_line, prm.lambdaL) || gate3_fails_ref(r.y - q.ang_pi_rot, prm.lambdaTheta);
                if (!fail && !((cand >> k) & 1ull)) ms->row_mismatch++;
            }
        }
        // phase 2: the listed candidates in increasing uj; each one's row comes from its own yf (it passed the row test)
        while (cand != 0ull) {
            int ujs[SDM_MASK_NB];
            float yfs[SDM_MASK_NB];
            v4f rs[SDM_MASK_NB];
            unsigned long long okm[SDM_MASK_NB];
#pragma unroll
            for (int k = 0; k < SDM_MASK_NB; k++) {
                okm[k] = __builtin_amdgcn_ballot_w64(cand != 0ull);  // (a lane mask in scalar registers, taken before the loads)
                const int uj = base + (cand != 0ull ? (int)__builtin_ctzll(cand) : 0);
                cand &= cand - 1ull;  // (0 stays 0)
                const float yfj = -(q.ab * (float)uj + q.cb);
                // a lane without a k-th candidate re-reads the chunk's first column at a clamped row: any valid address
                const float ycj = __builtin_amdgcn_fmed3f(yfj, 1.0f, q.hlim_b);
                ujs[k] = uj;
                yfs[k] = yfj;
                rs[k] = *reinterpret_cast<const v4f*>(q.nbase + (__umul24((unsigned)(int)ycj, q.W16) + ((unsigned)uj << 4)));
            }
#pragma unroll
            for (int k = 0; k < SDM_MASK_NB; k++)
                asm volatile("" : "+v"(rs[k]));
#pragma unroll
            for (int k = 0; k < SDM_MASK_NB; k++) {
                if (!__builtin_amdgcn_inverse_ballot_w64(okm[k])) continue;
                scan_candidate<STATS, CLEAN>(q, ujs[k], yfs[k], make_float4(rs[k].x, rs[k].y, rs[k].z, rs[k].w), prm, S, st);
            }
        }
        if (ended) break;
    }
}

// Wave-uniform plan (K1: all 64 lanes of the wave are here together): the wave's LONGEST range, Lmax candidates, is covered
// by n4 batches of four and n3 batches of three records -- exactly for every Lmax >= 6 and Lmax = 3, 4, with one padding
// slot at Lmax = 5 and up to two below 3 -- where the loop above pads every search to a multiple of four (16 % of the
// scanned slots on the App. D scene).  A lane enters a batch while its own range lasts; the order of the candidates,
// hence the arg-min, is the reference's.  The trip counts live in scalar registers.
__device__ __forceinline__ int wave_max_nonneg(int v)
{
    // row_shr 1, 2, 4, 8, then row_bcast:15 / :31: lane 63 ends up with the maximum.  Lanes without a source read 0.
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}
template <bool STATS, bool CLEAN>
__device__ __forceinline__ void scan_planned(const ScanConst& q, int lo, int Lmax, const DevParams& prm, ScanState& S,
                                             SearchStats* st)
{
    int n3 = (-Lmax) & 3;
    if (Lmax <= 3) n3 = 1;
    if (Lmax == 5) n3 = 2;
    const int n4 = max(Lmax - 3 * n3, 0) >> 2;
    int u0 = lo;
    float u0f = (float)lo;
    for (int i = 0; i < n4; i++, u0 += 4, u0f += 4.0f)
        if (u0 <= q.hi) scan_batch<STATS, CLEAN, 4>(q, u0, u0f, prm, S, st);
    for (int i = 0; i < n3; i++, u0 += 3, u0f += 3.0f)
        if (u0 <= q.hi) scan_batch<STATS, CLEAN, 3>(q, u0, u0f, prm, S, st);
}

// EpipolarSearch PM.cc:385-465 with ComputeInvDepthHypothesis PM.cc:806-829.
// nrec: the neighbour keyframe's record plane.  Returns true iff a hypothesis was produced
// (dh.supported).  Normative choices N3-N5 for the reference's undefined behaviour: DESIGN.md §3.
// cv: the pair's constants (layout CV_*); rcv: {fx, cx, min_depth, max_depth} of the reference keyframe;
// clean: PairConst::clean (wave-uniform).
// PLAN: the wave-uniform scan plan (scan_planned) -- every lane of the wave must make this call together, lanes without a
// pixel with on = false.  Without it (per-pixel entry points) `on` must be true.
// mv: the neighbour keyframe's gate bit planes (scan_masked), base == null: none; the lanes of the wave that reach the scan together choose
// between the two scans (DevParams::scan_mode).
// MASK: compile the mask scan in (K1 has an instantiation without it: on short ranges the mere presence of the second scan
// costs the batched one 6 % -- register allocation around the pair's scalar constants).
template <bool STATS, bool PLAN = false, bool MASK = false>
__device__ __forceinline__ bool epipolar_search(const float4* __restrict__ nrec, int W, int H,
                                                const float* __restrict__ cv, const float* __restrict__ rcv, int clean,
                                                bool on, int x, int y, float pixel, float grad1,
                                                float th_pi, float xp0, float xp1, const DevParams& prm, float& rho_o,
                                                float& sigma_o, float& best_u, float& best_v,
                                                SearchStats* st, const MaskView mv = MaskView{nullptr, 0u}, MaskStats* ms = nullptr)
{
    float fx = rcv[0], cx = rcv[1];
    const float mind = rcv[2], maxd = rcv[3];
    rho_o = 0.f;
    sigma_o = 0.f;
    best_u = 0.f;
    best_v = 0.f;
    if (STATS && on) st->searches++;
    const float* F = cv;
    float a = (float)x * F[0] + (float)y * F[3] + F[6];  // PM.cc:389-391
    float b = (float)x * F[1] + (float)y * F[4] + F[7];
    float c = (float)x * F[2] + (float)y * F[5] + F[8];
#if SDM_K1_OPT & 0x8000
    float ab, cb;
    if (clean & 2) {  // wave-uniform: line_quot_safe(F12)
        const float rb = rcp_fast(b);
        ab = quot_fast(a, b, rb);
        cb = quot_fast(c, b, rb);
    } else {
        ab = a / b;
        cb = c / b;
    }
    // PM.cc:393; a NaN line yields no hypothesis
    bool live = on & (ab >= -4) & (ab <= 4);
    if (!PLAN && !live) return false;
#else
    float ab = a / b;
    // PM.cc:393; a NaN line yields no hypothesis.  With PLAN the lanes stay together up to the scan (a lane that is out
    // gets an empty range) so that the wave-wide maximum below sees all of them
    bool live = on & (ab >= -4) & (ab <= 4);
    if (!PLAN && !live) return false;
    float cb = c / b;
#endif

    float rxxp = row_dot_xp(cv + CV_RX, xp0, xp1);
    float rzxp = row_dot_xp(cv + CV_RZ, xp0, xp1);
    float tx = cv[CV_TX], tz = cv[CV_TZ];
    int lo, hi;
#if SDM_K1_OPT & 0x20
    live &= search_range_int1(fx, cx, rxxp, rzxp, tx, tz, mind, maxd, W, lo, hi);  // PM.cc:404; N5
#else
    live &= search_range_int(fx, cx, rxxp, rzxp, tx, tz, mind, maxd, W, lo, hi);  // PM.cc:404; N5
    if (hi > W - 1) hi = W - 1;
#endif
    if (!PLAN && !live) return false;

    // PM.cc:414 cv::fastAtan2(-a/b, 1): (-a)/b == -(a/b) exactly in IEEE arithmetic; loop invariant
    float th_line = fast_atan2_deg_x1(-ab);
    float ang_pi_rot = wrap_once_360(th_pi + cv[CV_ROT]);  // PM.cc:424-426

    // closed-form gates need d < 360 and the default thresholds; with other thresholds the limit is -Inf and
    // every candidate takes the reference statement (a float limit keeps the test free of a uniform-bool VGPR)
    const float gate_lim = prm.closed_ok ? 360.0f : -__builtin_inff();
    // "no candidate yet": cost 1e6 (PM.cc:396) -- pe = 1000, ge = 0 ARE that cost under the exact statement, which is what the
    // approximate arg-min (bit 13) evaluates when its first near-tie involves the initial state; best_pixel = -1 marks it
    ScanState S = {1000000.0f, 1000.0f, 0.f, -1};
    const float hlim2 = (float)(H - 2);
    const char* __restrict__ nbase = reinterpret_cast<const char*>(nrec);
#if SDM_ABLATE == 10
    if (ang_pi_rot + th_line + (float)(hi - lo) != 12345.678f) live = false;  // keep the set-up alive, skip the rest
#endif
#if SDM_ABLATE == 11
    if (ab + cb + rxxp + rzxp != 12345.678f) live = false;  // only a,b,c, the two line quotients and the ray dot products
#endif
#if SDM_ABLATE == 6
    hi = lo - 1;
    S.old_err = ab;
    S.best_pixel = lo + 2;
#endif
    if (PLAN && !live) hi = lo - 1;  // an empty range: the lane enters no batch
    ScanConst sc;
    sc.nbase = nbase;
    sc.W16 = (unsigned)W << 4;
    sc.hlim_b = __uint_as_float(__float_as_uint((float)(H - 1)) - 1u);
    sc.ab = ab;
    sc.cb = cb;
    sc.pixel = pixel;
    sc.grad1 = grad1;
    sc.th_line = th_line;
    sc.ang_pi_rot = ang_pi_rot;
    sc.gate_lim = gate_lim;
    sc.hi = hi;
    if (PLAN) {
        const int Lmax = wave_max_nonneg(max(hi - lo + 1, 0));  // wave-uniform (a scalar register)
        if (Lmax > 0) {
#if SDM_K1_OPT & 0x02
            if ((clean & 1) && prm.default_gates && prm.fast_theta_div)  // wave-uniform
                scan_planned<STATS, true>(sc, lo, Lmax, prm, S, st);
            else
#endif
                scan_planned<STATS, false>(sc, lo, Lmax, prm, S, st);
        }
        if (!live) return false;
    } else {
        // which scan: a wave-uniform choice over the lanes that got here (the others returned above).  The mask scan pays
        // when ranges are long -- most of their candidates fail the gradient gate -- and lines are flat enough for a row run
        // to span many columns
        bool masked = false;
        const unsigned long long here = __builtin_amdgcn_ballot_w64(true);
#ifndef SDM_MASK_DISABLE  // (A/B builds: the batched scan alone, as rounds 1-4 shipped it)
        // clean bit 2: the pair's range at the principal point is long enough for the question to be worth three ballots
        if (MASK && mv.base != nullptr && prm.scan_mode != 1 && ((clean & 4) || prm.scan_mode == 2)) {
            // the orientation window needs ang in [0,360) -- a clean pair -- and the default thresholds the planes are cut for;
            // the gradient plane alone pays later
            const bool binned = (clean & 1) && prm.bins_ok && SDM_MASK_BINS;
            const int min_l = binned ? SDM_MASK_MIN_L : SDM_MASK_MIN_L_UNION;
            const unsigned long long lng = __builtin_amdgcn_ballot_w64(hi - lo + 1 >= min_l);
            const unsigned long long steep = __builtin_amdgcn_ballot_w64(fabsf(ab) > SDM_MASK_MAX_SLOPE);
            masked = prm.scan_mode == 2 || (2 * __popcll(lng) >= __popcll(here) && steep == 0ull);
        }
#endif
        if (MASK && masked) {
            if (STATS && (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (int)__builtin_ctzll(here)) ms->waves++;
            // the orientation window needs ang in [0,360) -- a clean pair -- and a lambdaTheta whose window six planes hold
#if SDM_K1_OPT & 0x02
            if ((clean & 1) && prm.default_gates && prm.fast_theta_div)  // wave-uniform
                scan_masked<STATS, true, SDM_MASK_BINS != 0>(sc, mv, lo, prm, S, st, ms);
            else
#endif
                if ((clean & 1) && prm.bins_ok && SDM_MASK_BINS)
                scan_masked<STATS, false, true>(sc, mv, lo, prm, S, st, ms);
            else
                scan_masked<STATS, false, false>(sc, mv, lo, prm, S, st, ms);
        } else {
#if SDM_K1_OPT & 0x02
            if ((clean & 1) && prm.default_gates && prm.fast_theta_div)  // wave-uniform
                scan_segment<STATS, true>(sc, lo, prm, S, st);
            else
#endif
                scan_segment<STATS, false>(sc, lo, prm, S, st);
        }
    }
    const float best_pe = S.best_pe, best_ge = S.best_ge;
    const int best_pixel = S.best_pixel;
    // PM.cc:446 "old_err < 1000000": old_err starts at 1e6 and only ever takes a smaller cost, so the test says "some
    // candidate was taken" (uj >= 0)
    if (best_pixel < 0) return false;
#if SDM_ABLATE == 2
    rho_o = S.old_err + best_pe + (float)best_pixel;
    sigma_o = best_ge + 1.0f;
    return true;
#endif


    int up = best_pixel + 1, um = best_pixel - 1;  // PM.cc:449-450
    if (um < 0 || up > W - 1) return false;
    float yfp = -(ab * (float)up + cb);
    float yfm = -(ab * (float)um + cb);
    float fyp = floorf(yfp), fym = floorf(yfm);
    if (!(fyp >= 0.0f && fyp <= hlim2)) return false;
    if (!(fym >= 0.0f && fym <= hlim2)) return false;
    int y0p = (int)fyp, y0m = (int)fym;
#if SDM_K1_OPT & 0x20000
    // one 16-byte gather per record (hipcc otherwise splits each into a 4-byte and an 8-byte load: .y is not used here)
    v4f vp = *reinterpret_cast<const v4f*>(nbase + ((__umul24((unsigned)y0p, (unsigned)W) + (unsigned)up) << 4));
    v4f vm = *reinterpret_cast<const v4f*>(nbase + ((__umul24((unsigned)y0m, (unsigned)W) + (unsigned)um) << 4));
    asm volatile("" : "+v"(vp));
    asm volatile("" : "+v"(vm));
    const float4 rp = make_float4(vp.x, vp.y, vp.z, vp.w), rm = make_float4(vm.x, vm.y, vm.z, vm.w);
#else
    float4 rp = *reinterpret_cast<const float4*>(nbase + ((__umul24((unsigned)y0p, (unsigned)W) + (unsigned)up) << 4));
    float4 rm = *reinterpret_cast<const float4*>(nbase + ((__umul24((unsigned)y0m, (unsigned)W) + (unsigned)um) << 4));
#endif
    const float y1p = fyp + 1.0f, y1m = fym + 1.0f;
    float g = (rec_lerp_im(rp, y1p, yfp) - rec_lerp_im(rm, y1m, yfm)) / 2;      // PM.cc:452
    float q = (rec_lerp_grad(rp, y1p, yfp) - rec_lerp_grad(rm, y1m, yfm)) / 2;  // PM.cc:453
    const double inv_theta = prm.inv_theta;
    float gg = g * g;
    float denom = (float)((double)gg + inv_theta * (double)q * (double)q);  // PM.cc:455
    float gpe = g * best_pe;
    float ustar = (float)((double)best_pixel +
                          ((double)gpe + inv_theta * (double)q * (double)best_ge) / (double)denom);
    float ustar_var = 2 * cv[CV_ISTD] * cv[CV_ISTD] / denom;  // PM.cc:457
    best_u = ustar;
    best_v = -(ab * ustar + cb);  // PM.cc:460

    // ComputeInvDepthHypothesis PM.cc:806-829
    float d0 = pixel_depth(ustar, fx, cx, rzxp, rxxp, tx, tz);
    float s = sqrt_exact(ustar_var);
    float dmin = pixel_depth(ustar - s, fx, cx, rzxp, rxxp, tx, tz);
    float dmax = pixel_depth(ustar + s, fx, cx, rzxp, rxxp, tx, tz);
    float e1 = fabsf(dmax - d0), e2 = fabsf(dmin - d0);
    rho_o = d0;
    sigma_o = (e1 < e2) ? e2 : e1;  // cv::max(a,b) = (a<b)?b:a
    return true;
}

// ChiTest, PM.cc:912-924 (both overloads share this arithmetic)
__device__ __forceinline__ bool chi_test(float a, float b, float sa, float sb)
{
    float d = a - b;
    float num = d * d;
    float chi = num / (sa * sa) + num / (sb * sb);
    return (double)chi < 5.99;
}

// one term of GetFusion (PM.cc:936-937 / 956-957): pow(sigma,2) is double, the sums are float.
// The two double quotients rho/s2 and 1/s2 share their divisor: r = 1/s2 is a true (correctly
// rounded) division and rho/s2 is recovered from it with two FMA residual corrections
// (Markstein: with r = RN(1/b) and q within an ulp of a/b, fma(fma(-q,b,a), r, q) = RN(a/b); b = s2
// is the exact square of a float, so its significand is never all ones).  Operands outside the
// comfortable range take the plain division.  sdm_selftest(4) compares the two forms.
__device__ __forceinline__ double div_by_with_rcp(double a, double b, double r)
{
    double q0 = a * r;
    double e0 = __builtin_fma(-q0, b, a);
    double q1 = __builtin_fma(e0, r, q0);
    double e1 = __builtin_fma(-q1, b, a);
    return __builtin_fma(e1, r, q1);
}
__device__ __forceinline__ void fusion_terms(float rho, float sg, double& t_rho, double& t_one)
{
    double s2 = (double)sg * (double)sg;
    const float ar = fabsf(rho);
#if SDM_K1_OPT & 512
    // 1.0 / s2 without the IEEE double division: float seed (<= 2^-21), two Newton steps (2^-42, then faithful),
    // and one residual step that rounds correctly -- s2 = sigma^2 has at most 48 significant bits, so its
    // significand is never all ones (the one case that step misses).  sdm_selftest(4) compares both terms with
    // the plain divisions.  Outside 2^-100 < s2 < 2^100 (float square in range) the divisions themselves.
    const unsigned hi = (unsigned)__double2hiint(s2);
    const bool safe = ((hi - 0x39B00000u) < (0x46400000u - 0x39B00000u)) & (ar > 1.0e-30f) & (ar < 1.0e30f);
    if (safe) {
        double y = (double)__builtin_amdgcn_rcpf(sg * sg);
        double e = __builtin_fma(-s2, y, 1.0);
        y = __builtin_fma(e, y, y);
        e = __builtin_fma(-s2, y, 1.0);
        y = __builtin_fma(e, y, y);
        e = __builtin_fma(-s2, y, 1.0);
        t_one = __builtin_fma(e, y, y);
        t_rho = div_by_with_rcp((double)rho, s2, t_one);
    } else {
        t_one = 1.0 / s2;
        t_rho = (double)rho / s2;
    }
#else
    t_one = 1.0 / s2;
    const bool safe = (s2 > 1.0e-60) & (s2 < 1.0e60) & (ar > 1.0e-30f) & (ar < 1.0e30f);
    t_rho = safe ? div_by_with_rcp((double)rho, s2, t_one) : (double)rho / s2;
#endif
}
__device__ __forceinline__ void fusion_accum(float rho, float sg, float& pjsj, float& rsj)
{
    double t_rho, t_one;
    fusion_terms(rho, sg, t_rho, t_one);
    pjsj = (float)((double)pjsj + t_rho);
    rsj = (float)((double)rsj + t_one);
}

// ---- the reference's "(double)x > 0.000001" / "< 0.000001" tests (PM.cc:345,497,510,560,662,705) ----------
// 0x358637bd (9.99999997e-7) is the largest float below the double 1e-6 and its successor lies above
// it, so the double comparison of a widened float is this float comparison (NaN: false both ways).
__device__ __forceinline__ bool gt_1em6(float x) { return x > __uint_as_float(0x358637bdu); }
__device__ __forceinline__ bool lt_1em6(float x) { return x <= __uint_as_float(0x358637bdu); }

// InverseDepthHypothesisFusion PM.cc:598-626 over a thread-private column hyp[i*stride], i < nh.
//
// The N^2 ChiTest divisions dominate a naive port.  Each decision "chi < 5.99" is first tried with
// reciprocals (v_rcp_f32, 1 ulp): chi~ = num*ra + num*rb differs from the reference's
// num/sa^2 + num/sb^2 by a relative 2^-20 at most, so outside the band 5.99*(1 +- 2^-14) the
// decision is already certain; only inside the band (or for zero / denormal / Inf / NaN sigmas,
// whose reciprocal is forced to NaN) is the exact chi_test evaluated.  Decisions are therefore
// bit-identical to the reference arithmetic.
__device__ __forceinline__ float safe_rcp_sq(float s)
{
    float s2 = s * s;
    bool ok = (s2 >= 1.0e-30f) && (s2 <= 1.0e30f);
    return ok ? __builtin_amdgcn_rcpf(s2) : __builtin_nanf("");
}

__device__ __forceinline__ bool chi_test_fast(float a, float b, float sa, float sb, float ra, float rb)
{
    float d = a - b;
    float num = d * d;
    float approx = num * ra + num * rb;
    if (approx < 5.9896f) return true;    // 5.99 * (1 - 2^-14)
    if (approx > 5.9904f) return false;   // 5.99 * (1 + 2^-14)
    return chi_test(a, b, sa, sb);        // band, or NaN from unsafe operands
}

// the same decision from {rho, 1/sigma^2} pairs; sigma of the second hypothesis is fetched (from LDS) only when
// the exact test is needed
__device__ __forceinline__ bool chi_test_lazy(float2 ha, float2 hb, float sa, const float* sb_ptr)
{
    float d = ha.x - hb.x;
    float num = d * d;
    float approx = num * ha.y + num * hb.y;
    if (approx < 5.9896f) return true;
    if (approx > 5.9904f) return false;
    return chi_test(ha.x, hb.x, sa, *sb_ptr);
}

__device__ __forceinline__ bool fuse_column(const float2* hyp, int stride, int nh, int lambdaN,
                                            float& rho_o, float& sigma_o)
{
    unsigned long long bestmask = 0;
    int best = 0;
    for (int a = 0; a < nh; a++) {
        float2 ha = hyp[a * stride];
        const float ra = safe_rcp_sq(ha.y);
        unsigned long long m = 0;
        for (int b = 0; b < nh; b++) {
            float2 hb = hyp[b * stride];
            if (chi_test_fast(ha.x, hb.x, ha.y, hb.y, ra, safe_rcp_sq(hb.y))) m |= 1ull << b;
        }
        int cnt = __popcll(m);
        if (cnt > best) {  // strict: first largest set wins, PM.cc:616
            best = cnt;
            bestmask = m;
        }
    }
    if (best < lambdaN) return false;  // PM.cc:623
    float pjsj = 0.f, rsj = 0.f;       // GetFusion overload B, PM.cc:947-970
    for (int b = 0; b < nh; b++) {
        if (!((bestmask >> b) & 1ull)) continue;
        float2 hb = hyp[b * stride];
        fusion_accum(hb.x, hb.y, pjsj, rsj);
    }
    rho_o = pjsj / rsj;
    sigma_o = sqrtf(1 / rsj);
    return true;
}

}  // namespace sdm
