/*
 * pm_oracle.c -- CPU ORACLE for the ProbabilityMapping hot path.  TEST INFRASTRUCTURE ONLY.
 * See pm_oracle.h for the rules (who may call this) and the "PARITY UNPINNED" statement.
 *
 * Every function cites the lines of /root/reference/src/Modeler/ProbabilityMapping.cc ("PM.cc")
 * it follows.  Arithmetic follows the C++ promotion rules in force in PM.cc:
 * float unless a double literal / pow() promotes the expression.  Compile with
 * -ffp-contract=off so no mul-add is fused.
 *
 * Normative choices (the ONLY intentional deviations; all concern undefined behaviour or
 * arithmetic that lives in OpenCV, which is absent here):
 *  N1 3x3 / 3x1 products are float, accumulated left to right (cv::Mat gemm rounding not pinned).
 *  N2 K^-1 is the closed form of an upper-triangular K (cv::Mat::inv rounding not pinned).
 *  N3 scan candidate valid iff 0 <= uj <= W-1 and 1 <= yf < H-1 (reference reads column W / row H).
 *  N4 sub-pixel refine needs 1 <= u <= W-2 and floor(yf(u+-1)) in [0,H-2], else no hypothesis.
 *  N5 NaN epipolar line or NaN search-range end => no hypothesis (reference: UB int conversion).
 *  N6 bilinear is the 2-tap vertical lerp it degenerates to for integer x (bit-identical, App. A.1).
 *  N7 I_stddev is float; cv::fastAtan2 is the OpenCV-3.x polynomial restated from memory.
 *  N8 inter-KF check: NaN projection counts as out of bounds; K*X omits the exact-zero terms.
 */
#include "pm_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PMO_MAX_NBR 64

void pmo_default_params(pmo_params *p)
{
    p->lambdaG = 8.0f;      /* PM.h:40 */
    p->lambdaL = 80.0f;     /* PM.h:41 */
    p->lambdaTheta = 45.0f; /* PM.h:42 */
    p->lambdaN = 3;         /* PM.h:43 */
    p->theta_var = 0.23;    /* PM.h:47 */
}

int pmo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- cv::fastAtan2 (PM.cc:414), OpenCV 3.x mathfuncs_core atan_f32, degrees in [0,360). ---- */
float pmo_fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ---- image ingest: what the fork does to a camera frame before the path sees it ---------------------
 * src/Tracking.cc:266-271 cv::undistort(im, imu, mK, mDistCoef) on the colour frame, kept 3-channel by
 * Modeler::AddFrameImage (src/Modeler/Modeler.cc:1496-1514), converted with cvtColor(CV_RGB2GRAY) at its
 * use (src/Modeler/Modeler.cc:154-155); Tracking's own gray: src/Tracking.cc:244-257.  OpenCV is absent:
 * cv::undistort (initUndistortRectifyMap in double -> CV_16SC2 1/32-pixel map -> remap INTER_LINEAR in 15-bit
 * fixed point, BORDER_CONSTANT 0) and RGB2Gray<uchar> (4899/9617/1868 >> 14) restated from memory, with the
 * source position evaluated directly per pixel (OpenCV accumulates it along the row).  PARITY UNPINNED (N9). */
static int ingest_gray(int r, int g, int b) { return (r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14; }

void pmo_ingest(const uint8_t *src, int W, int H, int channels, int r_idx, int g_idx, int b_idx,
                const float K[4], const float *dist, uint8_t *gray)
{
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const int nc = channels == 1 ? 1 : 3;
    const int idx[3] = {channels == 1 ? 0 : r_idx, channels == 1 ? 0 : g_idx, channels == 1 ? 0 : b_idx};
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            const int i = v * W + u;
            if (!dist) {
                const uint8_t *px = src + (size_t)i * channels;
                gray[i] = (uint8_t)(channels == 1 ? px[0] : ingest_gray(px[r_idx], px[g_idx], px[b_idx]));
                continue;
            }
            const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
            const double x = ((double)u - cx) / fx, y = ((double)v - cy) / fy;
            const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2;
            const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
            const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
            const double us = fx * xd + cx, vs = fy * yd + cy;
            double fu = rint(us * 32.0), fv = rint(vs * 32.0); /* saturate_cast<int>: round half to even */
            if (!(fu > -2147483648.0)) fu = -2147483648.0;
            if (!(fv > -2147483648.0)) fv = -2147483648.0;
            if (fu > 2147483647.0) fu = 2147483647.0;
            if (fv > 2147483647.0) fv = 2147483647.0;
            const int iu = (int)fu, iv = (int)fv;
            const int sx = iu >> 5, sy = iv >> 5, a = iu & 31, b = iv & 31;
            const int wt[4] = {(32 - a) * (32 - b) * 32, a * (32 - b) * 32, (32 - a) * b * 32, a * b * 32};
            int acc[3] = {0, 0, 0};
            for (int t = 0; t < 4; t++) {
                const int yy = sy + (t >> 1), xx = sx + (t & 1);
                if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue; /* BORDER_CONSTANT 0 */
                const uint8_t *px = src + ((size_t)yy * W + xx) * channels;
                for (int c = 0; c < nc; c++) acc[c] += wt[t] * (int)px[idx[c]];
            }
            int val[3] = {0, 0, 0};
            for (int c = 0; c < nc; c++) val[c] = (acc[c] + (1 << 14)) >> 15;
            gray[i] = (uint8_t)(channels == 1 ? val[0] : ingest_gray(val[0], val[1], val[2]));
        }
}

/* ---- input pre-pass (not in the reference; SURVEY.md App. B/D defines it for the build) ---- */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void pmo_gradient_prepass(const uint8_t *im, int W, int H, float *grad, float *theta,
                          float *I_stddev)
{
    for (int y = 0; y < H; y++) {
        int ym = clampi(y - 1, 0, H - 1), yp = clampi(y + 1, 0, H - 1);
        for (int x = 0; x < W; x++) {
            int xm = clampi(x - 1, 0, W - 1), xp = clampi(x + 1, 0, W - 1);
            int a00 = im[ym * W + xm], a01 = im[ym * W + x], a02 = im[ym * W + xp];
            int a10 = im[y * W + xm], a12 = im[y * W + xp];
            int a20 = im[yp * W + xm], a21 = im[yp * W + x], a22 = im[yp * W + xp];
            int sx = 3 * (a02 - a00) + 10 * (a12 - a10) + 3 * (a22 - a20);
            int sy = 3 * (a20 - a00) + 10 * (a21 - a01) + 3 * (a22 - a02);
            float gx = (float)sx * (1.0f / 32.0f);
            float gy = (float)sy * (1.0f / 32.0f);
            float xx = gx * gx, yy = gy * gy;
            grad[y * W + x] = sqrtf(xx + yy);
            theta[y * W + x] = pmo_fast_atan2(gy, gx);
        }
    }
    /* population sigma from exact integer sums */
    long long s = 0, sq = 0;
    for (long long i = 0; i < (long long)W * H; i++) {
        s += im[i];
        sq += (long long)im[i] * im[i];
    }
    double n = (double)W * (double)H;
    double mean = (double)s / n;
    double var = (double)sq / n - mean * mean;
    if (var < 0) var = 0;
    *I_stddev = (float)sqrt(var);
}

/* ---- 3x3 float algebra, left-to-right accumulation (N1) ---- */
static void mat3_mul(const float *A, const float *B, float *C)
{
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) {
            float p0 = A[i * 3 + 0] * B[0 * 3 + k];
            float p1 = A[i * 3 + 1] * B[1 * 3 + k];
            float p2 = A[i * 3 + 2] * B[2 * 3 + k];
            C[i * 3 + k] = (p0 + p1) + p2;
        }
}
static void mat3_mul_bt(const float *A, const float *B, float *C) /* C = A * B^T */
{
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) {
            float p0 = A[i * 3 + 0] * B[k * 3 + 0];
            float p1 = A[i * 3 + 1] * B[k * 3 + 1];
            float p2 = A[i * 3 + 2] * B[k * 3 + 2];
            C[i * 3 + k] = (p0 + p1) + p2;
        }
}
static void mat3_vec(const float *A, const float *v, float *o)
{
    for (int i = 0; i < 3; i++) {
        float p0 = A[i * 3 + 0] * v[0];
        float p1 = A[i * 3 + 1] * v[1];
        float p2 = A[i * 3 + 2] * v[2];
        o[i] = (p0 + p1) + p2;
    }
}
static void kf_Rt(const pmo_keyframe *kf, float *R, float *t)
{
    for (int i = 0; i < 3; i++) {
        R[i * 3 + 0] = kf->Tcw[i * 4 + 0];
        R[i * 3 + 1] = kf->Tcw[i * 4 + 1];
        R[i * 3 + 2] = kf->Tcw[i * 4 + 2];
        t[i] = kf->Tcw[i * 4 + 3];
    }
}

/* R21,t21: PM.cc:859-860 == 890-891 == 643-644.  F12: PM.cc:972-986 with the skew matrix of
 * src/LocalMapping.cc:711-716 (PM's own GetSkewSymmetricMatrix is declared but never defined). */
void pmo_pair_geometry(const pmo_keyframe *kf1, const pmo_keyframe *kf2, pmo_pair *out)
{
    float R1[9], t1[3], R2[9], t2[3], tmp[3];
    kf_Rt(kf1, R1, t1);
    kf_Rt(kf2, R2, t2);

    /* R21 = Rcw2*Rcw1.t();  t21 = -Rcw2*Rcw1.t()*tcw1 + tcw2 */
    mat3_mul_bt(R2, R1, out->R21);
    mat3_vec(out->R21, t1, tmp);
    for (int i = 0; i < 3; i++) out->t21[i] = (-tmp[i]) + t2[i];

    /* R12 = R1w*R2w.t();  t12 = -R1w*R2w.t()*t2w + t1w */
    float R12[9], t12[3];
    mat3_mul_bt(R1, R2, R12);
    mat3_vec(R12, t2, tmp);
    for (int i = 0; i < 3; i++) t12[i] = (-tmp[i]) + t1[i];

    float t12x[9] = {0.f, -t12[2], t12[1], t12[2], 0.f, -t12[0], -t12[1], t12[0], 0.f};
    /* K1.t().inv() and K2.inv(), closed form (N2) */
    float K1ti[9] = {1.0f / kf1->fx, 0.f, 0.f, 0.f, 1.0f / kf1->fy, 0.f,
                     -kf1->cx / kf1->fx, -kf1->cy / kf1->fy, 1.f};
    float K2i[9] = {1.0f / kf2->fx, 0.f, -kf2->cx / kf2->fx, 0.f, 1.0f / kf2->fy,
                    -kf2->cy / kf2->fy, 0.f, 0.f, 1.f};
    float A[9], B[9];
    mat3_mul(K1ti, t12x, A);
    mat3_mul(A, R12, B);
    mat3_mul(B, K2i, out->F12);
}

/* ---- PM.cc:370-383 ---- */
void pmo_stereo_search_constraints(const float *d, int n, float *min_depth, float *max_depth)
{
    double acc = 0.0; /* std::accumulate(..., 0.0): double accumulator */
    for (int i = 0; i < n; i++) acc = acc + (double)d[i];
    float sum = (float)acc;
    float mean = sum / (float)n;
    double acc2 = 0.0; /* std::inner_product(..., 0.0) over float diffs */
    for (int i = 0; i < n; i++) {
        float diff = d[i] - mean;
        float pr = diff * diff;
        acc2 = acc2 + (double)pr;
    }
    float variance = (float)(acc2 / (double)n);
    float stdev = sqrtf(variance);
    *max_depth = 1.0f / (mean + 2.0f * stdev);
    *min_depth = 1.0f / (mean - 2.0f * stdev);
}

/* ---- PM.cc:467-484 and the median of PM.cc:170-179 ---- */
static int cmp_float(const void *a, const void *b)
{
    float fa = *(const float *)a, fb = *(const float *)b;
    return (fa > fb) - (fa < fb);
}
float pmo_median_rot_in_plane(const int *mp1, const float *angle1, int n1, const int *mp2,
                              const float *angle2, int n2)
{
    float *rot = (float *)malloc(sizeof(float) * (size_t)(n1 > 0 ? n1 : 1) * (size_t)(n2 > 0 ? n2 : 1));
    int cnt = 0;
    for (int i = 0; i < n1; i++) {
        if (mp1[i] < 0) continue; /* if (vMPs1[idx1]) */
        for (int j = 0; j < n2; j++) {
            if (mp2[j] == mp1[i]) {
                float a1 = angle1[i], a2 = angle2[j];
                if (a1 < 0 || a2 < 0) continue;
                rot[cnt++] = a2 - a1;
            }
        }
    }
    float med = 0.f;
    if (cnt > 0) {
        qsort(rot, (size_t)cnt, sizeof(float), cmp_float);
        med = rot[(cnt - 1) / 2];
    }
    free(rot);
    return med;
}

/* per-pixel ray of the reference keyframe: PM.cc:862 == 893 == 677 */
static inline void pixel_ray(const pmo_keyframe *kf, int px, int py, float *xp0, float *xp1)
{
    *xp0 = ((float)px - kf->cx) / kf->fx;
    *xp1 = ((float)py - kf->cy) / kf->fy;
}
static inline float row_dot_xp(const float *r, float xp0, float xp1)
{
    /* R21.row(i) * xp with xp = (xp0, xp1, 1) */
    float p0 = r[0] * xp0;
    float p1 = r[1] * xp1;
    float p2 = r[2] * 1.0f;
    return (p0 + p1) + p2;
}

/* ---- PM.cc:877-910 ---- */
void pmo_search_range(const pmo_keyframe *kf1, const pmo_pair *pr, int px, int py, float mind,
                      float maxd, float *umin_o, float *umax_o)
{
    float fx = kf1->fx, cx = kf1->cx;
    float xp0, xp1;
    pixel_ray(kf1, px, py, &xp0, &xp1);
    float rx = row_dot_xp(pr->R21 + 0, xp0, xp1);
    float rz = row_dot_xp(pr->R21 + 6, xp0, xp1);
    /* xp2 = R21*xp1*d + t21 (rows 0 and 2 only are used) */
    float x_min = rx * mind + pr->t21[0], z_min = rz * mind + pr->t21[2];
    float x_max = rx * maxd + pr->t21[0], z_max = rz * maxd + pr->t21[2];
    float umin = fx * x_min / z_min + cx;
    float umax = fx * x_max / z_max + cx;
    if (umin > umax) {
        float t = umax;
        umax = umin;
        umin = t;
    }
    float cols = (float)kf1->W; /* kf->im_.cols of the REFERENCE keyframe, PM.cc:908 */
    if (umin < 0) umin = 0;
    if (umax < 0) umax = 0;
    if (umin > cols) umin = cols;
    if (umax > cols) umax = cols;
    *umin_o = umin;
    *umax_o = umax;
}

/* ---- PM.cc:845-875 (Eq. 8) ---- */
float pmo_pixel_depth(const pmo_keyframe *kf1, const pmo_pair *pr, float uj, int px, int py)
{
    float fx = kf1->fx, cx = kf1->cx;
    float ucx = uj - cx;
    float xp0, xp1;
    pixel_ray(kf1, px, py, &xp0, &xp1);
    float num1 = row_dot_xp(pr->R21 + 6, xp0, xp1) * ucx;
    float num2 = fx * row_dot_xp(pr->R21 + 0, xp0, xp1);
    float denom1 = -pr->t21[2] * ucx;
    float denom2 = fx * pr->t21[0];
    return (num1 - num2) / (denom1 + denom2);
}

/* bilinear<T> (PM.cc:40-59) at integer x: 2-tap vertical lerp (N6). y0 must be in [0,H-2]. */
static inline float lerp_u8(const uint8_t *img, int W, int y0, int x, float yf)
{
    float y0w = (float)(y0 + 1) - yf;
    float y1w = 1.0f - y0w;
    float v0 = (float)(int)img[y0 * W + x], v1 = (float)(int)img[(y0 + 1) * W + x];
    return v0 * y0w + v1 * y1w;
}
static inline float lerp_f32(const float *img, int W, int y0, int x, float yf)
{
    float y0w = (float)(y0 + 1) - yf;
    float y1w = 1.0f - y0w;
    return img[y0 * W + x] * y0w + img[(y0 + 1) * W + x] * y1w;
}

/* ---- PM.cc:385-465 EpipolarSearch, with ComputeInvDepthHypothesis PM.cc:806-829 inlined ---- */
void pmo_epipolar_search(const pmo_keyframe *kf1, const pmo_keyframe *kf2, const pmo_pair *pr,
                         int x, int y, float min_depth, float max_depth, float rot,
                         const pmo_params *prm, pmo_hypo *dh, float *best_u_o, float *best_v_o,
                         pmo_stats *st)
{
    const float *F = pr->F12;
    const int W1 = kf1->W;
    const int W2 = kf2->W, H2 = kf2->H;
    dh->rho = 0.f;
    dh->sigma = 0.f;
    dh->supported = 0;
    if (best_u_o) *best_u_o = 0.f;
    if (best_v_o) *best_v_o = 0.f;
    if (st) st->searches++;

    float pixel = (float)kf1->im[y * W1 + x];  /* PM.cc:202 */
    float grad1 = kf1->grad[y * W1 + x];       /* PM.cc:434 */
    float th_pi = kf1->theta[y * W1 + x];      /* PM.cc:214 */

    /* PM.cc:389-391 */
    float a = (float)x * F[0] + (float)y * F[3] + F[6];
    float b = (float)x * F[1] + (float)y * F[4] + F[7];
    float c = (float)x * F[2] + (float)y * F[5] + F[8];

    float ab = a / b;
    if (ab < -4 || ab > 4) return;  /* PM.cc:393 */
    if (ab != ab) return;           /* N5 */
    float cb = c / b;

    float old_err = 1000000.0f;
    float best_pe = 0.f, best_ge = 0.f;
    int best_pixel = 0;

    float umin, umax;
    pmo_search_range(kf1, pr, x, y, min_depth, max_depth, &umin, &umax); /* PM.cc:404 */
    if (umin != umin || umax != umax) return; /* N5 */

    /* loop-invariant in the reference's loop body (PM.cc:414,424-426) */
    float th_line = pmo_fast_atan2(-a / b, 1.0f);
    float ang_pi_rot = th_pi + rot;
    if (ang_pi_rot >= 360) ang_pi_rot -= 360;
    if (ang_pi_rot < 0) ang_pi_rot += 360;

    int lo = (int)ceilf(umin);
    int hi = (int)floorf(umax);
    if (hi > W2 - 1) hi = W2 - 1; /* N3 */
    for (int uj = lo; uj <= hi; uj++) { /* PM.cc:405 */
        if (st) st->candidates++;
        float yf = -(ab * (float)uj + cb); /* PM.cc:407,433 */
        if (!(yf >= 1.0f && yf < (float)(H2 - 1))) continue; /* PM.cc:408 + N3 */
        int vj = (int)yf;

        /* condition 1: PM.cc:411 */
        if (kf2->grad[vj * W2 + uj] < prm->lambdaG) continue;

        /* condition 2: PM.cc:414-421 */
        float th2 = kf2->theta[vj * W2 + uj];
        float ang_diff = th2 - th_line;
        if (ang_diff >= 360) ang_diff -= 360;
        if (ang_diff < 0) ang_diff += 360;
        if (ang_diff > 180) ang_diff = 360 - ang_diff;
        if (ang_diff > 90) ang_diff = 180 - ang_diff;
        if (ang_diff > prm->lambdaL) continue;

        /* condition 3: PM.cc:424-431 */
        float th_diff = th2 - ang_pi_rot;
        if (th_diff >= 360) th_diff -= 360;
        if (th_diff < 0) th_diff += 360;
        if (th_diff > 180) th_diff = 360 - th_diff;
        if (th_diff > prm->lambdaTheta) continue;

        if (st) st->gate_pass++;
        /* PM.cc:433-436 */
        float pe = pixel - lerp_u8(kf2->im, W2, vj, uj, yf);
        float ge = grad1 - lerp_f32(kf2->grad, W2, vj, uj, yf);
        float pe2 = pe * pe, ge2 = ge * ge;
        float err = (float)((double)pe2 + (double)ge2 / prm->theta_var);
        if (err < old_err) { /* PM.cc:437: strict, lowest uj wins ties */
            best_pixel = uj;
            old_err = err;
            best_pe = pe;
            best_ge = ge;
        }
    }

    if (!(old_err < 1000000.0f)) return; /* PM.cc:446 */

    /* PM.cc:449-457 sub-pixel refinement */
    int up = best_pixel + 1, um = best_pixel - 1;
    if (um < 0 || up > W2 - 1) return; /* N4 */
    float yfp = -(ab * (float)up + cb);
    float yfm = -(ab * (float)um + cb);
    float fyp = floorf(yfp), fym = floorf(yfm);
    if (!(fyp >= 0.0f && fyp <= (float)(H2 - 2))) return; /* N4 */
    if (!(fym >= 0.0f && fym <= (float)(H2 - 2))) return;
    int y0p = (int)fyp, y0m = (int)fym;

    float g = (lerp_u8(kf2->im, W2, y0p, up, yfp) - lerp_u8(kf2->im, W2, y0m, um, yfm)) / 2;
    float q = (lerp_f32(kf2->grad, W2, y0p, up, yfp) - lerp_f32(kf2->grad, W2, y0m, um, yfm)) / 2;

    const double inv_theta = 1 / prm->theta_var; /* (1/THETA), double */
    float gg = g * g;
    float denom = (float)((double)gg + inv_theta * (double)q * (double)q); /* PM.cc:455 */
    float gpe = g * best_pe;
    float ustar = (float)((double)best_pixel +
                          ((double)gpe + inv_theta * (double)q * (double)best_ge) / (double)denom);
    float ustar_var = 2 * kf2->I_stddev * kf2->I_stddev / denom; /* PM.cc:457 */

    if (best_u_o) *best_u_o = ustar;
    if (best_v_o) *best_v_o = -(ab * ustar + cb); /* PM.cc:460 */

    /* ComputeInvDepthHypothesis PM.cc:806-829 */
    float d0 = pmo_pixel_depth(kf1, pr, ustar, x, y);
    float s = sqrtf(ustar_var);
    float dmin = pmo_pixel_depth(kf1, pr, ustar - s, x, y);
    float dmax = pmo_pixel_depth(kf1, pr, ustar + s, x, y);
    float e1 = fabsf(dmax - d0), e2 = fabsf(dmin - d0);
    float sig = (e1 < e2) ? e2 : e1; /* cv::max(a,b) == std::max: (a<b)?b:a */
    dh->rho = d0;
    dh->sigma = sig;
    dh->supported = 1;
}

/* ---- PM.cc:912-918 / 920-924 (identical arithmetic) ---- */
static inline int chi_test(float a, float b, float sa, float sb)
{
    float num = (a - b) * (a - b);
    float chi = num / (sa * sa) + num / (sb * sb);
    return (double)chi < 5.99;
}

/* GetFusion overload B, PM.cc:947-970 */
static void get_fusion_b(const float *rho, const float *sig, int n, float *rho_o, float *sig_o,
                         float *min_sigma_o)
{
    float tmin = sig[0];
    float pjsj = 0, rsj = 0;
    for (int j = 0; j < n; j++) {
        double s2 = (double)sig[j] * (double)sig[j]; /* pow(sigma,2): exact in double */
        pjsj = (float)((double)pjsj + (double)rho[j] / s2);
        rsj = (float)((double)rsj + 1.0 / s2);
        double t2 = (double)tmin * (double)tmin;
        if (s2 < t2) tmin = sig[j];
    }
    *rho_o = pjsj / rsj;
    *sig_o = sqrtf(1 / rsj);
    *min_sigma_o = tmin;
}

/* GetFusion overload A, PM.cc:926-945 */
static void get_fusion_a(const float *rho, const float *sig, int n, float *rho_o, float *sig_o)
{
    float pjsj = 0, rsj = 0;
    float min_sigma = sig[0];
    for (int i = 0; i < n; i++) {
        double s2 = (double)sig[i] * (double)sig[i];
        pjsj = (float)((double)pjsj + (double)rho[i] / s2);
        rsj = (float)((double)rsj + 1.0 / s2);
        if (sig[i] < min_sigma) min_sigma = sig[i];
    }
    *rho_o = pjsj / rsj;
    *sig_o = min_sigma;
}

/* ---- PM.cc:598-626 ---- */
void pmo_fuse(const pmo_hypo *h, int n, const pmo_params *prm, pmo_hypo *dist)
{
    dist->rho = 0;
    dist->sigma = 0;
    dist->supported = 0;
    if (n > PMO_MAX_NBR) n = PMO_MAX_NBR;
    float br[PMO_MAX_NBR], bs[PMO_MAX_NBR], tr[PMO_MAX_NBR], ts[PMO_MAX_NBR];
    int nb = 0;
    for (int a = 0; a < n; a++) {
        int nt = 0;
        for (int b = 0; b < n; b++) {
            if (chi_test(h[a].rho, h[b].rho, h[a].sigma, h[b].sigma)) {
                tr[nt] = h[b].rho;
                ts[nt] = h[b].sigma;
                nt++;
            }
        }
        if (nt > nb) { /* strict: first largest set wins */
            nb = nt;
            memcpy(br, tr, sizeof(float) * (size_t)nt);
            memcpy(bs, ts, sizeof(float) * (size_t)nt);
        }
    }
    if (nb >= prm->lambdaN) {
        float ms;
        get_fusion_b(br, bs, nb, &dist->rho, &dist->sigma, &ms);
        dist->supported = 1;
    }
}

/* ---- PM.cc:197-231 hot loop 1 ---- */
/* one image row of the loop (the unit the reference's `#pragma omp parallel for ... collapse(2)` at
 * PM.cc:197 would hand out) */
static void search_fuse_row(const pmo_keyframe *ref, const pmo_keyframe *nbrs, const pmo_pair *pairs,
                            const float *rot, int n, float min_depth, float max_depth,
                            const pmo_params *prm, int y, float *rho, float *sigma, pmo_stats *st)
{
    const int W = ref->W;
    for (int x = 2; x < W - 2; x++) {
        if (ref->grad[y * W + x] < prm->lambdaG) continue; /* PM.cc:201 */
        pmo_hypo ho[PMO_MAX_NBR];
        int nh = 0;
        for (int j = 0; j < n; j++) {
            pmo_hypo dh;
            pmo_epipolar_search(ref, &nbrs[j], &pairs[j], x, y, min_depth, max_depth, rot ? rot[j] : 0.0f,
                                prm, &dh, 0, 0, st);
            if (dh.supported && (double)(1 / dh.rho) > 0.0) ho[nh++] = dh; /* PM.cc:216 */
        }
        st->hypotheses += nh;
        if (nh > prm->lambdaN) { /* PM.cc:221 */
            pmo_hypo f;
            pmo_fuse(ho, nh, prm, &f);
            if (f.supported) {
                rho[y * W + x] = f.rho;
                sigma[y * W + x] = f.sigma;
                st->fused++;
            }
        }
    }
}

void pmo_recon_search_fuse(const pmo_keyframe *ref, const pmo_keyframe *nbrs, const float *rot,
                           int n, float min_depth, float max_depth, const pmo_params *prm,
                           float *rho, float *sigma, pmo_stats *st)
{
    const int W = ref->W, H = ref->H;
    if (n > PMO_MAX_NBR) n = PMO_MAX_NBR;
    pmo_pair *pairs = (pmo_pair *)malloc(sizeof(pmo_pair) * (size_t)(n > 0 ? n : 1));
    for (int j = 0; j < n; j++) pmo_pair_geometry(ref, &nbrs[j], &pairs[j]); /* PM.cc:189-195 */
    memset(rho, 0, sizeof(float) * (size_t)W * H);
    memset(sigma, 0, sizeof(float) * (size_t)W * H);

    long long s_search = 0, s_cand = 0, s_gate = 0, s_hyp = 0, s_fused = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : s_search, s_cand, s_gate, s_hyp, s_fused)
#endif
    for (int y = 2; y < H - 2; y++) {
        pmo_stats ls = {0, 0, 0, 0, 0};
        search_fuse_row(ref, nbrs, pairs, rot, n, min_depth, max_depth, prm, y, rho, sigma, &ls);
        s_search += ls.searches;
        s_cand += ls.candidates;
        s_gate += ls.gate_pass;
        s_hyp += ls.hypotheses;
        s_fused += ls.fused;
    }
    if (st) {
        st->searches += s_search;
        st->candidates += s_cand;
        st->gate_pass += s_gate;
        st->hypotheses += s_hyp;
        st->fused += s_fused;
    }
    free(pairs);
}

/* ---- PM.cc:486-547 ---- */
void pmo_intra_check(float *rho, float *sigma, int W, int H)
{
    size_t bytes = sizeof(float) * (size_t)W * H;
    float *rn = (float *)malloc(bytes), *sn = (float *)malloc(bytes);
    memcpy(rn, rho, bytes); /* clone, PM.cc:488-489 */
    memcpy(sn, sigma, bytes);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int py = 2; py < H - 2; py++) {
        for (int px = 2; px < W - 2; px++) {
            float dp = rho[py * W + px];
            if (!((double)dp > 0.000001)) continue; /* PM.cc:497 */
            float sp = sigma[py * W + px];
            float cr[9], cs[9];
            int nc = 0;
            for (int y = py - 1; y <= py + 1; y++)
                for (int x = px - 1; x <= px + 1; x++) {
                    if (x == px && y == py) continue;
                    float dn = rho[y * W + x];
                    if ((double)dn > 0.000001) {
                        float sg = sigma[y * W + x];
                        if (chi_test(dn, dp, sg, sp)) { /* PM.cc:512 */
                            cr[nc] = dn;
                            cs[nc] = sg;
                            nc++;
                        }
                    }
                }
            cr[nc] = dp; /* itself, last: PM.cc:522 */
            cs[nc] = sp;
            nc++;
            if (nc >= 3) {
                float fr, fs, ms;
                get_fusion_b(cr, cs, nc, &fr, &fs, &ms);
                rn[py * W + px] = fr; /* PM.cc:530-531: sigma := MIN sigma, not the fused one */
                sn[py * W + px] = ms;
            } else {
                rn[py * W + px] = 0.0f;
                sn[py * W + px] = 0.0f;
            }
        }
    }
    memcpy(rho, rn, bytes);
    memcpy(sigma, sn, bytes);
    free(rn);
    free(sn);
}

/* ---- PM.cc:549-596 ---- */
void pmo_intra_grow(float *rho, float *sigma, const float *grad, int W, int H,
                    const pmo_params *prm)
{
    size_t bytes = sizeof(float) * (size_t)W * H;
    float *rn = (float *)malloc(bytes), *sn = (float *)malloc(bytes);
    memcpy(rn, rho, bytes);
    memcpy(sn, sigma, bytes);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int py = 2; py < H - 2; py++) {
        for (int px = 2; px < W - 2; px++) {
            float dp = rho[py * W + px];
            if (!((double)dp < 0.000001)) continue; /* PM.cc:560 */
            if (grad[py * W + px] < prm->lambdaG) continue; /* PM.cc:562 */
            float sp = sigma[py * W + px];
            float cr[8], cs[8];
            int nc = 0;
            for (int y = py - 1; y <= py + 1; y++)
                for (int x = px - 1; x <= px + 1; x++) {
                    if (x == px && y == py) continue;
                    float dn = rho[y * W + x], sg = sigma[y * W + x];
                    if (chi_test(dn, dp, sg, sp)) { /* PM.cc:571 */
                        cr[nc] = dn;
                        cs[nc] = sg;
                        nc++;
                    }
                }
            if (nc >= 2) { /* PM.cc:581 */
                float d, s;
                get_fusion_a(cr, cs, nc, &d, &s);
                rn[py * W + px] = d;
                sn[py * W + px] = s;
            }
        }
    }
    memcpy(rho, rn, bytes);
    memcpy(sigma, sn, bytes);
    free(rn);
    free(sn);
}

/* ---- PM.cc:137-256 per-keyframe driver ---- */
void pmo_semi_dense_recon(const pmo_keyframe *ref, const pmo_keyframe *nbrs, const float *rot,
                          int n, float min_depth, float max_depth, const pmo_params *prm,
                          float *rho, float *sigma, pmo_stats *st)
{
    pmo_recon_search_fuse(ref, nbrs, rot, n, min_depth, max_depth, prm, rho, sigma, st);
    pmo_intra_check(rho, sigma, ref->W, ref->H);          /* PM.cc:237 */
    pmo_intra_grow(rho, sigma, ref->grad, ref->W, ref->H, prm); /* PM.cc:238 */
}

/* ---- PM.cc:628-799 ---- */
/* one image row of the loop at PM.cc:659-660 */
static void inter_check_row(const pmo_keyframe *cur, float *cur_rho, const pmo_keyframe *nbrs,
                            const pmo_pair *pairs, const float *const *nbr_rho,
                            const float *const *nbr_sigma, int n, const pmo_params *prm, int py)
{
    const int cols = cur->W, rows = cur->H;
    const float fx = cur->fx, fy = cur->fy, cx = cur->cx, cy = cur->cy;
    for (int px = 2; px < cols - 2; px++) {
        float depthp = cur_rho[py * cols + px];
        if ((double)depthp < 0.000001) continue; /* PM.cc:662 */
        int kf_count = 0;
        /* Gauss-Newton sums over compatible (j,n) in order, PM.cc:771-791 */
        float sum_Jr = 0.f, sum_JJ = 0.f;
        float xp0 = ((float)px - cx) / fx, xp1 = ((float)py - cy) / fy; /* PM.cc:677 */
        float dp = 1 / depthp;                                          /* PM.cc:769 */
        for (int j = 0; j < n; j++) {
            const pmo_keyframe *kj = &nbrs[j];
            const float *R = pairs[j].R21, *t = pairs[j].t21;
            /* temp = Rji*xp/depthp + tji ; Xj = K*temp ; Xj /= Xj(2)   PM.cc:678-680 */
            float t0 = row_dot_xp(R + 0, xp0, xp1) / depthp + t[0];
            float t1 = row_dot_xp(R + 3, xp0, xp1) / depthp + t[1];
            float rzxp = row_dot_xp(R + 6, xp0, xp1);
            float t2 = rzxp / depthp + t[2];
            float u = kj->fx * t0 + kj->cx * t2; /* N8 */
            float v = kj->fy * t1 + kj->cy * t2;
            float xj = u / t2, yj = v / t2;
            /* Eq.12  PM.cc:684-688 */
            float denom2 = depthp * t[2];
            float depthj = depthp / (rzxp + denom2);
            if (!(xj >= 0 && xj < (float)(cols - 1) && yj >= 0 && yj < (float)(rows - 1)))
                continue; /* PM.cc:695 + N8 */
            int x0 = (int)floorf(xj), y0 = (int)floorf(yj);
            int x1 = x0 + 1, y1 = y0 + 1;
            const int tx[4] = {x0, x0, x1, x1}; /* order PM.cc:705,717,729,741 */
            const int ty[4] = {y0, y1, y0, y1};
            int nj = 0;
            for (int k = 0; k < 4; k++) {
                float d = nbr_rho[j][ty[k] * kj->W + tx[k]];
                float sg = nbr_sigma[j][ty[k] * kj->W + tx[k]];
                if ((double)d > 0.000001) {
                    float dd = depthj - d;
                    float test = (float)(((double)dd * (double)dd) / ((double)sg * (double)sg));
                    if ((double)test < 3.84) {
                        nj++;
                        /* PM.cc:777-783 */
                        float djn = 1 / d;
                        float d2sigma = djn * djn * sg;
                        float J = -rzxp / d2sigma;
                        float r0 = (djn - dp * rzxp - t[2]) / d2sigma;
                        sum_Jr = sum_Jr + J * r0;
                        sum_JJ = sum_JJ + J * J;
                    }
                }
            }
            if (nj >= 1) kf_count++;
        }
        if (kf_count < prm->lambdaN) {
            cur_rho[py * cols + px] = 0.0f; /* PM.cc:764: sigma untouched */
        } else {
            float dpDelta = (-sum_Jr) / sum_JJ;       /* PM.cc:788-791 */
            cur_rho[py * cols + px] = 1 / (dp + dpDelta); /* PM.cc:793 */
        }
    }
}

void pmo_inter_check(const pmo_keyframe *cur, float *cur_rho, const pmo_keyframe *nbrs,
                     const float *const *nbr_rho, const float *const *nbr_sigma, int n,
                     const pmo_params *prm)
{
    if (n > PMO_MAX_NBR) n = PMO_MAX_NBR;
    const int rows = cur->H;
    pmo_pair *pairs = (pmo_pair *)malloc(sizeof(pmo_pair) * (size_t)(n > 0 ? n : 1));
    for (int j = 0; j < n; j++) pmo_pair_geometry(cur, &nbrs[j], &pairs[j]); /* PM.cc:634-649 */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int py = 2; py < rows - 2; py++)
        inter_check_row(cur, cur_rho, nbrs, pairs, nbr_rho, nbr_sigma, n, prm, py);
    free(pairs);
}

/* ---- PM.cc:337-367 ---- */
void pmo_pointset(const pmo_keyframe *kf, const float *rho, float *xyz)
{
    const int W = kf->W, H = kf->H;
    /* Twc = [Rcw^T | -Rcw^T*tcw], src/KeyFrame.cc:70-84 */
    float R[9], t[3], Rwc[9], Ow[3];
    kf_Rt(kf, R, t);
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) Rwc[i * 3 + k] = R[k * 3 + i];
    mat3_vec(Rwc, t, Ow);
    for (int i = 0; i < 3; i++) Ow[i] = -Ow[i];
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int y = 2; y < H - 2; y++) {
        for (int x = 2; x < W - 2; x++) {
            float inv_d = rho[y * W + x];
            float *o = xyz + (size_t)y * 3 * W + 3 * x;
            if ((double)inv_d < 0.000001) { /* PM.cc:345 */
                o[0] = o[1] = o[2] = 0.0f;
                continue;
            }
            float Z = 1 / inv_d;
            float X = Z * ((float)x - kf->cx) / kf->fx;
            float Y = Z * ((float)y - kf->cy) / kf->fy;
            for (int i = 0; i < 3; i++) /* pos = Twc * (X,Y,Z,1) */
                o[i] = ((Rwc[i * 3 + 0] * X + Rwc[i * 3 + 1] * Y) + Rwc[i * 3 + 2] * Z) + Ow[i] * 1.0f;
        }
    }
}

/* ---- keyframe-batched drivers for the TIMED CPU baseline (bench.py cpu_baseline) -----------------
 * The reference's driver visits keyframes one after another (PM.cc:262-315) with its OpenMP pragmas
 * over image rows (PM.cc:197,491,554,658) -- which are inert in its build.  These two entry points run
 * the same per-keyframe functions as above over a whole batch, with the rows of ALL keyframes handed
 * out to the threads at once (collapse over (keyframe,row)), so that many cores stay busy and no
 * per-keyframe call goes through Python.  Results are identical to calling the per-keyframe
 * functions in a loop (the units are independent: snapshot semantics). */
void pmo_recon_batch(const pmo_keyframe *kfs, const int *ref_idx, int n_ref, const int *nbr_idx, int n,
                     float min_depth, float max_depth, const pmo_params *prm, float *rho,
                     float *sigma, pmo_stats *st)
{
    if (n_ref <= 0) return;
    if (n > PMO_MAX_NBR) n = PMO_MAX_NBR;
    const int W = kfs[ref_idx[0]].W, H = kfs[ref_idx[0]].H;
    const size_t P = (size_t)W * H;
    pmo_pair *pairs = (pmo_pair *)malloc(sizeof(pmo_pair) * (size_t)n_ref * (size_t)(n > 0 ? n : 1));
    pmo_keyframe *nb = (pmo_keyframe *)malloc(sizeof(pmo_keyframe) * (size_t)n_ref * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n_ref; i++)
        for (int j = 0; j < n; j++) {
            nb[(size_t)i * n + j] = kfs[nbr_idx[(size_t)i * n + j]];
            pmo_pair_geometry(&kfs[ref_idx[i]], &nb[(size_t)i * n + j], &pairs[(size_t)i * n + j]);
        }
    memset(rho, 0, sizeof(float) * P * (size_t)n_ref);
    memset(sigma, 0, sizeof(float) * P * (size_t)n_ref);
    long long s_search = 0, s_cand = 0, s_gate = 0, s_hyp = 0, s_fused = 0;
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(dynamic, 4) reduction(+ : s_search, s_cand, s_gate, s_hyp, s_fused)
#endif
    for (int i = 0; i < n_ref; i++)
        for (int y = 2; y < H - 2; y++) {
            pmo_stats ls = {0, 0, 0, 0, 0};
            search_fuse_row(&kfs[ref_idx[i]], nb + (size_t)i * n, pairs + (size_t)i * n, 0, n, min_depth,
                            max_depth, prm, y, rho + P * i, sigma + P * i, &ls);
            s_search += ls.searches;
            s_cand += ls.candidates;
            s_gate += ls.gate_pass;
            s_hyp += ls.hypotheses;
            s_fused += ls.fused;
        }
    /* PM.cc:237-238 per keyframe (the row loops inside run on one thread each: nested teams are off) */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int i = 0; i < n_ref; i++) {
        pmo_intra_check(rho + P * i, sigma + P * i, W, H);
        pmo_intra_grow(rho + P * i, sigma + P * i, kfs[ref_idx[i]].grad, W, H, prm);
    }
    if (st) {
        st->searches += s_search;
        st->candidates += s_cand;
        st->gate_pass += s_gate;
        st->hypotheses += s_hyp;
        st->fused += s_fused;
    }
    free(pairs);
    free(nb);
}

/* PM.cc:300-306 for a batch: chk[i] = rho_in[i] checked against the neighbours' maps (map_rho /
 * map_sigma are indexed like kfs), then the point set of the checked map (xyz may be NULL). */
void pmo_inter_pointset_batch(const pmo_keyframe *kfs, const int *ref_idx, int n_ref, const int *nbr_idx,
                              int n, const float *const *map_rho, const float *const *map_sigma,
                              const pmo_params *prm, const float *rho_in, float *chk, float *xyz)
{
    if (n_ref <= 0) return;
    if (n > PMO_MAX_NBR) n = PMO_MAX_NBR;
    const int W = kfs[ref_idx[0]].W, H = kfs[ref_idx[0]].H;
    const size_t P = (size_t)W * H, nn = (size_t)(n > 0 ? n : 1);
    pmo_pair *pairs = (pmo_pair *)malloc(sizeof(pmo_pair) * (size_t)n_ref * nn);
    pmo_keyframe *nb = (pmo_keyframe *)malloc(sizeof(pmo_keyframe) * (size_t)n_ref * nn);
    const float **nr = (const float **)malloc(sizeof(float *) * (size_t)n_ref * nn);
    const float **ns = (const float **)malloc(sizeof(float *) * (size_t)n_ref * nn);
    for (int i = 0; i < n_ref; i++)
        for (int j = 0; j < n; j++) {
            const int k = nbr_idx[(size_t)i * n + j];
            nb[(size_t)i * n + j] = kfs[k];
            nr[(size_t)i * n + j] = map_rho[k];
            ns[(size_t)i * n + j] = map_sigma[k];
            pmo_pair_geometry(&kfs[ref_idx[i]], &kfs[k], &pairs[(size_t)i * n + j]);
        }
    memcpy(chk, rho_in, sizeof(float) * P * (size_t)n_ref);
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
#endif
    for (int i = 0; i < n_ref; i++)
        for (int py = 2; py < H - 2; py++)
            inter_check_row(&kfs[ref_idx[i]], chk + P * i, nb + (size_t)i * n, pairs + (size_t)i * n,
                            nr + (size_t)i * n, ns + (size_t)i * n, n, prm, py);
    if (xyz) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
        for (int i = 0; i < n_ref; i++) pmo_pointset(&kfs[ref_idx[i]], chk + P * i, xyz + 3 * P * i);
    }
    free(pairs);
    free(nb);
    free(nr);
    free(ns);
}
