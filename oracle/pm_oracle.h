/*
 * pm_oracle.h -- CPU ORACLE for the ProbabilityMapping hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a dependency-free C restatement of the arithmetic that
 * /root/reference/src/Modeler/ProbabilityMapping.cc ("PM.cc") specifies.  It exists to CHECK the
 * HIP product path; nothing under orb-slam-free-space-carving_amd/ or include/ may include, link
 * or call it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY UNPINNED: the reference holds no tests, fixtures or golden vectors for this path, its
 * source for the path is orphaned and ill-formed (not in CMakeLists.txt:76-108, header/source
 * mismatch) and OpenCV/Eigen are absent from the image, so the reference itself cannot be built
 * or run.  The oracle is therefore pinned only by (a) line-by-line transcription, each function
 * citing the PM.cc lines it follows, and (b) an independent float32 NumPy restatement of the
 * closed-form pieces (tests/test_oracle_crosscheck.py).  Normative choices where the reference
 * has undefined behaviour or un-pinnable third-party arithmetic are listed in DESIGN.md §3.
 *
 * Build: oracle/Makefile  (-O2 -ffp-contract=off -fno-fast-math; results are bit-defined).
 */
#ifndef PM_ORACLE_H
#define PM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Read-only view of what PM.cc reads from an ORB_SLAM2::KeyFrame. */
typedef struct {
    int W, H;
    const uint8_t *im;   /* kf->im_       CV_8UC1  H*W row-major            (PM.cc:202,433)   */
    const float *grad;   /* kf->GradImg   CV_32FC1 gradient magnitude       (PM.cc:201,411)   */
    const float *theta;  /* kf->GradTheta CV_32FC1 degrees [0,360)          (PM.cc:214,415)   */
    float I_stddev;      /* kf->I_stddev                                    (PM.cc:457)       */
    float fx, fy, cx, cy;/* include/KeyFrame.h:162                                            */
    float Tcw[12];       /* world->camera [R|t] row-major 3x4 (src/KeyFrame.cc:70-121)        */
} pmo_keyframe;

/* PM.h:64-70 depthHo without the never-written Pw field. `rho` is INVERSE depth. */
typedef struct {
    float rho;
    float sigma;
    int supported;
} pmo_hypo;

/* PM.h:38-49 macros as a runtime struct. */
typedef struct {
    float lambdaG;      /* 8   gradient gate            PM.h:40 */
    float lambdaL;      /* 80  epipolar-angle gate deg  PM.h:41 */
    float lambdaTheta;  /* 45  orientation gate deg     PM.h:42 */
    int lambdaN;        /* 3                            PM.h:43 */
    double theta_var;   /* THETA 0.23 (double literal)  PM.h:47 */
} pmo_params;

/* Per-(reference,neighbour) constants: PM.cc:859-860, 890-891 (R21,t21) and :972-986 (F12). */
typedef struct {
    float R21[9];
    float t21[3];
    float F12[9];
} pmo_pair;

/* Search statistics (for the "mean scanned candidates per search" figure). */
typedef struct {
    long long searches;    /* EpipolarSearch invocations                     */
    long long candidates;  /* uj values visited by the scan loop PM.cc:405   */
    long long gate_pass;   /* candidates that reached the cost PM.cc:433     */
    long long hypotheses;  /* accepted hypotheses PM.cc:216                  */
    long long fused;       /* pixels written at PM.cc:225                    */
} pmo_stats;

void pmo_default_params(pmo_params *p);

/* cv::fastAtan2 restated from OpenCV 3.x -- PM.cc:414. */
float pmo_fast_atan2(float y, float x);

/* Image ingest (src/Tracking.cc:244-257, 266-271; src/Modeler/Modeler.cc:154-155): undistort the interleaved
 * 1/3/4-channel frame (dist = {k1,k2,p1,p2,k3} or NULL) and convert to gray.  OpenCV absent: parity unpinned. */
void pmo_ingest(const uint8_t *src, int W, int H, int channels, int r_idx, int g_idx, int b_idx,
                const float K[4], const float *dist, uint8_t *gray);

/* Input pre-pass the reference omits: Scharr/32, magnitude, phase, sigma_I. */
void pmo_gradient_prepass(const uint8_t *im, int W, int H, float *grad, float *theta,
                          float *I_stddev);

void pmo_pair_geometry(const pmo_keyframe *kf1, const pmo_keyframe *kf2, pmo_pair *out);

/* PM.cc:370-383 */
void pmo_stereo_search_constraints(const float *orb_depths, int n, float *min_depth,
                                   float *max_depth);

/* PM.cc:467-484 + median at PM.cc:170-179.  mp1/mp2: map-point ids (<0 = none). */
float pmo_median_rot_in_plane(const int *mp1, const float *angle1, int n1, const int *mp2,
                              const float *angle2, int n2);

/* PM.cc:877-910 */
void pmo_search_range(const pmo_keyframe *kf1, const pmo_pair *pr, int px, int py, float mind,
                      float maxd, float *umin, float *umax);

/* PM.cc:845-875 */
float pmo_pixel_depth(const pmo_keyframe *kf1, const pmo_pair *pr, float uj, int px, int py);

/* PM.cc:385-465 (+806-829).  Returns hypothesis (supported=0 if none). */
void pmo_epipolar_search(const pmo_keyframe *kf1, const pmo_keyframe *kf2, const pmo_pair *pr,
                         int x, int y, float min_depth, float max_depth, float rot,
                         const pmo_params *prm, pmo_hypo *dh, float *best_u, float *best_v,
                         pmo_stats *st);

/* PM.cc:598-626 */
void pmo_fuse(const pmo_hypo *h, int n, const pmo_params *prm, pmo_hypo *dist);

/* PM.cc:197-231: hot loop 1 over one reference keyframe.  rho/sigma must be zeroed by caller
 * semantics (the function zero-fills them itself, matching a fresh depth_map_). */
void pmo_recon_search_fuse(const pmo_keyframe *ref, const pmo_keyframe *nbrs, const float *rot,
                           int n, float min_depth, float max_depth, const pmo_params *prm,
                           float *rho, float *sigma, pmo_stats *st);

/* PM.cc:486-547 / 549-596 (in place, Jacobi) */
void pmo_intra_check(float *rho, float *sigma, int W, int H);
void pmo_intra_grow(float *rho, float *sigma, const float *grad, int W, int H,
                    const pmo_params *prm);

/* PM.cc:137-256 = search+fuse, intra check, intra grow for one keyframe. */
void pmo_semi_dense_recon(const pmo_keyframe *ref, const pmo_keyframe *nbrs, const float *rot,
                          int n, float min_depth, float max_depth, const pmo_params *prm,
                          float *rho, float *sigma, pmo_stats *st);

/* PM.cc:628-799: updates cur_rho in place; neighbour maps are read-only. */
void pmo_inter_check(const pmo_keyframe *cur, float *cur_rho, const pmo_keyframe *nbrs,
                     const float *const *nbr_rho, const float *const *nbr_sigma, int n,
                     const pmo_params *prm);

/* PM.cc:337-367: xyz is H x 3W; only the 2-px-inset domain is written. */
void pmo_pointset(const pmo_keyframe *kf, const float *rho, float *xyz);

/* Keyframe-batched drivers for the timed CPU baseline: rows of all keyframes are shared out to the
 * OpenMP threads at once.  kfs is indexed by the entries of ref_idx / nbr_idx ([n_ref] / [n_ref][n]);
 * rho, sigma, rho_in, chk are [n_ref][H*W], xyz [n_ref][3*H*W] or NULL; map_rho/map_sigma are
 * indexed like kfs (the neighbours' finished maps, snapshot semantics). */
void pmo_recon_batch(const pmo_keyframe *kfs, const int *ref_idx, int n_ref, const int *nbr_idx, int n,
                     float min_depth, float max_depth, const pmo_params *prm, float *rho,
                     float *sigma, pmo_stats *st);
void pmo_inter_pointset_batch(const pmo_keyframe *kfs, const int *ref_idx, int n_ref, const int *nbr_idx,
                              int n, const float *const *map_rho, const float *const *map_sigma,
                              const pmo_params *prm, const float *rho_in, float *chk, float *xyz);

/* number of OpenMP threads the oracle was built for / will use (1 if built without -fopenmp) */
int pmo_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
