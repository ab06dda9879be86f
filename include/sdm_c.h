/*
 * sdm_c.h -- C ABI of the MI355X semi-dense mapping engine (libsdm_hip.so).
 *
 * Drop-in boundary for the ProbabilityMapping hot path of atlas-jj/ORB-SLAM-free-space-carving.
 * The reference has no FFI/plugin registry for this path -- it is a plain C++ class
 * (/root/reference/include/Modeler/ProbabilityMapping.h:61-105, "PM.h").  Each entry point below
 * names the reference method it replaces; include/sdm/ProbabilityMapping.h keeps the reference's
 * class surface and forwards here.  Plain pointers and sizes only; no torch/HIP types.
 *
 * Conventions
 *  - "rho" is INVERSE depth (PM.cc:352-353: Z = 1/inv_d); maps are row-major, stride W.
 *  - keyframes live in numbered device slots [0, max_keyframes); all share W x H.
 *  - every call returns 0 on success or an SDM_E* code; sdm_last_error() has the text.  The
 *    reference's methods are void and unchecked (PM.cc passim); the C++ wrapper logs to cerr.
 *  - a context is single-caller (not re-entrant), like the reference's single mapping thread
 *    (PM.cc:65-87).  Work is queued on the context's HIP stream; host-pointer calls synchronise.
 *  - there is NO CPU fallback: without a visible gfx950 device sdm_create fails.
 */
#ifndef SDM_C_H
#define SDM_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDM_OK 0
#define SDM_EINVAL 1   /* bad argument (slot, size, null pointer, n > max_neighbours ...) */
#define SDM_EHIP 2     /* HIP runtime error (text in sdm_last_error) */
#define SDM_ENODEV 3   /* no usable GPU */
#define SDM_ESTATE 4   /* slot not uploaded / stage not run yet */
#define SDM_ECOMM 5    /* RCCL error or library not loadable (text in sdm_last_error) */

#define SDM_MAX_NEIGHBOURS 64

typedef struct sdm_ctx sdm_ctx;

/* PM.h:38-49 macros as runtime parameters (defaults = the reference's values). */
typedef struct {
    float lambdaG;     /* 8    PM.h:40  gradient gate (PM.cc:201,411,562)          */
    float lambdaL;     /* 80   PM.h:41  epipolar-line angle gate, deg (PM.cc:421)   */
    float lambdaTheta; /* 45   PM.h:42  orientation gate, deg (PM.cc:431)           */
    int lambdaN;       /* 3    PM.h:43  (PM.cc:221 '>', :623 '>=', :762 '<')        */
    double theta_var;  /* 0.23 PM.h:47  THETA, a double literal (PM.cc:436,455-456) */
} sdm_params;

typedef struct {
    int device;           /* HIP device ordinal */
    int W, H;             /* image size shared by all keyframes */
    int max_keyframes;    /* number of device slots */
    int max_neighbours;   /* <= SDM_MAX_NEIGHBOURS; reference covisN = 7 (PM.h:38) */
    int batch_capacity;   /* keyframes processed per launch group (scratch size); 0 = default */
    int with_pointset;    /* allocate the xyz pool (12 B/px/keyframe) for sdm_pointset */
    void *ext_depth_pool; /* optional caller-owned DEVICE buffer of sdm_depth_pool_bytes(); lets the
                             host framework (e.g. torch.distributed/RCCL) all-gather it in place */
    void *stream;         /* optional hipStream_t to run on (NULL = context-owned stream) */
} sdm_config;

typedef struct {
    long long searches;   /* EpipolarSearch invocations (pixel, neighbour)  PM.cc:213 */
    long long candidates; /* scan-loop iterations                           PM.cc:405 */
    long long gate_pass;  /* candidates that reached the cost               PM.cc:433 */
    long long hypotheses; /* accepted hypotheses                            PM.cc:216 */
    long long fused;      /* pixels written                                 PM.cc:225 */
    /* how K1 walked the ranges (not reference quantities): */
    long long mask_waves; /* (wave, neighbour) scans that took the gradient-mask scan instead of the batched one */
    long long mask_steps; /* mask words those scans examined (lane steps) */
    long long mask_row_mismatch; /* self-check of the mask scan's row runs: must stay 0 */
    long long open_pixels; /* fusing pixels neither shortcut of InverseDepthHypothesisFusion settled: all-pairs count, PM.cc:598-626 */
    long long table_stagings; /* host: compute calls whose slot / constant tables were not found in a cached set (K1 -> K4 -> K5
                                 of one step share one; counted whether or not statistics are enabled) */
} sdm_stats;

/* ---- lifetime ------------------------------------------------------------------------------- */
void sdm_default_params(sdm_params *p);
void sdm_default_config(sdm_config *c);
size_t sdm_depth_pool_bytes(int W, int H, int max_keyframes); /* = 8*W*H*max_keyframes */
int sdm_create(sdm_ctx **out, const sdm_config *cfg); /* replaces ProbabilityMapping(Map*) PM.cc:61 */
void sdm_destroy(sdm_ctx *ctx);
const char *sdm_last_error(void);
int sdm_set_params(sdm_ctx *ctx, const sdm_params *p);
int sdm_get_params(sdm_ctx *ctx, sdm_params *out); /* the parameters in force */
int sdm_set_stream(sdm_ctx *ctx, void *hip_stream);
int sdm_synchronize(sdm_ctx *ctx);
int sdm_device_count(void); /* number of visible HIP devices, 0 if none; never initialises a context */

/* ---- keyframe inputs (the KeyFrame members PM.cc reads) --------------------- */
/* im_: H*W u8; GradImg, GradTheta (degrees [0,360)): H*W f32; I_stddev; K = {fx,fy,cx,cy}
 * (include/KeyFrame.h:162); Tcw = world->camera [R|t] row-major 3x4 (src/KeyFrame.cc:70-121). */
int sdm_upload_keyframe(sdm_ctx *ctx, int slot, const uint8_t *im, const float *grad,
                        const float *theta, float I_stddev, const float K[4], const float Tcw[12]);
/* gray image only: GradImg/GradTheta/I_stddev computed on device (Scharr/32, magnitude,
 * fastAtan2 phase, population sigma) -- the pre-processing the reference leaves to the caller. */
int sdm_upload_image(sdm_ctx *ctx, int slot, const uint8_t *im, const float K[4],
                     const float Tcw[12]);
/* The step before the path: a camera frame as Tracking receives it.  Replaces, on the device,
 * Tracking::GrabImageMonocular's cvtColor(RGB/BGR/RGBA/BGRA -> GRAY) (src/Tracking.cc:244-257), the
 * cv::undistort(im, imu, mK, mDistCoef) of src/Tracking.cc:266-271 and the cvtColor(CV_RGB2GRAY) the Modeler applies
 * to the stored frame (src/Modeler/Modeler.cc:154-155): undistort the colour frame (1/32-pixel fixed-point map,
 * bilinear, zero border), convert to gray (4899/9617/1868 >> 14), then the gradient pre-pass of sdm_upload_image.
 * pixels: H*W interleaved pixels of 1, 3 or 4 bytes.  dist = {k1,k2,p1,p2,k3} as in Examples/Monocular/TUM1.yaml
 * (Camera.k1.. ; src/Tracking.cc:65-75), NULL = the frame is already undistorted.  The fork's Modeler converts
 * with CV_RGB2GRAY whatever Camera.RGB says: pass SDM_ORDER_RGB to reproduce that, the true order for a correct gray.
 * OpenCV is absent here: PARITY UNPINNED, the published algorithms restated. */
#define SDM_ORDER_RGB 0
#define SDM_ORDER_BGR 1
#define SDM_ORDER_RGBA 2
#define SDM_ORDER_BGRA 3
#define SDM_ORDER_GRAY 4
int sdm_upload_image_rgb(sdm_ctx *ctx, int slot, const uint8_t *pixels, int order, const float K[4],
                         const float dist[5], const float Tcw[12]);
/* The same two calls for n keyframes at once (what Tracking / Modeler::AddFrameImage hand over,
 * src/Tracking.cc:244-271, src/Modeler/Modeler.cc:1496-1514, for a whole window of keyframes): ONE launch each of the
 * pre-pass kernels over (keyframe, tile) instead of a dozen launch-latency-sized ones per keyframe, H2D copies on an
 * upload stream overlapping the previous chunk's kernels.  images[i] / pixels[i]: host pointers; memory from
 * sdm_host_alloc (pinned) is read in place by the copy engine, any other memory is staged through a pinned ring.
 * K: [n][4], Tcw: [n][12]; dist (shared by the batch, as is K when dist != NULL) as above.  Every caller buffer is free
 * on return.  Results are bit-identical to n single calls (tests/test_gpu_ingest.py). */
int sdm_upload_images_batch(sdm_ctx *ctx, int n, const int *slots, const uint8_t *const *images,
                            const float *K, const float *Tcw);
int sdm_upload_images_rgb_batch(sdm_ctx *ctx, int n, const int *slots, const uint8_t *const *pixels,
                                int order, const float *K, const float dist[5], const float *Tcw);
/* pinned host memory for frame queues (the fork's Modeler keeps its own copies of the frames, Modeler.cc:1496-1514:
 * kept in memory from here they reach the device without a staging copy); NULL when the allocation fails */
void *sdm_host_alloc(size_t bytes);
void sdm_host_free(void *p);
/* same, image already resident in device memory */
int sdm_upload_image_device(sdm_ctx *ctx, int slot, const void *d_im, const float K[4],
                            const float Tcw[12]);
/* pose changed after bundle adjustment (kf->poseChanged, PM.cc:329) */
int sdm_set_pose(sdm_ctx *ctx, int slot, const float Tcw[12]);
/* read back what the device derived for a slot (for tests): any pointer may be NULL */
int sdm_download_inputs(sdm_ctx *ctx, int slot, uint8_t *im, float *grad, float *theta,
                        float *I_stddev);

/* ---- SemiDenseRecon, PM.h:75 / PM.cc:137-256, batched over n_ref reference keyframes --------- */
/* nbr_slots and rot_deg are [n_ref][n]; rot_deg may be NULL (= 0, "kf pair without
 * covisibility", PM.cc:174-177).  min_depth/max_depth are [n_ref], named as in PM.cc:381-382. */
int sdm_search_fuse(sdm_ctx *ctx, int n_ref, const int *ref_slots, int n, const int *nbr_slots,
                    const float *rot_deg, const float *min_depth,
                    const float *max_depth);                            /* PM.cc:197-231 */
int sdm_intra_check(sdm_ctx *ctx, int n_ref, const int *ref_slots);     /* PM.cc:486-547 */
int sdm_intra_grow(sdm_ctx *ctx, int n_ref, const int *ref_slots);      /* PM.cc:549-596 */
int sdm_recon(sdm_ctx *ctx, int n_ref, const int *ref_slots, int n, const int *nbr_slots,
              const float *rot_deg, const float *min_depth, const float *max_depth);

/* ---- InterKeyFrameDepthChecking, PM.h:91 / PM.cc:628-799 -------------------------------------- */
/* Reads the neighbours' current {rho,sigma}; writes the checked rho of each reference keyframe to
 * its "checked" plane.  commit != 0 also stores it back into the keyframe's depth map, which is
 * the reference's in-place behaviour (call with n_ref = 1, in the caller's keyframe order). */
int sdm_inter_check(sdm_ctx *ctx, int n_ref, const int *ref_slots, int n, const int *nbr_slots,
                    int commit);

/* Precondition (the reference's gate at PM.cc:292-298: the keyframe and ALL its neighbours have
 * semidense_flag_ set): every reference and neighbour slot must hold a depth map -- produced by sdm_recon /
 * sdm_search_fuse, restored by sdm_upload_depth, received by an sdm_exchange_* call, or declared with
 * sdm_assume_pipeline_maps / sdm_mark_depth_present.  Otherwise SDM_ESTATE. */

/* ---- UpdateSemiDensePointSet, PM.h:88 / PM.cc:337-367 ----------------------------------------- */
/* source: 0 = depth map, 1 = checked plane.  Needs with_pointset. */
int sdm_pointset(sdm_ctx *ctx, int n_ref, const int *ref_slots, int source);
/* sdm_inter_check followed by sdm_pointset(source = 1) -- the pair the reference runs per ready keyframe
 * (PM.cc:300-306) -- with the back-projection done in the checking kernel when the maps came out of
 * SemiDenseRecon; identical results to the two calls. */
int sdm_inter_check_pointset(sdm_ctx *ctx, int n_ref, const int *ref_slots, int n, const int *nbr_slots,
                             int commit);

/* ---- map transfer ------------------------------------------------------------------------------ */
int sdm_upload_depth(sdm_ctx *ctx, int slot, const float *rho, const float *sigma);
int sdm_download_depth(sdm_ctx *ctx, int slot, float *rho, float *sigma); /* depth_map_/depth_sigma_ */
int sdm_download_checked(sdm_ctx *ctx, int slot, float *rho);
int sdm_download_pointset(sdm_ctx *ctx, int slot, float *xyz);            /* H x 3W */
/* device addresses for zero-copy interop (RCCL all-gather of per-keyframe {rho,sigma} maps):
 * the depth pool is [max_keyframes][H][W] of float2 {rho,sigma}. */
void *sdm_depth_pool_ptr(sdm_ctx *ctx);
/* The caller asserts that the depth maps of these slots are zero outside the keyframe's active-pixel
 * set {inset pixels with GradImg >= lambdaG} -- true for every map SemiDenseRecon produced, e.g. maps
 * restored through sdm_upload_depth or received by an all-gather.  Lets K2-K4 use their list kernels. */
int sdm_assume_pipeline_maps(sdm_ctx *ctx, int n, const int *slots);

/* Maps written into an ext_depth_pool from outside the engine (the host framework's own collective):
 * marks the slots as holding finished depth maps (kf->semidense_flag_). */
int sdm_mark_depth_present(sdm_ctx *ctx, int n, const int *slots);

/* ---- multi-GPU exchange: the path's one collective step -------------------------- */
/* Keyframes shard in contiguous blocks, one process per GPU.  K1-K3 need no communication; K4
 * (InterKeyFrameDepthChecking, PM.cc:628-799) reads the neighbours' FINISHED {rho,sigma} maps, which
 * cross GPUs once per pass -- over RCCL (xGMI inside a node).  The reference is single-process and has
 * nothing to replace here; SURVEY.md §8(b) sketches this entry as `sdm_allgather`.
 * A communicator is built from an RCCL unique id (rank 0 creates it and hands the 128 bytes to the other
 * ranks through any channel it likes -- a file, MPI, torch.distributed) or borrowed from the caller.
 * With world == 1 every exchange call is a no-op that returns SDM_OK and RCCL is never loaded. */
#define SDM_COMM_ID_BYTES 128
int sdm_comm_unique_id(unsigned char id[SDM_COMM_ID_BYTES]);                 /* ncclGetUniqueId  */
int sdm_comm_init(sdm_ctx *ctx, const unsigned char id[SDM_COMM_ID_BYTES], int world, int rank);
int sdm_comm_attach(sdm_ctx *ctx, void *nccl_comm); /* borrow an existing ncclComm_t (not destroyed) */
int sdm_comm_destroy(sdm_ctx *ctx);
int sdm_comm_info(sdm_ctx *ctx, int *world, int *rank);
/* Halo form: send the maps in send_slot[i] to rank send_peer[i], receive rank recv_peer[i]'s maps into
 * recv_slot[i]; per peer pair the k-th send matches the k-th receive.  _begin returns at once: the
 * transfers run on a second stream behind everything queued so far, and work queued afterwards (the
 * interior keyframes' sdm_recon) overlaps them; sdm_exchange_wait orders later work (sdm_inter_check)
 * behind the transfers.  Neither call waits on the host. */
int sdm_exchange_halo_begin(sdm_ctx *ctx, int n_send, const int *send_peer, const int *send_slot,
                            int n_recv, const int *recv_peer, const int *recv_slot);
int sdm_exchange_wait(sdm_ctx *ctx);
int sdm_exchange_halo(sdm_ctx *ctx, int n_send, const int *send_peer, const int *send_slot,
                      int n_recv, const int *recv_peer, const int *recv_slot); /* begin + wait */
/* All-gather form (BASELINE.json's wording): every rank contributes `count` maps from local slot
 * first_slot on.  n_fetch < 0: in place, the pool holds world*count slots with slot == global
 * keyframe index and first_slot == rank*count.  n_fetch >= 0: gathered into an engine-owned buffer;
 * map number fetch_index[i] (= owner_rank*count + position in the owner's block) is copied into local
 * slot dst_slot[i].  Stream-ordered; no host wait. */
int sdm_allgather_depth(sdm_ctx *ctx, int first_slot, int count, int n_fetch, const int *fetch_index,
                        const int *dst_slot);
/* The all-gather in pieces, overlapped with the reconstruction.  Every rank contributes maps_per_rank maps (the
 * same number on all ranks) in one or more pieces: _piece(count, slots) -- count equal on all ranks, padded by
 * repeating a slot if a rank has fewer -- is called right after the sdm_recon of those keyframes and gathers them on a
 * second stream behind an event, so the next sdm_recon overlaps the transfer; _finish orders later work
 * (sdm_inter_check) behind the last piece and copies the maps this rank reads (fetch_index = owner_rank *
 * maps_per_rank + position in the owner's contribution order) into dst_slot.  Used for the whole block in sub-blocks,
 * or for just the keyframes other ranks read (a third of the bytes on an index-local covisibility graph).
 * The fetch copies run on the second stream too: between _begin and _finish the caller must not queue work that
 * reads or writes the depth maps of the dst_slot keyframes -- anything else (sdm_recon of the next keyframes,
 * sdm_inter_check of keyframes whose neighbours are all local) overlaps the transfer AND the copies.
 * No host wait.  world == 1: the pieces are device copies and the fetch addressing still runs. */
int sdm_allgather_begin(sdm_ctx *ctx, int maps_per_rank);
int sdm_allgather_piece(sdm_ctx *ctx, int count, const int *slots);
int sdm_allgather_finish(sdm_ctx *ctx, int n_fetch, const int *fetch_index, const int *dst_slot);
/* Wire format of the maps that cross ranks in sdm_exchange_halo[_begin] and sdm_allgather_piece / _finish.
 * entries_per_map = 0 (default): whole maps, 8*W*H bytes each.  > 0: the {rho,sigma} of the first entries_per_map entries
 * of the keyframe's active-pixel list (the pixels that pass the gradient gate, PM.cc:201), in list order -- all a
 * reconstructed map holds, since it is zero elsewhere: 8*entries_per_map bytes (a fifth of the map on typical images).
 * The receiver scatters them through ITS list of that keyframe, which it has because the keyframe is part of its input
 * halo (SDM_ESTATE if the destination slot holds no keyframe).  Every rank must set the same value, at least the longest
 * list (sdm_active_count) among the keyframes it sends or receives; a call that meets a longer one fails with SDM_ESTATE
 * before posting anything -- so agree on the value across ranks beforehand (bench.py: an all-reduce(max) at set-up).
 * sdm_allgather_depth (the one-shot form) always moves whole maps and refuses to run while entries_per_map > 0.
 * sdm_comm_destroy resets the format to whole maps. */
int sdm_exchange_compact(sdm_ctx *ctx, int entries_per_map);
/* Every compact map carries its sender's list length and the 64-bit hash of its list; a receiver whose list of that
 * keyframe differs in either (its image differs from the sender's) does not scatter the map -- it would land on wrong
 * pixels -- and counts the event: a destination that already held a pipeline map keeps it; one that did not (an arbitrary
 * map, e.g. from sdm_upload_depth) has been zeroed for the scatter and stays zero.  *count = such maps since the last call
 * (host-blocking; 0 on a healthy job). */
int sdm_exchange_mismatches(sdm_ctx *ctx, int *count);
/* The compact wire format through host memory, one map per call (host-blocking): the payload the RCCL forms would send
 * for `slot` -> out[2 * (entries_per_map + 8)] floats ({rho,sigma} of the list entries, then the header: list length
 * and hash), and the receiving side for a payload that travelled by another route (*refused = 1: packed with another
 * list; checked on the host before anything is written: the slot's plane is left alone).  Same kernels and checks as sdm_exchange_* -- for transports the engine does not
 * drive (a host framework's own, or gloo on a one-GPU box: shard.py). */
int sdm_compact_pack_host(sdm_ctx *ctx, int slot, float *out);
int sdm_compact_unpack_host(sdm_ctx *ctx, int slot, const float *in, int *refused);
/* *ready = 1 iff every listed slot's map would be accepted as a compact SOURCE now (a pipeline map under the current
 * lambdaG); the query form of the check the compact sends make.  Fold it into the per-pass wire-format agreement
 * (ProbabilityMapping::SemiDenseReconBlock does): a rank that answers 0 makes all ranks use whole maps for the pass. */
int sdm_compact_sources_ready(sdm_ctx *ctx, int n, const int *slots, int *ready);
/* Length of the slot's active-pixel list under the current lambdaG (built now if need be; host-blocking). */
int sdm_active_count(sdm_ctx *ctx, int slot, int *count);
/* The list itself, (y << 16 | x) of the inset pixels with GradImg >= lambdaG in raster order (PM.cc:198-201), and the
 * 64-bit hash of that pixel set the compact wire header carries; any output may be NULL.  For tests and for rehearsing
 * the compact exchange on host arrays (orb-slam-free-space-carving_amd/shard.py). */
int sdm_download_active_list(sdm_ctx *ctx, int slot, unsigned *list, int capacity, int *count,
                             unsigned long long *hash);
/* Go / no-go before a collective pass: all ranks call it; *all_ok = min over ranks of local_ok (host-blocking).
 * A rank that cannot take part in the exchange it planned reports it here, so peers skip the pass instead of
 * waiting for transfers that never come. */
int sdm_comm_all_ok(sdm_ctx *ctx, int local_ok, int *all_ok);
/* *all_max = max over ranks of local_value (all ranks call it; host-blocking; world size 1: the local value) -- e.g. the
 * longest active list of the job, for sdm_exchange_compact. */
int sdm_comm_all_max(sdm_ctx *ctx, int local_value, int *all_max);

/* ---- stand-alone map operations with the reference's signatures -------------------------------- */
/* IntraKeyFrameDepthChecking(cv::Mat&, cv::Mat&, const cv::Mat) PM.h:85; host maps, in place. */
int sdm_intra_check_maps(sdm_ctx *ctx, float *rho, float *sigma, const float *grad);
/* IntraKeyFrameDepthGrowing(cv::Mat&, cv::Mat&, const cv::Mat)  PM.h:86 */
int sdm_intra_grow_maps(sdm_ctx *ctx, float *rho, float *sigma, const float *grad);

/* ---- per-pixel entry points (unit tests; same device functions as the batched kernels) -------- */
/* EpipolarSearch PM.h:79: out = {rho, sigma, supported(0/1), best_u, best_v} */
int sdm_epipolar_search(sdm_ctx *ctx, int ref_slot, int nbr_slot, int x, int y, float min_depth,
                        float max_depth, float rot_deg, float out[5]);
/* GetSearchRange PM.h:80 */
int sdm_search_range(sdm_ctx *ctx, int ref_slot, int nbr_slot, int x, int y, float mind,
                     float maxd, float *umin, float *umax);
/* InverseDepthHypothesisFusion PM.h:83: out = {rho, sigma, supported(0/1)} */
int sdm_fuse(sdm_ctx *ctx, const float *rho, const float *sigma, int n, float out[3]);
/* ComputeFundamental PM.h:101 and R21/t21 (PM.cc:859-860): F12[9], R21[9], t21[3] */
int sdm_pair_geometry(sdm_ctx *ctx, int ref_slot, int nbr_slot, float F12[9], float R21[9],
                      float t21[3]);

/* ---- host helpers of the class surface (no device work) ---------------------------------------- */
/* StereoSearchConstraints PM.h:77 / PM.cc:370-383 over the keyframe's ORB point depths */
int sdm_stereo_search_constraints(const float *orb_depths, int n, float *min_depth,
                                  float *max_depth);
/* GetRotInPlane PM.h:103 / PM.cc:467-484 + the median of PM.cc:170-179; ids < 0 = no map point */
float sdm_median_rot_in_plane(const int *mp1, const float *angle1, int n1, const int *mp2,
                              const float *angle2, int n2);

/* ---- instrumentation ---------------------------------------------------------------------------- */
/* when enabled, sdm_search_fuse runs its counting variant (slower) and accumulates into stats */
int sdm_enable_stats(sdm_ctx *ctx, int on);
int sdm_get_stats(sdm_ctx *ctx, sdm_stats *out, int reset);
/* diagnostic: how sdm_search_fuse / sdm_recon / sdm_epipolar_search walk the candidate range of PM.cc:405 -- 0 (default): per
 * wave and neighbour, from the wave's range lengths and line slopes; 1: always the batched scan; 2: always the scan over the
 * neighbour's gradient-gate bit plane.  Results are the same bit for bit in every mode (tests/test_gpu_longscan.py). */
int sdm_set_scan_mode(sdm_ctx *ctx, int mode);
/* Streaming ingest (default off).  On: the batched uploads (sdm_upload_images_batch / _rgb_batch) get twelve chunk buffers
 * instead of four, so the host staging and the H2D copies (upload stream) of up to three 64-keyframe blocks run ahead of
 * their pre-pass kernels, which stay on the compute stream in call order (and, the copies being ahead anyway, run as ONE
 * launch per group of four chunks behind one wait for the group's last copy).  A block uploaded BEFORE the previous block's step
 * is queued (double-buffer the slot sets: bench.py streaming_rate; frames arrive continuously in the fork,
 * src/Tracking.cc:266-271) is then copied while that step runs, pre-processed right behind it, and its list lengths are back
 * before its own step is queued (every compute call waits for the list lengths of ITS slots only, with or without this
 * switch).  Results are the serial order's by construction (one compute stream).  Costs 8 x the chunk size of device and
 * pinned host memory (8 x 4.9 MB at 640x480).  Host-blocking when it allocates. */
int sdm_set_ingest_overlap(sdm_ctx *ctx, int on);
/* per-stage device time measured with HIP events recorded on the context's stream around each
 * stage's kernel launches (K1 = one k_search_fuse launch per sdm_recon/sdm_search_fuse call) */
#define SDM_STAGE_SEARCH_FUSE 0 /* K1   PM.cc:197-231 */
#define SDM_STAGE_INTRA 1       /* K2+K3 PM.cc:237-238 */
#define SDM_STAGE_INTER 2       /* K4   PM.cc:628-799 */
#define SDM_STAGE_POINTSET 3    /* K5   PM.cc:337-367 */
#define SDM_NUM_STAGES 4
int sdm_enable_timing(sdm_ctx *ctx, int on);
int sdm_get_timing(sdm_ctx *ctx, double ms_total[SDM_NUM_STAGES], long long launches[SDM_NUM_STAGES],
                   int reset);
/* device arithmetic self-tests: out[0] = mismatches (must be 0), out[1] = auxiliary count.
 * which 0: reciprocal+FMA division by theta_var vs plain division over all 2^32 float inputs;
 * which 1: reciprocal-prefiltered ChiTest vs the exact ChiTest around the 5.99 threshold;
 * which 2: fast matching cost (PM.cc:436) vs the reference expression incl. rounding midpoints, and the distance of
 *          the scan's approximate cost from it (at most 4 float steps: the arg-min of PM.cc:437 compares exactly);
 * which 3: closed-form angle gates (PM.cc:414-431) vs the reference statement;
 * which 4: GetFusion's shared-reciprocal double quotients (PM.cc:956-957) vs plain divisions;
 * which 5: reciprocal-form float quotients (K4, PM.cc:678-680,782-783; the line quotients PM.cc:393,407) vs IEEE
 *          divisions inside the operand windows, all-ones divisor significands included;
 * which 6: reciprocal + one FMA step (K4/K5, PM.cc:769,777,793,349) vs IEEE 1/b, and rsq + one FMA step (PM.cc:818)
 *          vs sqrtf, over all 2^32 float inputs;
 * which 7: K4's straight-line per-neighbour body vs the reference statement (PM.cc:677-755,777-783) on 5*10^8
 *          random geometries / 2x2 tap patches; out[1] = cases whose fast result was accepted;
 * which 8: scan identities (lerp weight, in-plane-rotation wrap) over all 2^32 float inputs;
 * which 9: the same run as 7, reporting out[1] = projections (PM.cc:677-680, 695) decided by K4's approximate chain --
 *          each compared with the plain-division chain's cell, validity and offset. */
int sdm_selftest(sdm_ctx *ctx, int which, unsigned long long out[2]);
/* name of the device the context runs on, e.g. "gfx950" */
const char *sdm_device_arch(sdm_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
