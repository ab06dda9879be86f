// sdm_engine.hip -- context, memory pools and the C ABI (include/sdm_c.h) of libsdm_hip.so.
//
// Data layout in HBM (per keyframe slot, P = W*H pixels):
//   rec   float4[P]   16 B/px  search records {grad, theta, grad(y+1), im|im(y+1)<<8}   (inputs)
//   pool  float2[P]    8 B/px  depth map {rho, sigma}  = kf->depth_map_/depth_sigma_     (K1-K3)
//   chk   float [P]    4 B/px  inter-keyframe-checked rho                                 (K4)
//   xyz   float [3P]  12 B/px  SemiDensePointSets_ (optional)                             (K5)
// plus one scratch pool of batch_capacity slots for the Jacobi stencil passes (arbitrary maps: K2 writes scratch planes,
// K3 writes back; pipeline maps: K2's results compactly by list position, then committed -- sdm_kernels.h), one staging
// keyframe for uploads, and the per-batch constant tables.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sdm_c.h"
#include "sdm_kernels.h"

using namespace sdm;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess)                                                                \
            return fail(SDM_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__));        \
    } while (0)

}  // namespace

struct sdm_ctx {
    sdm_config cfg{};
    int W = 0, H = 0;
    long long P = 0;
    TileGeom geom{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string arch;

    float4* rec = nullptr;
    float2* pool = nullptr;
    bool own_pool = false;
    float2* scratch = nullptr;
    float* chk = nullptr;
    float* xyz = nullptr;

    KfMeta* d_meta = nullptr;
    std::vector<KfMeta> h_meta;
    std::vector<char> has_depth, has_chk;
    unsigned* d_act = nullptr;     // [max_keyframes][P] active-pixel lists (y<<16|x), raster order
    int* d_act_count = nullptr;    // [max_keyframes]
    int* d_theta_bad = nullptr;    // [max_keyframes] GradTheta plane holds a value outside [0,360] (k_pack)
    // K1's open-pixel list (pixels whose fusion the bounds do not settle; finished by k_fuse_open)
    unsigned* d_open_ctr = nullptr;  // two sets of 4 counters (OpenList::count), used alternately by successive launches
    long long* d_open_pix = nullptr;
    unsigned long long* d_open_vm = nullptr;
    float2* d_open_hyp = nullptr;
    unsigned open_capacity = 0;
    unsigned open_quota = 512;
    unsigned open_inplace_min = 40;
    unsigned open_grid = 2048;  // workgroups of k_fuse_open (grid-stride over the list's blocks): 8 per CU
    unsigned open_launch = 0;
    // K3's candidate list on pipeline maps (listed pixels with rho < 1e-6 and a non-zero sigma; normally empty)
    unsigned* d_grow_ctr = nullptr;  // two counters, used alternately by successive K2 launches
    long long* d_grow_pix = nullptr;
    float2* d_grow_val = nullptr;
    unsigned grow_capacity = 0;
    unsigned grow_launch = 0;
    unsigned k4_lds_pad = 0;  // experiment knob (SDM_K4_PAD): dynamic LDS requested by K4's list kernel = an occupancy cap
    int* h_act_count = nullptr;    // pinned host mirror (device-visible): k_prepass_finish stores the lengths into it, read behind the cnt_ev events
    std::vector<float> act_lambdaG;  // lambdaG each list was built with (NaN = no list)
    std::vector<char> chk_sparse, xyz_sparse;  // checked / xyz plane of the slot is zero outside its active list
    std::vector<float> recon_lambdaG;  // lambdaG a slot's depth map was reconstructed with (NaN = map
                                       // came from elsewhere, e.g. sdm_upload_depth): K4 may use the list

    // staging for one keyframe given as planes (sdm_upload_keyframe) and for sdm_download_inputs
    uint8_t* d_im = nullptr;
    // ---- batched ingest (sdm_ingest.h): two chunk buffers, filled by H2D copies on the upload stream while the compute
    // stream runs the pre-pass of the other one
    int ing_cap = 0;  // keyframes per chunk (their gray images together <= 32 MB)
    struct IngestBuf {
        uint8_t* d_img = nullptr;       // [ing_cap][P] gray images
        uint8_t* h_ring = nullptr;      // pinned staging for pageable host images (same size)
        uint8_t* d_src = nullptr;       // colour frames (sdm_upload_image_rgb*), src_bytes, allocated on first use
        uint8_t* h_src = nullptr;       // pinned staging for pageable colour frames
        IngestItem* h_items = nullptr;  // pinned
        IngestItem* d_items = nullptr;
        hipEvent_t copied = nullptr;    // this chunk's H2D copies (upload stream) have finished
        hipEvent_t consumed = nullptr;  // the kernels that read d_img / d_src / d_items (compute stream) have finished
        bool copied_pending = false, consumed_pending = false;
        int consumed_from = -1;         // >= 0: the `consumed` event of THAT buffer covers this one too (one record per merged launch)
    } ing[12];
    static constexpr int ING_BUFS_MAX = 12;
    int ing_bufs = 4;  // in use: four (a 64-keyframe block goes through in four chunks without a copy waiting for an earlier
                       // chunk's kernels); twelve with overlapped ingest on (three blocks in flight), allocated when it is switched on
    int ing_next = 0;
    size_t src_bytes = 0;
    hipStream_t up_stream = nullptr;
    // per GROUP of up to ING_GROUP chunks: the chunks' first kernels fill these, k_prepass_finish / k_list_write run once per group
    static constexpr int ING_GROUP = 4;
    static_assert(ING_GROUP == CHUNK_TABLES, "one merged launch covers a group");
    unsigned long long* d_part = nullptr;      // [ING_GROUP * ing_cap][ntiles][PART_WORDS] per-tile partial sums
    unsigned long long* d_seg_mask = nullptr;  // [ING_GROUP * ing_cap][nseg] lambdaG-gate lane mask of every 64-pixel row segment
    int* d_seg_off = nullptr;                  // [ING_GROUP * ing_cap][nseg] list offset of every row segment
    IngestItem* d_gitems = nullptr;            // [ING_GROUP * ing_cap] the group's keyframes (copied from the chunks' tables)
    int nseg = 0;                              // H * tiles_x
    unsigned long long* d_act_hash = nullptr;  // [max_keyframes] hash of the active-pixel set (compact wire header)
    unsigned* d_gmask = nullptr;               // [max_keyframes][H][mrow][MASK_PLANES] gate bit planes of every slot, one dword
                                               // per (row, 32-column word, plane): bit x%32 = pixel x passes the gradient gate (and
                                               // its angle lies in the plane's bin), written with the slot's list (k_prepass_batch /
                                               // k_gate_batch), read by K1's mask scan (sdm_device.h scan_masked)
    int mrow = 0;                              // 32-bit words per image row: 2 * (tiles_x + 1) (the last two stay zero: the scan
                                               // reads the word after the one a column lies in)
    int scan_mode = 0;                         // DevParams::scan_mode (SDM_SCAN_MODE, read once in sdm_create)
    // ---- streaming ingest (sdm_set_ingest_overlap): with twelve chunk buffers instead of four, the host staging and the H2D
    // copies of a batch upload (upload stream) run ahead of the pre-pass kernels (compute stream, in call order) by up to
    // three 64-keyframe blocks: a block uploaded BEFORE the previous block's step is queued is copied while that step runs
    bool ingest_overlap = false;
    // list-length read-backs: one event per upload / list rebuild (a ring; all recorded on the compute stream, so they
    // complete in order), and per slot the call whose read-back it still awaits -- a compute call waits for ITS slots'
    // lengths only, not for the stream to drain (the step that is executing) nor for a block uploaded behind it
    static constexpr int CNT_RING = 16;
    hipEvent_t cnt_ev[CNT_RING] = {};
    unsigned long long cnt_next = 1, cnt_done = 0;   // ids of the next / the newest completed read-back
    std::vector<unsigned long long> slot_cnt;        // [max_keyframes] id of the read-back the slot's length awaits (0: none)
    bool validated = false;                    // validate_params: the last non-default parameter set checked on the device ...
    sdm_params validated_prm{};
    int validated_closed = 0, validated_approx = 0;  // ... and what held for it
    float* d_grad = nullptr;
    float* d_theta = nullptr;
    float* d_small = nullptr;  // 16 floats of per-pixel results

    // per-call tables: one packed device block + pinned host mirror, staged with a single copy, and the
    // RefConst/PairConst blocks built from it.  TABLE_SETS such sets are kept, each remembering the call it was
    // staged for: a call whose tables equal a cached set's (same slots, same constants, nothing re-uploaded since)
    // reuses that set without any copy, set-up kernel or host wait -- K1->K4->K5 of one step, and the boundary /
    // interior / whole-block calls of the multi-GPU step, which alternate between three table sets.  The members
    // below the array alias the set selected for the current call.
    int cap_refs = 0;
    size_t tab_bytes = 0;
    struct TableKey {
        bool valid = false, has_consts = false;
        bool long_ranges = false;  // some pair's search range at the principal point is long: K1 runs its mask-scan instantiation
        int n_ref = 0, n = 0;
        unsigned long long epoch = 0;
        std::vector<int> refs, nbrs;
        std::vector<float> rot, mind, maxd;
    };
    static constexpr int TABLE_SETS = 8;  // sub-block calls of the pipelined all-gather step + the whole-block K4 call
    struct TableSet {
        unsigned char *d_tab = nullptr, *h_tab = nullptr;
        RefConst* d_refs = nullptr;
        PairConst* d_pairs = nullptr;
        TableKey key;
        hipEvent_t free_ev = nullptr;  // the pinned block may be rewritten once this has passed
        bool pending = false;
        unsigned long long last_use = 0;
    } sets[TABLE_SETS];
    int cur_set = 0;
    unsigned long long use_tick = 0;
    unsigned char *d_tab = nullptr, *h_tab = nullptr;
    int *d_ref_slots = nullptr, *d_nbr_slots = nullptr;
    float *d_rot = nullptr, *d_mind = nullptr, *d_maxd = nullptr;
    long long *d_off = nullptr, *h_off = nullptr;  // 3 offset tables of n_ref: pool, scratch, record
    unsigned long long epoch = 1;  // bumped whenever poses, intrinsics, lists or slots change
    long long table_stagings = 0;  // calls whose tables were not in a cached set (sdm_stats::table_stagings)
    RefConst* d_refs = nullptr;
    PairConst* d_pairs = nullptr;

    float2* h_f2 = nullptr;  // pinned interleave buffer, P elements

    sdm_params prm{};
    DevParams dprm{};
    bool stats_on = false;
    unsigned long long* d_stats = nullptr;

    // optional per-stage HIP-event timing (sdm_enable_timing)
    struct Span {
        hipEvent_t a, b;
        int stage;
    };
    bool timing_on = false;
    std::vector<Span> spans;
    size_t spans_used = 0;

    // multi-GPU exchange (sdm_comm.h): RCCL communicator, exchange stream, ordering events
    void* comm = nullptr;  // ncclComm_t
    bool own_comm = false;
    int world = 1, rank = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_maps_ready = nullptr, ev_xchg_done = nullptr;
    bool xchg_pending = false;
    float2* gather_buf = nullptr;  // [world*count] maps, the not-in-place all-gather's landing zone
    long long gather_slots = 0;
    // all-gather in pieces (sdm_allgather_begin / _piece / _finish)
    struct AgPiece {
        int offset, count;
    };
    bool ag_open = false;
    int ag_count = 0, ag_covered = 0;  // maps every rank contributes / contributed so far
    std::vector<AgPiece> ag_pieces;
    std::vector<char> ag_contributed;   // [max_keyframes] local slots this rank has contributed
    float2* stage_buf = nullptr;        // packing buffer for pieces that are not runs of consecutive slots
    int stage_slots = 0;
    int* d_agree = nullptr;  // sdm_comm_all_ok
    int xchg_entries = 0;    // sdm_exchange_compact: > 0 = maps cross ranks as their first xchg_entries active-list entries
    unsigned* d_xchg_mismatch = nullptr;  // compact maps refused by k_unpack_lists (list lengths differ)
};

namespace {

void set_dev_params(sdm_ctx* c)
{
    c->dprm.lambdaG = c->prm.lambdaG;
    c->dprm.lambdaL = c->prm.lambdaL;
    c->dprm.lambdaTheta = c->prm.lambdaTheta;
    c->dprm.lambdaN = c->prm.lambdaN;
    c->dprm.theta_var = c->prm.theta_var;
    c->dprm.inv_theta = 1 / c->prm.theta_var;  // (1/THETA), PM.cc:455
    c->dprm.fast_theta_div = (c->prm.theta_var == 0.23) ? 1 : 0;
    c->dprm.default_gates = (c->prm.lambdaL == 80.0f && c->prm.lambdaTheta == 45.0f) ? 1 : 0;
    c->dprm.scan_mode = c->scan_mode;
    // constants of the closed-form gates for the thresholds in force (sdm_device.h gate2_fails_k / gate3_fails_k): the real
    // bounds 90 - lambdaL and 360 - lambdaTheta are exact in double; a float compared with "<" against a real bound is
    // compared against the smallest float at or above it.  Whether the forms hold for these values is decided on the device
    // (validate_params), not here.
    auto round_up = [](double v) {
        float f = (float)v;
        if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
        return f;
    };
    const float lL = c->prm.lambdaL, lT = c->prm.lambdaTheta;
    c->dprm.g2c = (lL == lL) ? round_up(90.0 - (double)lL) : 0.0f;  // (NaN: "d > NaN" never fails)
    if (lT >= 0.0f && lT < 180.0f) {
        unsigned lo, hi;
        const float t_up = round_up(360.0 - (double)lT);
        memcpy(&lo, &lT, 4);
        memcpy(&hi, &t_up, 4);
        c->dprm.g3lo = lo + 1u;
        c->dprm.g3span = hi - (lo + 1u);
    } else if (lT < 0.0f) {  // every non-NaN distance exceeds a negative threshold
        c->dprm.g3lo = 0u;
        c->dprm.g3span = 0x7F800001u;
    } else {  // >= 180 or NaN: nothing fails
        c->dprm.g3lo = 0u;
        c->dprm.g3span = 0u;
    }
    c->dprm.inv_theta_f = (float)(1.0 / c->prm.theta_var);
    c->dprm.bins_ok = (lT >= 0.0f && lT <= MASK_MAX_LAMBDA_THETA) ? 1 : 0;
    const bool defaults = c->dprm.default_gates && c->dprm.fast_theta_div;
    c->dprm.closed_ok = c->dprm.default_gates;  // the defaults are covered by sdm_selftest(3); anything else: validate_params
    c->dprm.approx_ok = defaults ? 1 : 0;
}

// Thresholds other than the defaults: run the two device self-tests for exactly these values -- every float d in [-400, 360)
// through both gate forms, and the sampled distance between the approximate and the reference matching cost -- and switch
// the closed forms / the approximate arg-min on only if they hold (a few milliseconds, once per distinct parameter set).
int validate_params(sdm_ctx* c)
{
    if (c->dprm.default_gates && c->dprm.fast_theta_div) return SDM_OK;
    if (c->validated && memcmp(&c->validated_prm, &c->prm, sizeof(sdm_params)) == 0) {
        c->dprm.closed_ok = c->validated_closed;
        c->dprm.approx_ok = c->validated_approx;
        return SDM_OK;
    }
    HIP_TRY(hipSetDevice(c->cfg.device));
    unsigned long long out[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(unsigned long long) * STATS_WORDS, c->stream));
    hipLaunchKernelGGL(k_selftest_gates, dim3(4096), dim3(BLOCK), 0, c->stream, c->dprm, c->d_stats + 4, c->d_stats + 5);
    hipLaunchKernelGGL(k_selftest_cost, dim3(256), dim3(BLOCK), 0, c->stream, c->dprm, 256, c->d_stats + 6, c->d_stats + 7);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->d_stats + 4, sizeof(out), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(unsigned long long) * STATS_WORDS, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const double th = c->prm.theta_var;
    c->validated_closed = (out[0] == 0 && out[1] > 0) ? 1 : 0;
    c->validated_approx = (out[2] == 0 && th >= 1.0e-6 && th <= 1.0e6) ? 1 : 0;
    c->validated_prm = c->prm;
    c->validated = true;
    c->dprm.closed_ok = c->validated_closed;
    c->dprm.approx_ok = c->validated_approx;
    return SDM_OK;
}

int blocks_for(long long n) { return (int)((n + BLOCK - 1) / BLOCK); }

// ---- ingest chunks ---------------------------------------------------------------------------------------------------------
// chunk buffer b is free: the host may rewrite its pinned blocks (their H2D copies have finished) and the upload stream may
// overwrite its device blocks once the kernels that read them have finished
int ingest_acquire(sdm_ctx* c, int b, hipStream_t copy_stream)
{
    sdm_ctx::IngestBuf& B = c->ing[b];
    if (B.copied_pending) {
        HIP_TRY(hipEventSynchronize(B.copied));
        B.copied_pending = false;
    }
    if (B.consumed_pending) {
        // (the buffers are reused in ring order, so a group's last buffer has not been recorded again when an earlier one is
        // acquired; if it had been, this would wait for later work, never for less)
        HIP_TRY(hipStreamWaitEvent(copy_stream, B.consumed_from >= 0 ? c->ing[B.consumed_from].consumed : B.consumed, 0));
        B.consumed_pending = false;
    }
    return SDM_OK;
}
// the chunk's item table goes up behind whatever image copies were queued on the upload stream; the compute stream waits
// for all of it
int ingest_publish(sdm_ctx* c, int b, int m, hipStream_t copy_stream, bool compute_waits = true)
{
    const hipStream_t ks = c->stream;
    sdm_ctx::IngestBuf& B = c->ing[b];
    HIP_TRY(hipMemcpyAsync(B.d_items, B.h_items, sizeof(IngestItem) * (size_t)m, hipMemcpyHostToDevice, copy_stream));
    HIP_TRY(hipEventRecord(B.copied, copy_stream));
    B.copied_pending = true;
    if (copy_stream != ks && compute_waits) HIP_TRY(hipStreamWaitEvent(ks, B.copied, 0));
    return SDM_OK;
}
// the list lengths of these slots are on their way (k_prepass_finish stores them into the pinned host mirror): one event behind
// the kernels queued so far; the host reads h_act_count only after sync_counts*()
int counts_queued(sdm_ctx* c, int n, const int* slots)
{
    const unsigned long long g = c->cnt_next++;
    HIP_TRY(hipEventRecord(c->cnt_ev[g % sdm_ctx::CNT_RING], c->stream));
    for (int i = 0; i < n; i++) c->slot_cnt[(size_t)slots[i]] = g;
    return SDM_OK;
}
// a chunk's first kernel(s) on the compute stream: its m keyframes are numbers kf0 .. kf0 + m - 1 of their group
int ingest_launch_chunk(sdm_ctx* c, int b, int m, int kf0, bool from_images, const IngestParams* q)
{
    const hipStream_t ks = c->stream;
    sdm_ctx::IngestBuf& B = c->ing[b];
    const int tiles_x = c->geom.tiles_x, ntiles = c->geom.ntiles;
    if (kf0 < 0 || kf0 + m > sdm_ctx::ING_GROUP * c->ing_cap) return fail(SDM_EINVAL, "ingest group overflow (internal)");
    if (q) hipLaunchKernelGGL(k_ingest_batch, dim3(blocks_for(c->P), m), dim3(BLOCK), 0, ks, B.d_items, c->W, c->H, *q);
    if (from_images) {
        ChunkTables tabs;
        for (int i = 0; i < CHUNK_TABLES; i++) tabs.p[i] = B.d_items;
        tabs.per = std::max(m, 1);
        hipLaunchKernelGGL(k_prepass_batch<true>, dim3(ntiles, m), dim3(BLOCK), 0, ks, tabs, c->W, c->H, tiles_x, c->P, c->rec,
                           c->pool, c->chk, c->xyz, c->dprm.lambdaG, c->d_part, c->d_seg_mask, c->nseg, c->d_gmask, c->mrow, kf0,
                           c->d_gitems);
    } else {
        hipLaunchKernelGGL(k_gate_batch, dim3(ntiles, m), dim3(BLOCK), 0, ks, B.d_items, c->W, c->H, tiles_x, c->P, c->rec,
                           c->dprm.lambdaG, c->d_part, c->d_seg_mask, c->nseg, c->d_gmask, c->mrow, kf0, c->d_gitems);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(B.consumed, ks));
    B.consumed_pending = true;
    B.consumed_from = -1;
    return SDM_OK;
}
// the same for the chunks of a whole group in ONE launch (gray images whose copies were queued on the upload stream and are
// not awaited chunk by chunk: the streaming ingest): chunk i lies in buffer bufs[i] and holds `per` keyframes, the last one
// total - (n_chunks - 1) * per
int ingest_launch_chunks_merged(sdm_ctx* c, int n_chunks, const int* bufs, int per, int total)
{
    const hipStream_t ks = c->stream;
    if (n_chunks < 1 || n_chunks > CHUNK_TABLES || total > sdm_ctx::ING_GROUP * c->ing_cap || per < 1 ||
        total <= (n_chunks - 1) * per || total > n_chunks * per)
        return fail(SDM_EINVAL, "merged ingest launch: bad chunk shape (internal)");
    ChunkTables tabs;
    for (int i = 0; i < CHUNK_TABLES; i++) tabs.p[i] = c->ing[bufs[std::min(i, n_chunks - 1)]].d_items;
    tabs.per = per;
    HIP_TRY(hipStreamWaitEvent(ks, c->ing[bufs[n_chunks - 1]].copied, 0));  // (the upload stream copies in order)
    hipLaunchKernelGGL(k_prepass_batch<true>, dim3(c->geom.ntiles, total), dim3(BLOCK), 0, ks, tabs, c->W, c->H, c->geom.tiles_x,
                       c->P, c->rec, c->pool, c->chk, c->xyz, c->dprm.lambdaG, c->d_part, c->d_seg_mask, c->nseg, c->d_gmask,
                       c->mrow, 0, c->d_gitems);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ing[bufs[n_chunks - 1]].consumed, ks));
    for (int i = 0; i < n_chunks; i++) {
        c->ing[bufs[i]].consumed_pending = true;
        c->ing[bufs[i]].consumed_from = i == n_chunks - 1 ? -1 : bufs[n_chunks - 1];
    }
    return SDM_OK;
}
// ... and the two launches that finish a group of chunks (n_kf keyframes): list offsets, lengths (device table and the host's
// pinned mirror), hashes, metadata; then the lists
int ingest_launch_group(sdm_ctx* c, int n_kf, bool from_images)
{
    if (n_kf <= 0) return SDM_OK;
    const hipStream_t ks = c->stream;
    hipLaunchKernelGGL(k_prepass_finish, dim3(n_kf), dim3(FIN_BLOCK), 0, ks, c->d_gitems, c->W, c->H, c->geom.ntiles, c->nseg,
                       c->d_part, c->d_seg_mask, c->d_seg_off, c->d_meta, c->d_act_count, c->h_act_count, c->d_theta_bad,
                       c->d_act_hash, from_images ? 1 : 0);
    const int waves = (c->nseg + LIST_SEGS - 1) / LIST_SEGS;
    hipLaunchKernelGGL(k_list_write, dim3((waves + BLOCK / 64 - 1) / (BLOCK / 64), n_kf), dim3(BLOCK), 0, ks, c->d_gitems,
                       c->geom.tiles_x, c->nseg, c->P, c->d_seg_mask, c->d_seg_off, c->d_act);
    HIP_TRY(hipGetLastError());
    return SDM_OK;
}

// (re)build the active-pixel lists of `n` slots from their records for the current lambdaG (sdm_set_params changed it, or
// the records came from the caller's own planes); the counts come back asynchronously
int rebuild_lists(sdm_ctx* c, int n, const int* slots)
{
    int rc;
    int group_kfs = 0, group_chunks = 0;
    for (int i0 = 0; i0 < n; i0 += c->ing_cap) {
        const int m = std::min(c->ing_cap, n - i0);
        const int b = c->ing_next;
        c->ing_next = (c->ing_next + 1) % c->ing_bufs;
        if ((rc = ingest_acquire(c, b, c->stream))) return rc;
        for (int i = 0; i < m; i++) {
            IngestItem& it = c->ing[b].h_items[i];
            memset(&it, 0, sizeof(it));
            it.slot = slots[i0 + i];
        }
        if ((rc = ingest_publish(c, b, m, c->stream))) return rc;
        if ((rc = ingest_launch_chunk(c, b, m, group_kfs, false, nullptr))) return rc;
        group_kfs += m;
        if (++group_chunks == sdm_ctx::ING_GROUP || i0 + m >= n) {
            if ((rc = ingest_launch_group(c, group_kfs, false))) return rc;
            group_kfs = group_chunks = 0;
        }
    }
    if ((rc = counts_queued(c, n, slots))) return rc;
    for (int i = 0; i < n; i++) {
        const int slot = slots[i];
        if (c->act_lambdaG[slot] == c->act_lambdaG[slot]) {
            // the slot had a list under another lambdaG: its checked / xyz planes were written through that list and may
            // be non-zero outside the new one
            c->chk_sparse[slot] = 0;
            c->xyz_sparse[slot] = 0;
        }
        c->act_lambdaG[slot] = c->dprm.lambdaG;
    }
    c->epoch++;
    return SDM_OK;
}
int build_active(sdm_ctx* c, int slot) { return rebuild_lists(c, 1, &slot); }

// the host mirror of the list lengths is valid after this (uploads leave their read-backs in flight): every read-back ...
int sync_counts(sdm_ctx* c)
{
    if (c->cnt_done + 1 < c->cnt_next) {  // (they complete in order: one stream)
        HIP_TRY(hipEventSynchronize(c->cnt_ev[(c->cnt_next - 1) % sdm_ctx::CNT_RING]));
        c->cnt_done = c->cnt_next - 1;
    }
    return SDM_OK;
}
// ... or the ones of these slots only (stage_tables): a compute call does not wait for the step that is executing to drain,
// nor for a block that was uploaded behind it into other slots
int sync_counts_for(sdm_ctx* c, int n_a, const int* a, size_t n_b, const int* b)
{
    unsigned long long g = 0;
    for (int i = 0; i < n_a; i++) g = std::max(g, c->slot_cnt[(size_t)a[i]]);
    for (size_t i = 0; i < n_b; i++) g = std::max(g, c->slot_cnt[(size_t)b[i]]);
    if (g > c->cnt_done) {
        if (c->cnt_next - g >= (unsigned long long)sdm_ctx::CNT_RING) g = c->cnt_next - 1;  // its event was recycled: the newest
        HIP_TRY(hipEventSynchronize(c->cnt_ev[g % sdm_ctx::CNT_RING]));
        c->cnt_done = g;
    }
    return SDM_OK;
}

int check_slot(sdm_ctx* c, int slot, bool need_upload)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (slot < 0 || slot >= c->cfg.max_keyframes) return fail(SDM_EINVAL, "slot out of range");
    if (need_upload && !c->h_meta[slot].uploaded) return fail(SDM_ESTATE, "slot has no keyframe uploaded");
    return SDM_OK;
}

// the current set's pinned block is free for rewriting after this
int wait_tables(sdm_ctx* c)
{
    sdm_ctx::TableSet& s = c->sets[c->cur_set];
    if (s.pending) {
        HIP_TRY(hipEventSynchronize(s.free_ev));
        s.pending = false;
    }
    return SDM_OK;
}

// make set `si` the current one and carve the packed block for a call of (n_ref, np)
// packed layout (4-byte units): refs[n_ref] nbrs[np] rot[np] mind[n_ref] maxd[n_ref] | 8-byte: off[3*n_ref]
size_t select_set(sdm_ctx* c, int si, int n_ref, size_t np)
{
    sdm_ctx::TableSet& s = c->sets[si];
    c->cur_set = si;
    c->d_tab = s.d_tab;
    c->h_tab = s.h_tab;
    c->d_refs = s.d_refs;
    c->d_pairs = s.d_pairs;
    size_t words = (size_t)n_ref * 3 + np * 2;
    words = (words + 1) & ~(size_t)1;
    c->h_off = reinterpret_cast<long long*>(c->h_tab + words * 4);
    c->d_ref_slots = reinterpret_cast<int*>(c->d_tab);
    c->d_nbr_slots = c->d_ref_slots + n_ref;
    c->d_rot = reinterpret_cast<float*>(c->d_nbr_slots + np);
    c->d_mind = c->d_rot + np;
    c->d_maxd = c->d_mind + n_ref;
    c->d_off = reinterpret_cast<long long*>(c->d_tab + words * 4);
    return words;
}

// One dispatch stays at or below 2^31 work-items, half of HIP's documented limit of 2^32 - 1.  The limit is NOT enforced by
// the runtime on this stack (ROCm 7.2, HIP 7.2.26015): a launch whose gridDim.x * blockDim.x exceeds 2^32 returns hipSuccess
// and executes the grid MODULO 2^32 work-items (tools/ubench/big_grid.hip; profiles/r04_big_grid.txt).  That is what round 3
// ran into: K1 over 2048 keyframes of 1920x1080 in one launch is 2048 x 9880 workgroups x 256 = 5.18e9 work-items (the
// pixel lists grow along the sequence: 405 k entries at keyframe 0, 632 k at keyframe 2047), of which the first 0.89e9 ran
// -- every keyframe's list positions below 108 032 -- and the rest of every map kept what it held before
// (tools/debug/unsliced_holes.py).  Every launch whose grid grows with the number of reference keyframes is therefore issued
// in slices of reference keyframes: fn(first, count) launches one slice.
// (SDM_MAX_DISPATCH_LOG2: debugging knob, read once per process; values above 31 -- how the finding was reproduced --
// need a build with -DSDM_DEBUG_KNOBS: tools/debug/unsliced_*.py.)
template <typename F>
void for_ref_slices(int n_ref, long long blocks_per_ref, int threads, F&& fn)
{
    // read once; above 31 only in SDM_DEBUG_KNOBS builds (beyond 2^32 work-items the launch silently runs a partial grid)
    static const int lg = [] {
        int v = 31;
        if (const char* e = getenv("SDM_MAX_DISPATCH_LOG2")) {
#ifdef SDM_DEBUG_KNOBS
            v = std::max(20, std::min(40, atoi(e)));
#else
            v = std::max(20, std::min(31, atoi(e)));
#endif
        }
        return v;
    }();
    const long long max_blocks = (1ll << lg) / threads;
    const int per = (int)std::max<long long>(1, std::min<long long>(n_ref, max_blocks / std::max<long long>(blocks_per_ref, 1)));
    for (int first = 0; first < n_ref; first += per) fn(first, std::min(per, n_ref - first));
}

// HIP events around one stage's launches, on the stream the kernels run on
struct StageTimer {
    sdm_ctx* c;
    sdm_ctx::Span* sp = nullptr;
    StageTimer(sdm_ctx* ctx, int stage) : c(ctx)
    {
        if (!c->timing_on) return;
        if (c->spans_used == c->spans.size()) {
            if (c->spans.size() >= 65536) return;
            sdm_ctx::Span n{};
            if (hipEventCreate(&n.a) != hipSuccess || hipEventCreate(&n.b) != hipSuccess) return;
            c->spans.push_back(n);
        }
        sp = &c->spans[c->spans_used++];
        sp->stage = stage;
        (void)hipEventRecord(sp->a, c->stream);
    }
    ~StageTimer()
    {
        if (sp) (void)hipEventRecord(sp->b, c->stream);
    }
};

// Stage the tables of a call (slot lists, per-reference offsets, search constants) and build
// RefConst/PairConst on device.  need_consts: rot/mind/maxd matter (K1); otherwise any cached value is fine.
int stage_tables(sdm_ctx* c, int n_ref, const int* ref_slots, int n, const int* nbr_slots, const float* rot,
                 const float* mind, const float* maxd, bool need_consts = false)
{
    if (n_ref <= 0 || !ref_slots) return fail(SDM_EINVAL, "n_ref <= 0 or null ref_slots");
    if (n_ref > c->cap_refs) return fail(SDM_EINVAL, "n_ref exceeds max_keyframes");
    if (n < 0 || n > c->cfg.max_neighbours) return fail(SDM_EINVAL, "n exceeds max_neighbours");
    if (n > 0 && !nbr_slots) return fail(SDM_EINVAL, "null nbr_slots");
    for (int r = 0; r < n_ref; r++) {
        int rc = check_slot(c, ref_slots[r], true);
        if (rc) return rc;
        for (int j = 0; j < n; j++) {
            rc = check_slot(c, nbr_slots[r * n + j], true);
            if (rc) return rc;
        }
    }
    HIP_TRY(hipSetDevice(c->cfg.device));
    int rc;
    {  // lists follow lambdaG (sdm_set_params); so do the gradient-gate bit planes K1's scan reads of the NEIGHBOURS
        std::vector<int> stale;
        std::vector<char> seen((size_t)c->cfg.max_keyframes, 0);
        auto look = [&](int slot) {
            if (!(c->act_lambdaG[slot] == c->dprm.lambdaG) && !seen[slot]) {
                seen[slot] = 1;
                stale.push_back(slot);
            }
        };
        for (int r = 0; r < n_ref; r++) look(ref_slots[r]);
        if (need_consts)
            for (size_t i = 0; i < (size_t)n_ref * (size_t)n; i++) look(nbr_slots[i]);
        if (!stale.empty() && (rc = rebuild_lists(c, (int)stale.size(), stale.data()))) return rc;
    }
    static const bool dbg_st = getenv("SDM_DEBUG_STAGE_TIMING") != nullptr;
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_s0 = dbg_st ? now_ms() : 0.0;
    if ((rc = sync_counts_for(c, n_ref, ref_slots, (size_t)n_ref * (size_t)n, nbr_slots))) return rc;  // grids follow h_act_count
    if (dbg_st && now_ms() - t_s0 > 0.2)
        fprintf(stderr, "[sdm stage] waited %.3f ms for list lengths (slot %d: read-back %llu, done %llu, next %llu)\n",
                now_ms() - t_s0, ref_slots[0], c->slot_cnt[(size_t)ref_slots[0]], c->cnt_done, c->cnt_next);

    const size_t np = (size_t)n_ref * (size_t)n;
    auto matches = [&](const sdm_ctx::TableKey& k) {
        bool hit = k.valid && k.epoch == c->epoch && k.n_ref == n_ref &&
                   memcmp(k.refs.data(), ref_slots, sizeof(int) * n_ref) == 0;
        if (hit && n > 0) hit = (k.n == n) && memcmp(k.nbrs.data(), nbr_slots, sizeof(int) * np) == 0;
        if (hit && need_consts) {
            hit = k.has_consts && memcmp(k.mind.data(), mind, sizeof(float) * n_ref) == 0 &&
                  memcmp(k.maxd.data(), maxd, sizeof(float) * n_ref) == 0;
            if (hit) {
                if (rot)
                    hit = memcmp(k.rot.data(), rot, sizeof(float) * np) == 0;
                else
                    for (size_t i = 0; i < np && hit; i++) hit = (k.rot[i] == 0.0f);
            }
        }
        return hit;
    };
    for (int si = 0; si < sdm_ctx::TABLE_SETS; si++) {
        if (!matches(c->sets[si].key)) continue;
        // the carve-out follows the sizes the set was staged with (a key with n > 0 serves an n == 0 call too)
        select_set(c, si, n_ref, (size_t)n_ref * (size_t)c->sets[si].key.n);
        c->sets[si].last_use = ++c->use_tick;
        return SDM_OK;
    }
    int victim = 0;  // an empty set, else the least recently used one
    for (int si = 0; si < sdm_ctx::TABLE_SETS; si++) {
        if (!c->sets[si].key.valid) {
            victim = si;
            break;
        }
        if (c->sets[si].last_use < c->sets[victim].last_use) victim = si;
    }
    const size_t words = select_set(c, victim, n_ref, np);
    c->table_stagings++;
    const double t_w0 = dbg_st ? now_ms() : 0.0;
    if ((rc = wait_tables(c))) return rc;  // only if that set was staged within the last few calls
    if (dbg_st && now_ms() - t_w0 > 0.2) fprintf(stderr, "[sdm stage] waited %.3f ms for table set %d\n", now_ms() - t_w0, victim);
    sdm_ctx::TableKey& k = c->sets[victim].key;
    k.valid = false;
    const size_t bytes = words * 4 + sizeof(long long) * 3 * (size_t)n_ref;
    if (bytes > c->tab_bytes) return fail(SDM_EINVAL, "table staging overflow");
    int* h_refs = reinterpret_cast<int*>(c->h_tab);
    int* h_nbrs = h_refs + n_ref;
    float* h_rot = reinterpret_cast<float*>(h_nbrs + np);
    float* h_mind = h_rot + np;
    float* h_maxd = h_mind + n_ref;
    memcpy(h_refs, ref_slots, sizeof(int) * n_ref);
    if (np) memcpy(h_nbrs, nbr_slots, sizeof(int) * np);
    for (size_t i = 0; i < np; i++) h_rot[i] = rot ? rot[i] : 0.0f;
    const int cap = c->cfg.batch_capacity;
    for (int r = 0; r < n_ref; r++) {
        h_mind[r] = mind ? mind[r] : 0.f;
        h_maxd[r] = maxd ? maxd[r] : 0.f;
        c->h_off[r] = (long long)ref_slots[r] * c->P;              // depth-pool offset
        c->h_off[n_ref + r] = (long long)(r % cap) * c->P;         // scratch offset
        c->h_off[2 * n_ref + r] = (long long)ref_slots[r] * c->P * 4;  // record offset in floats
    }
    HIP_TRY(hipMemcpyAsync(c->d_tab, c->h_tab, bytes, hipMemcpyHostToDevice, c->stream));
    if (n > 0) {
        hipLaunchKernelGGL(k_pair_setup, dim3(blocks_for((long long)np)), dim3(BLOCK), 0, c->stream, c->d_meta,
                           c->d_ref_slots, c->d_nbr_slots, c->d_rot, c->d_mind, c->d_maxd, c->d_act_count, c->d_theta_bad,
                           n_ref, n, c->W, c->H, c->d_refs, c->d_pairs);
    } else {
        hipLaunchKernelGGL(k_ref_setup, dim3(blocks_for(n_ref)), dim3(BLOCK), 0, c->stream, c->d_meta, c->d_ref_slots,
                           c->d_act_count, n_ref, c->d_refs);
    }
    HIP_TRY(hipGetLastError());
    k.valid = true;
    k.has_consts = (n > 0) && mind && maxd;
    k.n_ref = n_ref;
    k.n = n;
    k.epoch = c->epoch;
    k.refs.assign(ref_slots, ref_slots + n_ref);
    k.nbrs.assign(nbr_slots, nbr_slots + np);
    k.rot.assign(h_rot, h_rot + np);
    k.mind.assign(h_mind, h_mind + n_ref);
    k.maxd.assign(h_maxd, h_maxd + n_ref);
    k.long_ranges = false;
    if (k.has_consts) {
        // The search range of PM.cc:877-910 at the principal point (xp = (0, 0, 1)), averaged over the call's pairs, restated on
        // the host: a hint that selects K1's instantiation (the device's own per-pair bit and the waves' range lengths decide
        // each scan).  The mask-scan instantiation costs the batched scan 6 % (measured), so it only runs where the long
        // ranges are the bulk of the work.
        double sum = 0.0;
        for (size_t i = 0; i < np; i++) {
            const KfMeta& m1 = c->h_meta[ref_slots[i / (size_t)n]];
            const KfMeta& m2 = c->h_meta[nbr_slots[i]];
            float F[9], R21[9], t21[3];
            pair_geometry(m1, m2, F, R21, t21);
            const float d0 = h_mind[i / (size_t)n], d1 = h_maxd[i / (size_t)n];
            float u0 = m1.fx * (R21[2] * d0 + t21[0]) / (R21[8] * d0 + t21[2]) + m1.cx;
            float u1 = m1.fx * (R21[2] * d1 + t21[0]) / (R21[8] * d1 + t21[2]) + m1.cx;
            const float cols = (float)c->W;
            u0 = u0 < 0 ? 0 : (u0 > cols ? cols : u0);
            u1 = u1 < 0 ? 0 : (u1 > cols ? cols : u1);
            const float len = std::fabs(u1 - u0);
            sum += (len == len) ? (double)len : (double)cols;  // (NaN ends: cannot tell -- count the pair as long)
        }
        k.long_ranges = np > 0 && sum / (double)np >= (double)MASK_CALL_MEAN_L;
    }
    HIP_TRY(hipEventRecord(c->sets[victim].free_ev, c->stream));  // the pinned block may be rewritten after this point
    c->sets[victim].pending = true;
    c->sets[victim].last_use = ++c->use_tick;
    return SDM_OK;
}

int tables_staged(sdm_ctx* c)
{
    (void)c;  // the staging block is released by the event recorded in stage_tables
    return SDM_OK;
}

// stream-ordered, no copy: the 80-byte record travels as a kernel argument.  keep_istd: leave the device's
// I_stddev alone (it was derived on the device by the pre-pass; the host holds no mirror of it)
int push_meta(sdm_ctx* c, int slot, bool keep_istd)
{
    c->epoch++;
    hipLaunchKernelGGL(k_set_meta, dim3(1), dim3(1), 0, c->stream, c->d_meta + slot, c->h_meta[slot], keep_istd ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return SDM_OK;
}

// a keyframe uploaded into a slot starts with fresh (zero) maps and no stage flags: the host-side bookkeeping ...
void reset_slot_state(sdm_ctx* c, int slot)
{
    c->epoch++;
    c->has_depth[slot] = 0;
    c->has_chk[slot] = 0;
    c->recon_lambdaG[slot] = std::nanf("");
    c->act_lambdaG[slot] = std::nanf("");
    c->chk_sparse[slot] = 1;
    c->xyz_sparse[slot] = 1;
}
// ... and the planes (the image paths clear them inside k_prepass_batch instead)
int reset_slot(sdm_ctx* c, int slot)
{
    reset_slot_state(c, slot);
    HIP_TRY(hipMemsetAsync(c->pool + (long long)slot * c->P, 0, sizeof(float2) * c->P, c->stream));
    HIP_TRY(hipMemsetAsync(c->chk + (long long)slot * c->P, 0, sizeof(float) * c->P, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_theta_bad + slot, 0, sizeof(int), c->stream));  // k_pack sets it again if need be
    if (c->xyz) HIP_TRY(hipMemsetAsync(c->xyz + (long long)slot * c->P * 3, 0, sizeof(float) * 3 * c->P, c->stream));
    return SDM_OK;
}

void fill_meta(KfMeta& m, const float K[4], const float Tcw[12])
{
    m.fx = K[0];
    m.fy = K[1];
    m.cx = K[2];
    m.cy = K[3];
    memcpy(m.Tcw, Tcw, sizeof(float) * 12);
}

int pack_staged(sdm_ctx* c, int slot)
{
    hipLaunchKernelGGL(k_pack, dim3(blocks_for(c->P)), dim3(BLOCK), 0, c->stream, c->d_im, c->d_grad, c->d_theta,
                       c->W, c->H, c->rec + (long long)slot * c->P, c->d_theta_bad + slot);
    HIP_TRY(hipGetLastError());
    return SDM_OK;
}

template <typename T>
int dev_alloc(T** p, size_t count)
{
    HIP_TRY(hipMalloc((void**)p, sizeof(T) * std::max<size_t>(count, 1)));
    return SDM_OK;
}
template <typename T>
int host_alloc(T** p, size_t count)
{
    HIP_TRY(hipHostMalloc((void**)p, sizeof(T) * std::max<size_t>(count, 1), hipHostMallocDefault));
    return SDM_OK;
}

// chunk buffers [have, want) of the batched ingest (device image block, pinned staging ring, item tables, two events each)
int alloc_ingest_bufs(sdm_ctx* c, int want)
{
    int rc;
    for (int b = 0; b < want; b++) {
        sdm_ctx::IngestBuf& B = c->ing[b];
        if (B.d_img) continue;
        if ((rc = dev_alloc(&B.d_img, (size_t)c->ing_cap * c->P)) || (rc = host_alloc(&B.h_ring, (size_t)c->ing_cap * c->P)) ||
            (rc = dev_alloc(&B.d_items, (size_t)c->ing_cap)) || (rc = host_alloc(&B.h_items, (size_t)c->ing_cap)))
            return rc;
        if (hipEventCreateWithFlags(&B.copied, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&B.consumed, hipEventDisableTiming) != hipSuccess)
            return fail(SDM_EHIP, "hipEventCreate failed");
    }
    return SDM_OK;
}

}  // namespace

#include "sdm_comm.h"

extern "C" {

void sdm_default_params(sdm_params* p)
{
    p->lambdaG = 8.0f;
    p->lambdaL = 80.0f;
    p->lambdaTheta = 45.0f;
    p->lambdaN = 3;
    p->theta_var = 0.23;
}

void sdm_default_config(sdm_config* c)
{
    memset(c, 0, sizeof(*c));
    c->W = 640;
    c->H = 480;
    c->max_keyframes = 16;
    c->max_neighbours = 7;  // covisN, PM.h:38
    c->with_pointset = 1;
}

size_t sdm_depth_pool_bytes(int W, int H, int max_keyframes)
{
    return sizeof(float2) * (size_t)W * (size_t)H * (size_t)max_keyframes;
}

const char* sdm_last_error(void) { return g_err.c_str(); }

int sdm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sdm_create(sdm_ctx** out, const sdm_config* cfg)
{
    if (!out || !cfg) return fail(SDM_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->W < 8 || cfg->H < 8 || cfg->W > 16384 || cfg->H > 16384) return fail(SDM_EINVAL, "bad image size");
    // kernels address records with 32-bit byte offsets inside a plane (16 B per pixel)
    if ((long long)cfg->W * cfg->H > (1ll << 27)) return fail(SDM_EINVAL, "image too large (more than 2^27 pixels)");
    if (cfg->max_keyframes < 1) return fail(SDM_EINVAL, "max_keyframes < 1");
    if (cfg->max_neighbours < 1 || cfg->max_neighbours > SDM_MAX_NEIGHBOURS)
        return fail(SDM_EINVAL, "max_neighbours out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SDM_ENODEV, "no HIP device visible: this engine has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(SDM_EINVAL, "device ordinal out of range");
    HIP_TRY(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));

    sdm_ctx* c = new sdm_ctx();
    c->cfg = *cfg;
    c->arch = prop.gcnArchName;
    c->W = cfg->W;
    c->H = cfg->H;
    c->P = (long long)cfg->W * cfg->H;
    c->geom = make_geom(cfg->W, cfg->H);
    if (c->cfg.batch_capacity <= 0) c->cfg.batch_capacity = std::min(cfg->max_keyframes, 64);
    c->cfg.batch_capacity = std::max(2, std::min(c->cfg.batch_capacity, std::max(cfg->max_keyframes, 2)));
    sdm_default_params(&c->prm);
    set_dev_params(c);
    const int K = cfg->max_keyframes;
    c->cap_refs = K;
    c->h_meta.assign(K, KfMeta{});
    c->has_depth.assign(K, 0);
    c->has_chk.assign(K, 0);
    c->act_lambdaG.assign(K, std::nanf(""));
    c->recon_lambdaG.assign(K, std::nanf(""));
    c->chk_sparse.assign(K, 1);  // planes start zeroed
    c->xyz_sparse.assign(K, 1);
    c->slot_cnt.assign(K, 0);

    int rc = SDM_OK;
    auto bail = [&](int code) {
        std::string keep = g_err;
        sdm_destroy(c);
        g_err = keep;
        return code;
    };
    if (cfg->stream) {
        c->stream = (hipStream_t)cfg->stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
            return bail(fail(SDM_EHIP, "hipStreamCreate failed"));
        c->own_stream = true;
    }
    if ((rc = dev_alloc(&c->rec, (size_t)c->P * K))) return bail(rc);
    if (cfg->ext_depth_pool) {
        c->pool = (float2*)cfg->ext_depth_pool;
    } else {
        if ((rc = dev_alloc(&c->pool, (size_t)c->P * K))) return bail(rc);
        c->own_pool = true;
    }
    if ((rc = dev_alloc(&c->scratch, (size_t)c->P * c->cfg.batch_capacity))) return bail(rc);
    if ((rc = dev_alloc(&c->chk, (size_t)c->P * K))) return bail(rc);

    if (cfg->with_pointset)
        if ((rc = dev_alloc(&c->xyz, (size_t)c->P * 3 * K))) return bail(rc);
    if ((rc = dev_alloc(&c->d_meta, (size_t)K))) return bail(rc);
    if ((rc = dev_alloc(&c->d_act, (size_t)c->P * K))) return bail(rc);
    if ((rc = dev_alloc(&c->d_act_count, (size_t)K))) return bail(rc);
    if ((rc = dev_alloc(&c->d_theta_bad, (size_t)K))) return bail(rc);
    {
        // a quarter of all pixels, at most 2^20 entries of 16 + 8 n bytes; when it fills up the remaining workgroups
        // count their open pixels in place (slower, same result)
        long long cap = std::min<long long>(1ll << 20, std::max<long long>(K1_PX, c->P * K / 4));
        if (const char* e = getenv("SDM_K4_PAD")) c->k4_lds_pad = (unsigned)atoi(e);
        if (const char* e = getenv("SDM_OPEN_QUOTA")) c->open_quota = (unsigned)atoll(e);  // tests: 0 defers every open pixel
        if (const char* e = getenv("SDM_OPEN_INPLACE")) c->open_inplace_min = (unsigned)atoll(e);  // tests / A-B: 65 = never
        if (const char* e = getenv("SDM_OPEN_GRID")) c->open_grid = (unsigned)std::max(1ll, atoll(e));
        if (const char* e = getenv("SDM_OPEN_CAPACITY"))  // tests: a tiny list forces the in-place fallback
            cap = std::max<long long>(K1_PX, std::min<long long>(cap, atoll(e)));
        c->open_capacity = (unsigned)(cap / K1_PX * K1_PX);
        if ((rc = dev_alloc(&c->d_open_ctr, 8)) || (rc = dev_alloc(&c->d_open_pix, (size_t)c->open_capacity)) ||
            (rc = dev_alloc(&c->d_open_vm, (size_t)c->open_capacity)) ||
            (rc = dev_alloc(&c->d_open_hyp, (size_t)c->open_capacity * cfg->max_neighbours)))
            return bail(rc);
        const unsigned init[8] = {0u, 0xFFFFFFFFu, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, 0u};
        if (hipMemcpy(c->d_open_ctr, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(SDM_EHIP, "open-list counter initialisation failed"));
        long long gcap = std::min<long long>(1ll << 18, c->P);
        if (const char* e = getenv("SDM_GROW_CAPACITY")) gcap = std::max<long long>(1, std::min<long long>(gcap, atoll(e)));  // tests
        c->grow_capacity = (unsigned)gcap;
        if ((rc = dev_alloc(&c->d_grow_ctr, 2)) || (rc = dev_alloc(&c->d_grow_pix, (size_t)gcap)) ||
            (rc = dev_alloc(&c->d_grow_val, (size_t)gcap)))
            return bail(rc);
        if (hipMemset(c->d_grow_ctr, 0, 2 * sizeof(unsigned)) != hipSuccess)
            return bail(fail(SDM_EHIP, "grow-list counter initialisation failed"));
    }
    {
        // ingest chunks: as many keyframes as keep a chunk's gray images within 32 MB (64 at 640x480, 16 at 1920x1080)
        c->nseg = c->H * c->geom.tiles_x;
        c->ing_cap = (int)std::max<long long>(1, std::min<long long>(std::min(64, K), ((long long)32 << 20) / c->P));
        c->src_bytes = (size_t)std::max<long long>((long long)c->ing_cap * c->P, 4 * c->P);
        {
            // the upload stream (H2D copies of batch uploads) at the highest priority the device offers
            int lo_p = 0, hi_p = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
            if (hipStreamCreateWithPriority(&c->up_stream, hipStreamNonBlocking, hi_p) != hipSuccess)
                return bail(fail(SDM_EHIP, "hipStreamCreate failed"));
        }
        for (int i = 0; i < sdm_ctx::CNT_RING; i++)
            if (hipEventCreateWithFlags(&c->cnt_ev[i], hipEventDisableTiming) != hipSuccess)
                return bail(fail(SDM_EHIP, "hipEventCreate failed"));
        if ((rc = alloc_ingest_bufs(c, c->ing_bufs))) return bail(rc);
        const size_t gcap = (size_t)sdm_ctx::ING_GROUP * c->ing_cap;
        if ((rc = dev_alloc(&c->d_part, gcap * c->geom.ntiles * PART_WORDS)) || (rc = dev_alloc(&c->d_seg_mask, gcap * c->nseg)) ||
            (rc = dev_alloc(&c->d_seg_off, gcap * c->nseg)) || (rc = dev_alloc(&c->d_gitems, gcap)) ||
            (rc = dev_alloc(&c->d_act_hash, (size_t)K)))
            return bail(rc);
        c->mrow = 2 * (c->geom.tiles_x + 1);
        if ((rc = dev_alloc(&c->d_gmask, (size_t)K * c->H * MASK_PLANES * c->mrow))) return bail(rc);
        if (const char* e = getenv("SDM_SCAN_MODE")) c->scan_mode = std::max(0, std::min(2, atoi(e)));  // tests / A-B
        set_dev_params(c);
    }
    if ((rc = dev_alloc(&c->d_im, (size_t)c->P))) return bail(rc);
    if ((rc = dev_alloc(&c->d_grad, (size_t)c->P))) return bail(rc);
    if ((rc = dev_alloc(&c->d_theta, (size_t)c->P))) return bail(rc);
    if ((rc = dev_alloc(&c->d_small, 16))) return bail(rc);
    if ((rc = dev_alloc(&c->d_stats, STATS_WORDS))) return bail(rc);
    const size_t np = (size_t)K * cfg->max_neighbours;
    c->tab_bytes = 4 * ((size_t)K * 3 + np * 2 + 2) + sizeof(long long) * 3 * (size_t)K;
    for (int si = 0; si < sdm_ctx::TABLE_SETS; si++) {
        sdm_ctx::TableSet& s = c->sets[si];
        if ((rc = dev_alloc(&s.d_tab, c->tab_bytes)) || (rc = host_alloc(&s.h_tab, c->tab_bytes))) return bail(rc);
        if ((rc = dev_alloc(&s.d_refs, K)) || (rc = dev_alloc(&s.d_pairs, np))) return bail(rc);
        if (hipEventCreateWithFlags(&s.free_ev, hipEventDisableTiming) != hipSuccess)
            return bail(fail(SDM_EHIP, "hipEventCreate failed"));
    }
    select_set(c, 0, 1, 0);
    if ((rc = host_alloc(&c->h_f2, (size_t)c->P))) return bail(rc);
    if ((rc = host_alloc(&c->h_act_count, (size_t)K))) return bail(rc);
    memset(c->h_act_count, 0, sizeof(int) * (size_t)K);

    // zero-initialised maps, as a fresh KeyFrame's depth_map_/depth_sigma_/SemiDensePointSets_
    if (hipMemsetAsync(c->rec, 0, sizeof(float4) * c->P * K, c->stream) != hipSuccess ||
        hipMemsetAsync(c->pool, 0, sizeof(float2) * c->P * K, c->stream) != hipSuccess ||
        hipMemsetAsync(c->chk, 0, sizeof(float) * c->P * K, c->stream) != hipSuccess ||
        hipMemsetAsync(c->d_meta, 0, sizeof(KfMeta) * K, c->stream) != hipSuccess ||
        hipMemsetAsync(c->d_act_count, 0, sizeof(int) * K, c->stream) != hipSuccess ||
        hipMemsetAsync(c->d_theta_bad, 0, sizeof(int) * K, c->stream) != hipSuccess ||
        hipMemsetAsync(c->d_act_hash, 0, sizeof(unsigned long long) * K, c->stream) != hipSuccess ||
        hipMemsetAsync(c->d_gmask, 0, sizeof(unsigned) * (size_t)K * c->H * MASK_PLANES * c->mrow, c->stream) != hipSuccess ||
        hipMemsetAsync(c->d_stats, 0, sizeof(unsigned long long) * STATS_WORDS, c->stream) != hipSuccess ||
        (c->xyz && hipMemsetAsync(c->xyz, 0, sizeof(float) * 3 * c->P * K, c->stream) != hipSuccess) ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return bail(fail(SDM_EHIP, "initial memset failed"));

    // K1's hypothesis columns can need more than the default 64 KB of dynamic LDS
    const int max_lds = (int)k1_lds_bytes(cfg->max_neighbours);
    if (hipFuncSetAttribute((const void*)k_search_fuse<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_search_fuse<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_search_fuse<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_search_fuse<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess)
        return bail(fail(SDM_EHIP, "hipFuncSetAttribute(max dynamic LDS) failed"));
    *out = c;
    return SDM_OK;
}

void sdm_destroy(sdm_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    comm_release(c);
    (void)hipFree(c->rec);
    if (c->own_pool) (void)hipFree(c->pool);
    (void)hipFree(c->scratch);
    (void)hipFree(c->chk);

    (void)hipFree(c->xyz);
    (void)hipFree(c->d_meta);
    (void)hipFree(c->d_act);
    (void)hipFree(c->d_act_count);
    (void)hipFree(c->d_theta_bad);
    (void)hipFree(c->d_open_ctr);
    (void)hipFree(c->d_open_pix);
    (void)hipFree(c->d_open_vm);
    (void)hipFree(c->d_open_hyp);
    (void)hipFree(c->d_grow_ctr);
    (void)hipFree(c->d_grow_pix);
    (void)hipFree(c->d_grow_val);
    if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
    for (int b = 0; b < sdm_ctx::ING_BUFS_MAX; b++) {
        sdm_ctx::IngestBuf& B = c->ing[b];
        (void)hipFree(B.d_img);
        (void)hipFree(B.d_src);
        (void)hipFree(B.d_items);
        (void)hipHostFree(B.h_ring);
        (void)hipHostFree(B.h_src);
        (void)hipHostFree(B.h_items);
        if (B.copied) (void)hipEventDestroy(B.copied);
        if (B.consumed) (void)hipEventDestroy(B.consumed);
    }
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    for (int i = 0; i < sdm_ctx::CNT_RING; i++)
        if (c->cnt_ev[i]) (void)hipEventDestroy(c->cnt_ev[i]);
    (void)hipFree(c->d_part);
    (void)hipFree(c->d_seg_mask);
    (void)hipFree(c->d_seg_off);
    (void)hipFree(c->d_gitems);
    (void)hipFree(c->d_act_hash);
    (void)hipFree(c->d_gmask);
    (void)hipFree(c->d_im);
    (void)hipFree(c->d_grad);
    (void)hipFree(c->d_theta);
    (void)hipFree(c->d_small);
    (void)hipFree(c->d_stats);
    for (int si = 0; si < sdm_ctx::TABLE_SETS; si++) {
        sdm_ctx::TableSet& s = c->sets[si];
        (void)hipFree(s.d_tab);
        (void)hipFree(s.d_refs);
        (void)hipFree(s.d_pairs);
        (void)hipHostFree(s.h_tab);
        if (s.free_ev) (void)hipEventDestroy(s.free_ev);
    }
    (void)hipHostFree(c->h_f2);
    (void)hipHostFree(c->h_act_count);
    for (auto& sp : c->spans) {
        (void)hipEventDestroy(sp.a);
        (void)hipEventDestroy(sp.b);
    }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int sdm_set_params(sdm_ctx* c, const sdm_params* p)
{
    if (!c || !p) return fail(SDM_EINVAL, "null argument");
    if (!(p->theta_var > 0) || p->lambdaN < 0) return fail(SDM_EINVAL, "bad parameter value");
    c->prm = *p;
    set_dev_params(c);
    return validate_params(c);
}

int sdm_get_params(sdm_ctx* c, sdm_params* out)
{
    if (!c || !out) return fail(SDM_EINVAL, "null argument");
    *out = c->prm;
    return SDM_OK;
}

int sdm_set_stream(sdm_ctx* c, void* s)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->cnt_done = c->cnt_next - 1;  // everything queued on the old stream has finished
    if (c->own_stream) {
        (void)hipStreamDestroy(c->stream);
        c->own_stream = false;
    }
    c->stream = (hipStream_t)s;
    return SDM_OK;
}

int sdm_synchronize(sdm_ctx* c)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SDM_OK;
}

const char* sdm_device_arch(sdm_ctx* c) { return c ? c->arch.c_str() : ""; }

// ---- keyframe inputs -----------------------------------------------------------------------------------------
int sdm_upload_keyframe(sdm_ctx* c, int slot, const uint8_t* im, const float* grad, const float* theta, float I_stddev,
                        const float K[4], const float Tcw[12])
{
    int rc = check_slot(c, slot, false);
    if (rc) return rc;
    if (!im || !grad || !theta || !K || !Tcw) return fail(SDM_EINVAL, "null input plane");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if ((rc = reset_slot(c, slot))) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_im, im, (size_t)c->P, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_grad, grad, sizeof(float) * c->P, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_theta, theta, sizeof(float) * c->P, hipMemcpyHostToDevice, c->stream));
    if ((rc = pack_staged(c, slot))) return rc;
    KfMeta& m = c->h_meta[slot];
    fill_meta(m, K, Tcw);
    m.I_stddev = I_stddev;
    m.uploaded = 1;
    if ((rc = push_meta(c, slot, false))) return rc;
    if ((rc = build_active(c, slot))) return rc;
    return sync_counts(c);  // the caller's (pageable) planes are released on return
}

// ---- image ingest, batched (sdm_ingest.h) ------------------------------------------------------------------------------------
namespace {

// Pinned blocks handed out by sdm_host_alloc, [base, base + size): images inside one are read in place by the copy engine.
// (Asking the driver about every image pointer instead -- hipPointerGetAttributes -- costs a system call that now and then
// takes milliseconds for ordinary memory; memory the application pinned by other means is simply staged like pageable memory.)
std::mutex g_pinned_mutex;
std::map<const uint8_t*, size_t> g_pinned_blocks;
bool host_pinned(const void* p, size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    const uint8_t* q = (const uint8_t*)p;
    auto it = g_pinned_blocks.upper_bound(q);
    if (it == g_pinned_blocks.begin()) return false;
    --it;
    return q >= it->first && q + bytes <= it->first + it->second;
}

// Staging copies of pageable images into the pinned ring, chunk after chunk, shared between the calling thread and a few
// helper threads that live for ONE batch call (started only when the batch is worth their start-up): chunk k's plan is
// posted once its ring half is free, every thread copies the images i = t (mod nt) of it, the caller waits for the helpers
// of that chunk and queues its H2D copy while they wait for the next plan.
struct Stager {
    struct Plan {
        uint8_t* dst = nullptr;
        const uint8_t* const* src = nullptr;
        int m = 0;
        size_t bytes = 0;
    };
    int nt = 1;  // threads that copy: the caller + the helpers that actually started
    std::vector<std::thread> th;
    std::vector<Plan> plans;
    std::vector<std::atomic<int> > done;
    std::mutex mu;
    std::condition_variable cv;  // helpers sleep here between chunks (the caller may sit in a HIP wait for a long time)
    int posted = 0;              // guarded by mu
    bool quit = false;           // guarded by mu

    // never throws: a helper that cannot be started (EAGAIN under a thread limit, out of memory) is simply not there, and the
    // caller stages alone if none can -- this runs inside a C-ABI entry point
    Stager(int threads, int chunks) noexcept
    {
        try {
            plans.resize((size_t)chunks);
            done = std::vector<std::atomic<int> >((size_t)chunks);
            for (auto& d : done) d.store(0);
            th.reserve((size_t)std::max(0, threads - 1));
            for (int t = 1; t < threads; t++)
                th.emplace_back([this, t] {
                    for (size_t k = 0; k < plans.size(); k++) {
                        {
                            std::unique_lock<std::mutex> lock(mu);
                            cv.wait(lock, [&] { return quit || posted > (int)k; });
                            if (quit) return;
                        }
                        share(plans[k], t);  // nt is final before the first plan is posted
                        done[k].fetch_add(1, std::memory_order_release);
                    }
                });
        } catch (...) {
            // keep the helpers that did start (their indices are 1 .. th.size(): the loop stops at the first failure)
        }
        nt = 1 + (int)th.size();
    }
    bool usable() const { return !plans.empty() && done.size() == plans.size(); }
    void share(const Plan& p, int t) const
    {
        for (int i = t; i < p.m; i += nt) memcpy(p.dst + (size_t)i * p.bytes, p.src[i], p.bytes);
    }
    // chunk k (chunks are staged in order): returns when all of it is in the ring
    void stage(int k, uint8_t* dst, const uint8_t* const* src, int m, size_t bytes)
    {
        Plan p;
        p.dst = dst;
        p.src = src;
        p.m = m;
        p.bytes = bytes;
        if (!usable() || nt == 1) {  // no helpers (or the tables could not be allocated): this thread copies everything
            for (int i = 0; i < m; i++) memcpy(dst + (size_t)i * bytes, src[i], bytes);
            return;
        }
        plans[(size_t)k] = p;
        {
            std::lock_guard<std::mutex> lock(mu);
            posted = k + 1;
        }
        cv.notify_all();
        share(p, 0);
        while (done[(size_t)k].load(std::memory_order_acquire) < nt - 1) std::this_thread::yield();  // (they copy as long as we did)
    }
    ~Stager()
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            quit = true;
        }
        cv.notify_all();
        for (auto& x : th)
            if (x.joinable()) x.join();
    }
};

int check_batch_slots(sdm_ctx* c, int n, const int* slots)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (n < 0 || (n > 0 && !slots)) return fail(SDM_EINVAL, "bad slot list");
    std::vector<char> seen((size_t)c->cfg.max_keyframes, 0);
    for (int i = 0; i < n; i++) {
        int rc = check_slot(c, slots[i], false);
        if (rc) return rc;
        if (seen[slots[i]]) return fail(SDM_EINVAL, "duplicate slot in one upload batch");
        seen[slots[i]] = 1;
    }
    return SDM_OK;
}

int ensure_src_buffers(sdm_ctx* c)
{
    if (c->ing[0].d_src && c->ing[c->ing_bufs - 1].d_src) return SDM_OK;
    int rc;
    for (int b = 0; b < c->ing_bufs; b++) {
        if (c->ing[b].d_src) continue;
        if ((rc = dev_alloc(&c->ing[b].d_src, c->src_bytes))) return rc;
        if ((rc = host_alloc(&c->ing[b].h_src, c->src_bytes))) return rc;
    }
    return SDM_OK;
}

// n keyframes from host memory (gray: q == nullptr, P bytes each; else interleaved frames of P * q->channels bytes) or
// from device memory (on_device: gray only).  Chunk k+1's copies run on the upload stream while chunk k's pre-pass runs
// on the compute stream.  Pinned host images are copied from where they lie; pageable ones go through the pinned ring.
int ingest_images_impl(sdm_ctx* c, int n, const int* slots, const uint8_t* const* images, bool on_device, const IngestParams* q,
                  const float* K, const float* Tcw)
{
    static const bool dbg = getenv("SDM_DEBUG_INGEST_TIMING") != nullptr;
    static const double dbg_ms = dbg && atof(getenv("SDM_DEBUG_INGEST_TIMING")) > 0 ? atof(getenv("SDM_DEBUG_INGEST_TIMING")) : 2.0;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_entry = dbg ? now() : 0.0;
    int rc = check_batch_slots(c, n, slots);
    if (rc) return rc;
    if (n == 0) return SDM_OK;
    if (!images || !K || !Tcw) return fail(SDM_EINVAL, "null input");
    for (int i = 0; i < n; i++)
        if (!images[i]) return fail(SDM_EINVAL, "null image");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (q && (rc = ensure_src_buffers(c))) return rc;
    const size_t bytes = (size_t)c->P * (q ? q->channels : 1);
    int cap = q ? (int)std::max<size_t>(1, std::min<size_t>((size_t)c->ing_cap, c->src_bytes / bytes)) : c->ing_cap;
    // a batch goes through in (at least) four chunks, so that the host's staging copy and the H2D copy of one chunk run
    // while the device works on the previous one
    cap = std::max(1, std::min(cap, std::max(4, (n + 3) / 4)));
    bool direct = false;  // some H2D copy reads the caller's (pinned) memory
    int last = -1;
    // pageable sources are staged by this thread and, for batches of 2 MB and more, up to three helpers
    std::vector<char> pinned((size_t)n, 0);
    size_t pageable_bytes = 0;
    for (int i = 0; i < n && !on_device; i++) {
        pinned[i] = host_pinned(images[i], bytes) ? 1 : 0;
        if (!pinned[i]) pageable_bytes += bytes;
    }
    const int hw = (int)std::thread::hardware_concurrency();
    const int n_chunks = (n + cap - 1) / cap;
    static const int max_stagers = [] {  // A/B knob, read once per process (default 4: tools/debug/upload_probe.py)
        const char* e = getenv("SDM_STAGER_THREADS");
        return e ? std::max(1, std::min(16, atoi(e))) : 4;
    }();
    Stager stager((pageable_bytes >= ((size_t)2 << 20) && hw >= 4) ? std::min(max_stagers, hw / 2) : 1, n_chunks);
    // one chunk (a single new keyframe, the online use): nothing to overlap with, so its copies stay on the compute stream
    // and no cross-stream hand-over is paid; more chunks: copies on the upload stream, kernels behind an event
    const hipStream_t cs = n_chunks > 1 ? c->up_stream : c->stream;
    double tdbg[5] = {0, 0, 0, 0, 0};
    const double t_loop = dbg ? now() : 0.0;
    // An error in the middle of a batch: copies that read the caller's pinned images may still be in flight -- they are awaited
    // before the call returns (the promise "every caller buffer is free on return" holds on the error path too).  Slots of
    // chunks that were already launched hold their new keyframes; the slots of the failing and the later chunks are reset
    // (no keyframe): the call reports failure for the batch, sdm_upload_* can simply be repeated.
    int failed_from = n;
    int group_kfs = 0, group_chunks = 0;  // chunks whose first kernels are queued and whose group is not finished yet
    int group_first = 0;                  // index of that group's first keyframe
    // streaming ingest: the copies run ahead of the compute stream anyway, so the chunks of a group share ONE pre-pass launch
    // behind ONE wait for the group's last copy (per-chunk launches start the first chunk's kernel a copy earlier, which only
    // matters when the compute stream has nothing else to do)
    const bool merge = c->ingest_overlap && n_chunks > 1 && !q && !on_device && cs != c->stream;
    int merged_bufs[CHUNK_TABLES] = {0, 0, 0, 0};
    auto finish_group = [&]() -> int {
        const int g = group_kfs, nch = group_chunks;
        group_kfs = group_chunks = 0;
        int r = SDM_OK;
        if (merge && nch > 0) r = ingest_launch_chunks_merged(c, nch, merged_bufs, cap, g);
        if (!r) r = ingest_launch_group(c, g, true);
        return r;
    };
    auto bail = [&](int code) {
        const std::string keep = g_err;
        if (direct) (void)hipStreamSynchronize(cs);
        (void)finish_group();  // the chunks before the failing one keep their keyframes
        (void)counts_queued(c, failed_from, slots);
        for (int i = failed_from; i < n; i++) {
            reset_slot_state(c, slots[i]);
            c->h_meta[slots[i]].uploaded = 0;
        }
        g_err = keep;
        return code;
    };
    for (int i0 = 0, chunk = 0; i0 < n; i0 += cap, chunk++) {
        const int m = std::min(cap, n - i0);
        failed_from = i0;
        const int b = c->ing_next;
        c->ing_next = (c->ing_next + 1) % c->ing_bufs;
        sdm_ctx::IngestBuf& B = c->ing[b];
        if (dbg) tdbg[0] = now();
        if ((rc = ingest_acquire(c, b, cs))) return bail(rc);
        if (dbg) tdbg[1] = now();
        uint8_t* d_dst = q ? B.d_src : B.d_img;
        uint8_t* h_dst = q ? B.h_src : B.h_ring;
        for (int i = 0; i < m; i++) {
            IngestItem& it = B.h_items[i];
            memset(&it, 0, sizeof(it));
            it.slot = slots[i0 + i];
            it.img = on_device ? images[i0 + i] : B.d_img + (size_t)i * c->P;
            it.src = q ? B.d_src + (size_t)i * bytes : nullptr;
            fill_meta(it.meta, K + 4 * (size_t)(i0 + i), Tcw + 12 * (size_t)(i0 + i));
            it.meta.uploaded = 1;
        }
        if (!on_device) {
            int n_pinned = 0;
            for (int i = 0; i < m; i++) n_pinned += pinned[i0 + i];
            if (n_pinned == m) {
                // images that lie back to back in the caller's pinned block (a frame queue) travel as one copy
                for (int i = 0; i < m;) {
                    int r = 1;
                    while (i + r < m && images[i0 + i + r] == images[i0 + i + r - 1] + bytes) r++;
                    if (hipMemcpyAsync(d_dst + (size_t)i * bytes, images[i0 + i], (size_t)r * bytes, hipMemcpyHostToDevice, cs) != hipSuccess)
                        return bail(fail(SDM_EHIP, "hipMemcpyAsync (pinned image) failed"));
                    direct = true;
                    i += r;
                }
                direct = true;
            } else {
                stager.stage(chunk, h_dst, images + i0, m, bytes);
                if (hipMemcpyAsync(d_dst, h_dst, (size_t)m * bytes, hipMemcpyHostToDevice, cs) != hipSuccess)
                    return bail(fail(SDM_EHIP, "hipMemcpyAsync (staged images) failed"));
            }
        }
        if (dbg) tdbg[2] = now();
        if ((rc = ingest_publish(c, b, m, cs, !merge))) return bail(rc);
        if (dbg) tdbg[3] = now();
        for (int i = 0; i < m; i++) {
            const int slot = slots[i0 + i];
            reset_slot_state(c, slot);
            c->h_meta[slot] = B.h_items[i].meta;  // (I_stddev lives on the device only)
            c->act_lambdaG[slot] = c->dprm.lambdaG;
            c->recon_lambdaG[slot] = c->dprm.lambdaG;  // k_prepass_batch<ZERO> leaves an all-zero map: K1 need not clear it again
        }
        if (merge)
            merged_bufs[group_chunks] = b;
        else if ((rc = ingest_launch_chunk(c, b, m, group_kfs, true, q)))
            return bail(rc);
        group_kfs += m;
        group_chunks++;
        failed_from = i0 + m;
        if (group_chunks == sdm_ctx::ING_GROUP || i0 + m >= n) {
            if ((rc = finish_group())) {
                failed_from = group_first;  // no lists for any keyframe of this group
                return bail(rc);
            }
            group_first = i0 + m;
        }
        if (dbg) {
            tdbg[4] = now();
            if (tdbg[4] - tdbg[0] > dbg_ms)
                fprintf(stderr, "[sdm ingest] slow chunk: acquire %.3f  images %.3f  publish %.3f  launch %.3f ms\n",
                        tdbg[1] - tdbg[0], tdbg[2] - tdbg[1], tdbg[3] - tdbg[2], tdbg[4] - tdbg[3]);
        }
        last = b;
    }
    // the caller's buffers are free on return: pageable images were copied into the ring; copies that read pinned images
    // in place are awaited here (the pre-pass kernels are not)
    if ((rc = counts_queued(c, n, slots))) return rc;  // one event behind the last group's kernels (they store the list lengths)
    const double t_tail = dbg ? now() : 0.0;
    if (direct && last >= 0) {
        HIP_TRY(hipEventSynchronize(c->ing[last].copied));
        c->ing[last].copied_pending = false;
    }
    if (dbg && now() - t_entry > dbg_ms)
        fprintf(stderr, "[sdm ingest] slow call (%d keyframes): set-up %.3f  chunks %.3f  final wait %.3f ms\n", n, t_loop - t_entry,
                t_tail - t_loop, now() - t_tail);
    return SDM_OK;
}

int ingest_images(sdm_ctx* c, int n, const int* slots, const uint8_t* const* images, bool on_device, const IngestParams* q,
                  const float* K, const float* Tcw)
{
    try {  // (std::vector / std::string allocations: nothing may escape a C-ABI entry point)
        return ingest_images_impl(c, n, slots, images, on_device, q, K, Tcw);
    } catch (const std::exception& e) {
        return fail(SDM_EHIP, std::string("ingest: ") + e.what());
    } catch (...) {
        return fail(SDM_EHIP, "ingest: unknown exception");
    }
}

int colour_order(int order, IngestParams& q)
{
    switch (order) {
        case SDM_ORDER_RGB: q.channels = 3; q.r_idx = 0; q.g_idx = 1; q.b_idx = 2; break;
        case SDM_ORDER_BGR: q.channels = 3; q.r_idx = 2; q.g_idx = 1; q.b_idx = 0; break;
        case SDM_ORDER_RGBA: q.channels = 4; q.r_idx = 0; q.g_idx = 1; q.b_idx = 2; break;
        case SDM_ORDER_BGRA: q.channels = 4; q.r_idx = 2; q.g_idx = 1; q.b_idx = 0; break;
        case SDM_ORDER_GRAY: q.channels = 1; break;
        default: return fail(SDM_EINVAL, "unknown colour order");
    }
    return SDM_OK;
}

}  // namespace

int sdm_upload_images_batch(sdm_ctx* c, int n, const int* slots, const uint8_t* const* images, const float* K, const float* Tcw)
{
    return ingest_images(c, n, slots, images, false, nullptr, K, Tcw);
}

int sdm_upload_image(sdm_ctx* c, int slot, const uint8_t* im, const float K[4], const float Tcw[12])
{
    return ingest_images(c, 1, &slot, &im, false, nullptr, K, Tcw);
}

// Tracking.cc:244-257 + 266-271 and Modeler.cc:154-155 on the device: colour order, lens undistortion, gray conversion,
// then the same pre-pass as sdm_upload_image.
int sdm_upload_images_rgb_batch(sdm_ctx* c, int n, const int* slots, const uint8_t* const* pixels, int order, const float* K,
                                const float dist[5], const float* Tcw)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (n > 0 && !K) return fail(SDM_EINVAL, "null input");
    IngestParams q{};
    int rc = colour_order(order, q);
    if (rc) return rc;
    if (n > 0) {  // one camera per batch: the first keyframe's intrinsics undistort every frame
        q.fx = K[0]; q.fy = K[1]; q.cx = K[2]; q.cy = K[3];
        for (int i = 1; i < n && dist; i++)
            if (memcmp(K, K + 4 * (size_t)i, sizeof(float) * 4) != 0)
                return fail(SDM_EINVAL, "one batch undistorts with one camera: the K of its frames differ");
    }
    q.undistort = dist != nullptr;
    if (dist) { q.k1 = dist[0]; q.k2 = dist[1]; q.p1 = dist[2]; q.p2 = dist[3]; q.k3 = dist[4]; }
    if (dist && n > 0 && !(K[0] != 0.0f && K[1] != 0.0f)) return fail(SDM_EINVAL, "zero focal length");
    return ingest_images(c, n, slots, pixels, false, &q, K, Tcw);
}

int sdm_upload_image_rgb(sdm_ctx* c, int slot, const uint8_t* pixels, int order, const float K[4], const float dist[5],
                         const float Tcw[12])
{
    return sdm_upload_images_rgb_batch(c, 1, &slot, &pixels, order, K, dist, Tcw);
}

int sdm_upload_image_device(sdm_ctx* c, int slot, const void* d_im, const float K[4], const float Tcw[12])
{
    const uint8_t* im = (const uint8_t*)d_im;
    int rc = ingest_images(c, 1, &slot, &im, true, nullptr, K, Tcw);
    if (rc) return rc;
    return sync_counts(c);  // the caller's device image has been consumed on return
}

void* sdm_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    g_pinned_blocks[(const uint8_t*)p] = std::max<size_t>(bytes, 1);
    return p;
}

void sdm_host_free(void* p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        g_pinned_blocks.erase((const uint8_t*)p);
    }
    (void)hipHostFree(p);
}

int sdm_set_pose(sdm_ctx* c, int slot, const float Tcw[12])
{
    int rc = check_slot(c, slot, true);
    if (rc) return rc;
    if (!Tcw) return fail(SDM_EINVAL, "null pose");
    if (memcmp(c->h_meta[slot].Tcw, Tcw, sizeof(float) * 12) == 0) return SDM_OK;  // unchanged: keep cached tables
    memcpy(c->h_meta[slot].Tcw, Tcw, sizeof(float) * 12);
    return push_meta(c, slot, true);
}

int sdm_download_inputs(sdm_ctx* c, int slot, uint8_t* im, float* grad, float* theta, float* I_stddev)
{
    int rc = check_slot(c, slot, true);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->cfg.device));
    hipLaunchKernelGGL(k_unpack, dim3(blocks_for(c->P)), dim3(BLOCK), 0, c->stream, c->rec + (long long)slot * c->P,
                       (int)c->P, c->d_im, c->d_grad, c->d_theta);
    HIP_TRY(hipGetLastError());
    if (im) HIP_TRY(hipMemcpyAsync(im, c->d_im, (size_t)c->P, hipMemcpyDeviceToHost, c->stream));
    if (grad) HIP_TRY(hipMemcpyAsync(grad, c->d_grad, sizeof(float) * c->P, hipMemcpyDeviceToHost, c->stream));
    if (theta) HIP_TRY(hipMemcpyAsync(theta, c->d_theta, sizeof(float) * c->P, hipMemcpyDeviceToHost, c->stream));
    if (I_stddev)
        HIP_TRY(hipMemcpyAsync(I_stddev, &c->d_meta[slot].I_stddev, sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SDM_OK;
}

// ---- K1..K3 ----------------------------------------------------------------------------------------------------
static bool all_pipeline_maps(sdm_ctx* c, int n_ref, const int* ref_slots);

static int launch_search_fuse(sdm_ctx* c, int n_ref, int n, const int* ref_slots)
{
    // K1 writes every listed pixel; the rest of the map must be zero (a fresh depth_map_).  It already is
    // when the slot's current map came out of SemiDenseRecon under the same lambdaG.
    if (!all_pipeline_maps(c, n_ref, ref_slots)) {
        for_ref_slices(n_ref, blocks_for(c->P), BLOCK, [&](int first, int count) {
            hipLaunchKernelGGL(k_zero_maps, dim3(blocks_for(c->P * count)), dim3(BLOCK), 0, c->stream, c->pool, c->P,
                               c->d_ref_slots + first, count);
        });
        HIP_TRY(hipGetLastError());
    }
    StageTimer tm(c, SDM_STAGE_SEARCH_FUSE);  // brackets one k_search_fuse launch and its k_fuse_open
    int max_chunks = 0;
    for (int r = 0; r < n_ref; r++) max_chunks = std::max(max_chunks, (c->h_act_count[ref_slots[r]] + K1_PX - 1) / K1_PX);
    if (max_chunks == 0) return SDM_OK;  // no pixel passes the gradient gate: the maps stay zero
    const size_t lds = k1_lds_bytes(n);
    const int blocks_per_ref = 8 * ((max_chunks + 7) / 8);
    OpenList ol;
    ol.count = c->d_open_ctr + 4 * (c->open_launch & 1u);
    ol.next = c->d_open_ctr + 4 * ((c->open_launch + 1u) & 1u);
    ol.capacity = c->open_capacity;
    ol.quota = c->open_quota;
    ol.inplace_min = c->open_inplace_min;
    ol.pix = c->d_open_pix;
    ol.vm = c->d_open_vm;
    ol.hyp = c->d_open_hyp;
    c->open_launch++;
    // the pixels the fusion bounds left open, 64 per workgroup; the list length stays on the device (a fixed grid walks
    // whatever is there, nothing when the list is empty)
    const size_t lds_open = (sizeof(float2) + sizeof(float)) * (size_t)K1_PX * n + sizeof(unsigned) * (size_t)K1_PX * ((n + 3) / 4);
    const int grid_open = (int)std::min<unsigned>(c->open_capacity / K1_PX, c->open_grid);
    // (the slices of one call append to the same open list; k_fuse_open runs once behind the last one)
    // the mask-scan instantiation only where a pair of the call can use it (or the diagnostic mode asks for it)
    const bool mask = c->scan_mode == 2 || (c->scan_mode == 0 && c->sets[c->cur_set].key.long_ranges);
    auto k1 = c->stats_on ? (mask ? k_search_fuse<true, true> : k_search_fuse<true, false>)
                          : (mask ? k_search_fuse<false, true> : k_search_fuse<false, false>);
    for_ref_slices(n_ref, blocks_per_ref, K1_BLOCK, [&](int first, int count) {
        hipLaunchKernelGGL(k1, dim3(blocks_per_ref * count), dim3(K1_BLOCK), lds, c->stream, c->rec, c->P, c->d_refs + first,
                           c->d_pairs + (size_t)first * n, count, n, c->W, c->H, max_chunks, c->dprm, c->d_act, c->pool, c->d_stats,
                           ol, c->d_gmask, c->mrow);
    });
    if (c->stats_on)
        hipLaunchKernelGGL(k_fuse_open<true>, dim3(grid_open), dim3(K1_BLOCK), lds_open, c->stream, ol, n, c->dprm, c->pool,
                           c->P * c->cfg.max_keyframes, c->d_stats);
    else
        hipLaunchKernelGGL(k_fuse_open<false>, dim3(grid_open), dim3(K1_BLOCK), lds_open, c->stream, ol, n, c->dprm, c->pool,
                           c->P * c->cfg.max_keyframes, c->d_stats);
    HIP_TRY(hipGetLastError());
    return SDM_OK;
}

int sdm_active_count(sdm_ctx* c, int slot, int* count)
{
    int rc = check_slot(c, slot, true);
    if (rc) return rc;
    if (!count) return fail(SDM_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (!(c->act_lambdaG[slot] == c->dprm.lambdaG))
        if ((rc = build_active(c, slot))) return rc;
    if ((rc = sync_counts(c))) return rc;
    *count = c->h_act_count[slot];
    return SDM_OK;
}

int sdm_download_active_list(sdm_ctx* c, int slot, unsigned* list, int capacity, int* count, unsigned long long* hash)
{
    int n = 0;
    int rc = sdm_active_count(c, slot, &n);
    if (rc) return rc;
    if (count) *count = n;
    if (list && capacity < n) return fail(SDM_EINVAL, "list buffer too small");
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (list && n > 0)
        HIP_TRY(hipMemcpy(list, c->d_act + (long long)slot * c->P, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToHost));
    if (hash) HIP_TRY(hipMemcpy(hash, c->d_act_hash + slot, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return SDM_OK;
}

int sdm_search_fuse(sdm_ctx* c, int n_ref, const int* ref_slots, int n, const int* nbr_slots, const float* rot,
                    const float* mind, const float* maxd)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (n < 1) return fail(SDM_EINVAL, "need at least one neighbour");
    if (!mind || !maxd) return fail(SDM_EINVAL, "null depth bounds");
    int rc = stage_tables(c, n_ref, ref_slots, n, nbr_slots, rot, mind, maxd, true);
    if (rc) return rc;
    if ((rc = launch_search_fuse(c, n_ref, n, ref_slots))) return rc;
    for (int r = 0; r < n_ref; r++) {
        c->has_depth[ref_slots[r]] = 1;
        c->recon_lambdaG[ref_slots[r]] = c->dprm.lambdaG;
    }
    return tables_staged(c);
}

// stencil passes over [first, first+count) of the staged reference list.
// mode 0: check pool -> scratch; mode 1: grow scratch -> pool; mode 2: check pool->scratch then copy back
static int launch_intra(sdm_ctx* c, int n_ref, int first, int count, bool check, bool grow)
{
    // offsets: [0..K) pool offsets, [K..2K) scratch offsets, [2K..3K) record offsets (in floats)
    const int K = n_ref;  // offset tables are [3][n_ref]
    if (check) {
        for_ref_slices(count, grid_blocks(c->geom, 1), BLOCK, [&](int f, int cn) {
            hipLaunchKernelGGL(k_intra_check, dim3(grid_blocks(c->geom, cn)), dim3(BLOCK), 0, c->stream, c->pool, c->scratch,
                               c->d_off + first + f, c->d_off + K + first + f, cn, c->geom);
        });
        HIP_TRY(hipGetLastError());
    }
    if (grow) {
        for_ref_slices(count, grid_blocks(c->geom, 1), BLOCK, [&](int f, int cn) {
            hipLaunchKernelGGL(k_intra_grow, dim3(grid_blocks(c->geom, cn)), dim3(BLOCK), 0, c->stream, c->scratch, c->pool,
                               c->d_off + K + first + f, c->d_off + first + f, (const float*)c->rec,
                               c->d_off + 2 * K + first + f, 4, cn, c->geom, c->dprm.lambdaG);
        });
        HIP_TRY(hipGetLastError());
    }
    return SDM_OK;
}

// K2/K3 on "pipeline" maps (zero outside the keyframe's active list: written by K1, or declared so by
// sdm_assume_pipeline_maps).  RefConst (act_count, slot) of the staged batch is valid here.
static int run_intra_lists(sdm_ctx* c, int n_ref, const int* ref_slots, bool check, bool grow)
{
    int rc = SDM_OK;
    StageTimer tm(c, SDM_STAGE_INTRA);
    const int K = n_ref, cap = c->cfg.batch_capacity;  // offset tables are [3][n_ref], staged with the tables
    for (int first = 0; first < n_ref; first += cap) {
        const int count = std::min(cap, n_ref - first);
        int max_chunks = 0;
        for (int r = 0; r < count; r++)
            max_chunks = std::max(max_chunks, (c->h_act_count[ref_slots[first + r]] + BLOCK - 1) / BLOCK);
        if (max_chunks == 0) continue;
        const int per_ref = band_grid_per_ref(max_chunks, SDM_K23_GROUP, SDM_K23_BAND);
#if SDM_INTRA_COMPACT
        // K2 -> compact results (the scratch memory, indexed by list position) -> commit into the pool; K3 = the (normally
        // empty) candidate list K2 collected, grown by one small launch, reads before writes (sdm_kernels.h)
        GrowList gl;
        gl.count = c->d_grow_ctr + (c->grow_launch & 1u);
        gl.next = c->d_grow_ctr + ((c->grow_launch + 1u) & 1u);
        gl.capacity = c->grow_capacity;
        gl.pix = c->d_grow_pix;
        gl.val = c->d_grow_val;
        for_ref_slices(count, per_ref, BLOCK, [&](int f, int cn) {
            if (check && grow)
                hipLaunchKernelGGL((k_intra_compact<true, true>), dim3(per_ref * cn), dim3(BLOCK), 0, c->stream, c->pool,
                                   c->scratch, c->d_off, c->d_off + K, c->d_refs, first + f, cn, c->W, max_chunks, c->P,
                                   c->d_act, gl);
            else if (check)
                hipLaunchKernelGGL((k_intra_compact<true, false>), dim3(per_ref * cn), dim3(BLOCK), 0, c->stream, c->pool,
                                   c->scratch, c->d_off, c->d_off + K, c->d_refs, first + f, cn, c->W, max_chunks, c->P,
                                   c->d_act, gl);
            else
                hipLaunchKernelGGL((k_intra_compact<false, true>), dim3(per_ref * cn), dim3(BLOCK), 0, c->stream, c->pool,
                                   c->scratch, c->d_off, c->d_off + K, c->d_refs, first + f, cn, c->W, max_chunks, c->P,
                                   c->d_act, gl);
        });
        HIP_TRY(hipGetLastError());
        if (check) {
            for_ref_slices(count, per_ref, BLOCK, [&](int f, int cn) {
                hipLaunchKernelGGL(k_intra_commit, dim3(per_ref * cn), dim3(BLOCK), 0, c->stream, c->pool, c->scratch, c->d_off,
                                   c->d_off + K, c->d_refs, first + f, cn, c->W, max_chunks, c->P, c->d_act);
            });
            HIP_TRY(hipGetLastError());
        }
        if (grow) {
            hipLaunchKernelGGL(k_grow, dim3(1), dim3(GROW_BLOCK), 0, c->stream, c->pool, gl, c->W, c->P, c->scratch, c->d_off,
                               c->d_off + K, c->d_refs, first, count, c->d_act);
            HIP_TRY(hipGetLastError());
            c->grow_launch++;
        }
#else
        if (check) {
            // K2 writes every listed pixel of the scratch planes, and K3's list kernel substitutes zeros for
            // neighbours outside the list instead of reading them, so the planes need no clearing -- unless the
            // pass stands alone and the whole plane is copied back below
            if (!grow) HIP_TRY(hipMemsetAsync(c->scratch, 0, sizeof(float2) * c->P * count, c->stream));
            for_ref_slices(count, per_ref, BLOCK, [&](int f, int cn) {
                hipLaunchKernelGGL(k_intra_list<false>, dim3(per_ref * cn), dim3(BLOCK), 0, c->stream, c->pool, c->scratch,
                                   c->d_off, c->d_off + K, c->d_refs, first + f, cn, c->W, max_chunks, c->P, c->d_act, c->rec,
                                   c->H, c->dprm.lambdaG);
            });
            HIP_TRY(hipGetLastError());
        } else {
            for (int r = 0; r < count; r++)
                HIP_TRY(hipMemcpyAsync(c->scratch + (long long)r * c->P, c->pool + c->h_off[first + r],
                                       sizeof(float2) * c->P, hipMemcpyDeviceToDevice, c->stream));
        }
        if (grow) {
            for_ref_slices(count, per_ref, BLOCK, [&](int f, int cn) {
                hipLaunchKernelGGL(k_intra_list<true>, dim3(per_ref * cn), dim3(BLOCK), 0, c->stream, c->scratch, c->pool,
                                   c->d_off + K, c->d_off, c->d_refs, first + f, cn, c->W, max_chunks, c->P, c->d_act, c->rec,
                                   c->H, c->dprm.lambdaG);
            });
            HIP_TRY(hipGetLastError());
        } else {
            for (int r = 0; r < count; r++)
                HIP_TRY(hipMemcpyAsync(c->pool + c->h_off[first + r], c->scratch + (long long)r * c->P,
                                       sizeof(float2) * c->P, hipMemcpyDeviceToDevice, c->stream));
        }
#endif
    }
    return SDM_OK;
}

static bool all_pipeline_maps(sdm_ctx* c, int n_ref, const int* ref_slots)
{
    for (int r = 0; r < n_ref; r++)
        if (!(c->recon_lambdaG[ref_slots[r]] == c->dprm.lambdaG)) return false;
    return true;
}

static int run_intra(sdm_ctx* c, int n_ref, const int* ref_slots, bool check, bool grow)
{
    // K2 writes scratch, K3 writes back to the pool.  A lone pass is completed by a device copy.
    int rc = SDM_OK;
    StageTimer tm(c, SDM_STAGE_INTRA);
    const int cap = c->cfg.batch_capacity;
    for (int first = 0; first < n_ref; first += cap) {
        const int count = std::min(cap, n_ref - first);
        if (check && grow) {
            if ((rc = launch_intra(c, n_ref, first, count, true, true))) return rc;
        } else if (check) {
            if ((rc = launch_intra(c, n_ref, first, count, true, false))) return rc;
            for (int r = 0; r < count; r++)
                HIP_TRY(hipMemcpyAsync(c->pool + c->h_off[first + r], c->scratch + (long long)r * c->P,
                                       sizeof(float2) * c->P, hipMemcpyDeviceToDevice, c->stream));
        } else {
            for (int r = 0; r < count; r++)
                HIP_TRY(hipMemcpyAsync(c->scratch + (long long)r * c->P, c->pool + c->h_off[first + r],
                                       sizeof(float2) * c->P, hipMemcpyDeviceToDevice, c->stream));
            if ((rc = launch_intra(c, n_ref, first, count, false, true))) return rc;
        }
    }
    return SDM_OK;
}

int sdm_intra_check(sdm_ctx* c, int n_ref, const int* ref_slots)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    int rc = stage_tables(c, n_ref, ref_slots, 0, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (all_pipeline_maps(c, n_ref, ref_slots))
        rc = run_intra_lists(c, n_ref, ref_slots, true, false);
    else
        rc = run_intra(c, n_ref, ref_slots, true, false);
    if (rc) return rc;
    return tables_staged(c);
}

int sdm_intra_grow(sdm_ctx* c, int n_ref, const int* ref_slots)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    int rc = stage_tables(c, n_ref, ref_slots, 0, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (all_pipeline_maps(c, n_ref, ref_slots))
        rc = run_intra_lists(c, n_ref, ref_slots, false, true);
    else
        rc = run_intra(c, n_ref, ref_slots, false, true);
    if (rc) return rc;
    return tables_staged(c);
}

int sdm_recon(sdm_ctx* c, int n_ref, const int* ref_slots, int n, const int* nbr_slots, const float* rot,
              const float* mind, const float* maxd)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (n < 1) return fail(SDM_EINVAL, "need at least one neighbour");
    if (!mind || !maxd) return fail(SDM_EINVAL, "null depth bounds");
    int rc = stage_tables(c, n_ref, ref_slots, n, nbr_slots, rot, mind, maxd, true);
    if (rc) return rc;
    if ((rc = launch_search_fuse(c, n_ref, n, ref_slots))) return rc;  // PM.cc:197-231
    static const bool sync_k1 = [] {  // debugging knob (tools/debug/unsliced_vs_sliced.py), read once per process
        const char* e = getenv("SDM_DEBUG_SYNC_K1");
        return e && atoi(e) == 1;
    }();
    if (sync_k1) HIP_TRY(hipStreamSynchronize(c->stream));
    if ((rc = run_intra_lists(c, n_ref, ref_slots, true, true))) return rc;  // PM.cc:237-238
    for (int r = 0; r < n_ref; r++) {
        c->has_depth[ref_slots[r]] = 1;  // kf->semidense_flag_, PM.cc:244
        c->recon_lambdaG[ref_slots[r]] = c->dprm.lambdaG;
    }
    return tables_staged(c);
}

// ---- K4 / K5 ------------------------------------------------------------------------------------------------------
static int inter_check_core(sdm_ctx* c, int n_ref, const int* ref_slots, int n, const int* nbr_slots, int commit,
                            bool want_xyz, bool* xyz_done)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (n < 1) return fail(SDM_EINVAL, "need at least one neighbour");
    if (!ref_slots || !nbr_slots || n_ref < 1) return fail(SDM_EINVAL, "null or empty slot list");
    // the reference runs the check only when the keyframe and all its neighbours have been reconstructed
    // (semidense_flag_, PM.cc:292-298): a slot whose map is still the zero map of a fresh upload is a caller error
    for (int r = 0; r < n_ref; r++) {
        if (ref_slots[r] < 0 || ref_slots[r] >= c->cfg.max_keyframes) return fail(SDM_EINVAL, "slot out of range");
        if (!c->has_depth[ref_slots[r]]) return fail(SDM_ESTATE, "reference slot has no depth map (run sdm_recon first)");
        for (int j = 0; j < n; j++) {
            const int s = nbr_slots[r * n + j];
            if (s < 0 || s >= c->cfg.max_keyframes) return fail(SDM_EINVAL, "slot out of range");
            if (!c->has_depth[s])
                return fail(SDM_ESTATE, "neighbour slot has no depth map (sdm_recon, sdm_upload_depth, an sdm_exchange_* "
                                        "call or sdm_mark_depth_present must come first)");
        }
    }
    int rc = stage_tables(c, n_ref, ref_slots, n, nbr_slots, nullptr, nullptr, nullptr);
    if (rc) return rc;
    *xyz_done = false;
    {
        StageTimer tm(c, SDM_STAGE_INTER);
        bool from_recon = true;  // every reference map produced by SemiDenseRecon under the current lambdaG?
        int max_chunks = 0;
        for (int r = 0; r < n_ref; r++) {
            from_recon = from_recon && (c->recon_lambdaG[ref_slots[r]] == c->dprm.lambdaG);
            max_chunks = std::max(max_chunks, (c->h_act_count[ref_slots[r]] + BLOCK - 1) / BLOCK);
        }
        if (from_recon) {
            bool chk_ok = true;  // checked planes already zero outside the lists?
            for (int r = 0; r < n_ref; r++) chk_ok = chk_ok && c->chk_sparse[ref_slots[r]];
            if (!chk_ok) {
                for_ref_slices(n_ref, blocks_for(c->P), BLOCK, [&](int first, int count) {
                    hipLaunchKernelGGL(k_rho_copy, dim3(blocks_for(c->P * count)), dim3(BLOCK), 0, c->stream, c->pool, c->chk,
                                       c->P, c->d_ref_slots + first, count);
                });
                HIP_TRY(hipGetLastError());
            }
            for (int r = 0; r < n_ref; r++) c->chk_sparse[ref_slots[r]] = 1;
            // the point set can ride along when its plane is zero outside the lists as well (K5's list form)
            bool fuse = want_xyz && c->xyz != nullptr;
            for (int r = 0; r < n_ref && fuse; r++) fuse = c->xyz_sparse[ref_slots[r]] != 0;
            if (max_chunks > 0) {
                max_chunks = (max_chunks * BLOCK + K4_BLOCK - 1) / K4_BLOCK;  // in units of the list kernel's workgroup
                const int per_ref = band_grid_per_ref(max_chunks, SDM_K4_GROUP, SDM_K4_BAND);
                for_ref_slices(n_ref, per_ref, K4_BLOCK, [&](int first, int count) {
                    const dim3 grid(per_ref * count);
                    if (fuse)
                        hipLaunchKernelGGL(k_inter_check_list<true>, grid, dim3(K4_BLOCK), c->k4_lds_pad, c->stream, c->pool, c->P,
                                           c->d_refs + first, c->d_pairs + (size_t)first * n, count, n, c->W, c->H, max_chunks,
                                           c->dprm.lambdaN, c->d_act, c->chk, c->d_meta, c->xyz);
                    else
                        hipLaunchKernelGGL(k_inter_check_list<false>, grid, dim3(K4_BLOCK), c->k4_lds_pad, c->stream, c->pool, c->P,
                                           c->d_refs + first, c->d_pairs + (size_t)first * n, count, n, c->W, c->H, max_chunks,
                                           c->dprm.lambdaN, c->d_act, c->chk, c->d_meta, c->xyz);
                });
                HIP_TRY(hipGetLastError());
            }
            *xyz_done = fuse;
        } else {
            for (int r = 0; r < n_ref; r++) c->chk_sparse[ref_slots[r]] = 0;  // the generic kernel copies arbitrary maps
            for_ref_slices(n_ref, grid_blocks(c->geom, 1), BLOCK, [&](int first, int count) {
                hipLaunchKernelGGL(k_inter_check, dim3(grid_blocks(c->geom, count)), dim3(BLOCK), 0, c->stream, c->pool, c->P,
                                   c->d_refs + first, c->d_pairs + (size_t)first * n, count, n, c->geom, c->dprm.lambdaN,
                                   c->chk);
            });
            HIP_TRY(hipGetLastError());
        }
        if (commit) {
            for_ref_slices(n_ref, blocks_for(c->P), BLOCK, [&](int first, int count) {
                hipLaunchKernelGGL(k_commit, dim3(blocks_for(c->P * count)), dim3(BLOCK), 0, c->stream, c->chk, c->pool, c->P,
                                   c->d_ref_slots + first, count);
            });
            HIP_TRY(hipGetLastError());
        }
    }
    for (int r = 0; r < n_ref; r++) c->has_chk[ref_slots[r]] = 1;  // kf->interKF_depth_flag_, PM.cc:306
    return tables_staged(c);
}

int sdm_inter_check(sdm_ctx* c, int n_ref, const int* ref_slots, int n, const int* nbr_slots, int commit)
{
    bool xyz_done;
    return inter_check_core(c, n_ref, ref_slots, n, nbr_slots, commit, false, &xyz_done);
}

int sdm_inter_check_pointset(sdm_ctx* c, int n_ref, const int* ref_slots, int n, const int* nbr_slots, int commit)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (!c->xyz) return fail(SDM_ESTATE, "context created without with_pointset");
    bool xyz_done = false;
    int rc = inter_check_core(c, n_ref, ref_slots, n, nbr_slots, commit, true, &xyz_done);
    if (rc || xyz_done) return rc;
    return sdm_pointset(c, n_ref, ref_slots, 1);  // maps that are not pipeline maps: the two passes
}

int sdm_pointset(sdm_ctx* c, int n_ref, const int* ref_slots, int source)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (!c->xyz) return fail(SDM_ESTATE, "context created without with_pointset");
    int rc = stage_tables(c, n_ref, ref_slots, 0, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    const float* src = source ? c->chk : (const float*)c->pool;
    const int sstride = source ? 1 : 2;
    StageTimer tm(c, SDM_STAGE_POINTSET);
    // list form when source map and xyz plane are both zero outside the active lists
    bool sparse = all_pipeline_maps(c, n_ref, ref_slots);
    int max_chunks = 0;
    for (int r = 0; r < n_ref; r++) {
        sparse = sparse && c->xyz_sparse[ref_slots[r]] && (!source || c->chk_sparse[ref_slots[r]]);
        max_chunks = std::max(max_chunks, (c->h_act_count[ref_slots[r]] + BLOCK - 1) / BLOCK);
    }
    if (sparse) {
        if (max_chunks > 0) {
            const int per_ref = band_grid_per_ref(max_chunks, SDM_K23_GROUP, SDM_K23_BAND);
            for_ref_slices(n_ref, per_ref, BLOCK, [&](int first, int count) {
                hipLaunchKernelGGL(k_pointset_list, dim3(per_ref * count), dim3(BLOCK), 0, c->stream, src, sstride, c->P,
                                   c->d_meta, c->d_refs + first, count, c->W, max_chunks, c->d_act, c->xyz);
            });
        }
    } else {
        for_ref_slices(n_ref, blocks_for(c->P), BLOCK, [&](int first, int count) {
            hipLaunchKernelGGL(k_pointset, dim3(blocks_for(c->P), count), dim3(BLOCK), 0, c->stream, src, sstride, c->P,
                               c->d_meta, c->d_ref_slots + first, count, c->W, c->H, c->xyz);
        });
        // a full rewrite leaves zeros wherever the source is zero: sparse again iff the source was
        for (int r = 0; r < n_ref; r++)
            c->xyz_sparse[ref_slots[r]] = (c->recon_lambdaG[ref_slots[r]] == c->dprm.lambdaG) &&
                                          (!source || c->chk_sparse[ref_slots[r]]);
    }
    HIP_TRY(hipGetLastError());
    return tables_staged(c);
}

// ---- map transfer ----------------------------------------------------------------------------------------------------
static int upload_f2(sdm_ctx* c, float2* dst, const float* rho, const float* sigma)
{
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (long long i = 0; i < c->P; i++) c->h_f2[i] = make_float2(rho[i], sigma[i]);
    HIP_TRY(hipMemcpyAsync(dst, c->h_f2, sizeof(float2) * c->P, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SDM_OK;
}
static int download_f2(sdm_ctx* c, const float2* src, float* rho, float* sigma)
{
    HIP_TRY(hipMemcpyAsync(c->h_f2, src, sizeof(float2) * c->P, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (long long i = 0; i < c->P; i++) {
        if (rho) rho[i] = c->h_f2[i].x;
        if (sigma) sigma[i] = c->h_f2[i].y;
    }
    return SDM_OK;
}

int sdm_upload_depth(sdm_ctx* c, int slot, const float* rho, const float* sigma)
{
    int rc = check_slot(c, slot, false);
    if (rc) return rc;
    if (!rho || !sigma) return fail(SDM_EINVAL, "null map");
    HIP_TRY(hipSetDevice(c->cfg.device));
    c->has_depth[slot] = 1;
    c->recon_lambdaG[slot] = std::nanf("");  // arbitrary map: support is no longer tied to the active list
    return upload_f2(c, c->pool + (long long)slot * c->P, rho, sigma);
}

int sdm_download_depth(sdm_ctx* c, int slot, float* rho, float* sigma)
{
    int rc = check_slot(c, slot, false);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->cfg.device));
    return download_f2(c, c->pool + (long long)slot * c->P, rho, sigma);
}

int sdm_download_checked(sdm_ctx* c, int slot, float* rho)
{
    int rc = check_slot(c, slot, false);
    if (rc) return rc;
    if (!rho) return fail(SDM_EINVAL, "null map");
    if (!c->has_chk[slot]) return fail(SDM_ESTATE, "slot has not been inter-keyframe checked");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipMemcpyAsync(rho, c->chk + (long long)slot * c->P, sizeof(float) * c->P, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SDM_OK;
}

int sdm_download_pointset(sdm_ctx* c, int slot, float* xyz)
{
    int rc = check_slot(c, slot, false);
    if (rc) return rc;
    if (!xyz) return fail(SDM_EINVAL, "null map");
    if (!c->xyz) return fail(SDM_ESTATE, "context created without with_pointset");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipMemcpyAsync(xyz, c->xyz + (long long)slot * c->P * 3, sizeof(float) * 3 * c->P, hipMemcpyDeviceToHost,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SDM_OK;
}

void* sdm_depth_pool_ptr(sdm_ctx* c) { return c ? (void*)c->pool : nullptr; }

int sdm_assume_pipeline_maps(sdm_ctx* c, int n, const int* slots)
{
    if (!c || !slots) return fail(SDM_EINVAL, "null argument");
    for (int i = 0; i < n; i++) {
        int rc = check_slot(c, slots[i], true);
        if (rc) return rc;
        if (!(c->act_lambdaG[slots[i]] == c->dprm.lambdaG))
            if ((rc = build_active(c, slots[i]))) return rc;
        c->recon_lambdaG[slots[i]] = c->dprm.lambdaG;
        c->has_depth[slots[i]] = 1;
    }
    return SDM_OK;
}

// ---- stand-alone map operations (PM.h:85-86 signatures): scratch slot 0 in, slot 1 out -------------------------------------
static int intra_maps(sdm_ctx* c, float* rho, float* sigma, const float* grad, bool grow)
{
    if (!c || !rho || !sigma) return fail(SDM_EINVAL, "null argument");
    if (grow && !grad) return fail(SDM_EINVAL, "IntraKeyFrameDepthGrowing needs gradimg");
    HIP_TRY(hipSetDevice(c->cfg.device));
    int rc = wait_tables(c);
    if (rc) return rc;
    if ((rc = upload_f2(c, c->scratch, rho, sigma))) return rc;
    // private use of the staging block: [in, out, grad] offsets of one map; the cached tables are gone
    c->sets[c->cur_set].key.valid = false;
    c->h_off = reinterpret_cast<long long*>(c->h_tab);
    c->d_off = reinterpret_cast<long long*>(c->d_tab);
    const int K = 1;
    c->h_off[0] = 0;
    c->h_off[K] = c->P;
    c->h_off[2 * K] = 0;
    HIP_TRY(hipMemcpyAsync(c->d_off, c->h_off, sizeof(long long) * 3, hipMemcpyHostToDevice, c->stream));
    const int grid = grid_blocks(c->geom, 1);
    if (!grow) {
        hipLaunchKernelGGL(k_intra_check, dim3(grid), dim3(BLOCK), 0, c->stream, c->scratch, c->scratch, c->d_off,
                           c->d_off + K, 1, c->geom);
    } else {
        HIP_TRY(hipMemcpyAsync(c->d_grad, grad, sizeof(float) * c->P, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_intra_grow, dim3(grid), dim3(BLOCK), 0, c->stream, c->scratch, c->scratch, c->d_off,
                           c->d_off + K, (const float*)c->d_grad, c->d_off + 2 * K, 1, 1, c->geom, c->dprm.lambdaG);
    }
    HIP_TRY(hipGetLastError());
    return download_f2(c, c->scratch + c->P, rho, sigma);
}

int sdm_intra_check_maps(sdm_ctx* c, float* rho, float* sigma, const float* grad)
{
    (void)grad;  // the reference's IntraKeyFrameDepthChecking never reads gradimg (PM.cc:486-547)
    return intra_maps(c, rho, sigma, grad, false);
}
int sdm_intra_grow_maps(sdm_ctx* c, float* rho, float* sigma, const float* grad)
{
    return intra_maps(c, rho, sigma, grad, true);
}

// ---- per-pixel entry points ----------------------------------------------------------------------------------------------
static int read_small(sdm_ctx* c, float* out, int n)
{
    HIP_TRY(hipMemcpyAsync(out, c->d_small, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SDM_OK;
}

int sdm_epipolar_search(sdm_ctx* c, int ref_slot, int nbr_slot, int x, int y, float mind, float maxd, float rot,
                        float out[5])
{
    if (!c || !out) return fail(SDM_EINVAL, "null argument");
    if (x < 0 || x >= c->W || y < 0 || y >= c->H) return fail(SDM_EINVAL, "pixel out of range");
    int rc = stage_tables(c, 1, &ref_slot, 1, &nbr_slot, &rot, &mind, &maxd, true);
    if (rc) return rc;
    hipLaunchKernelGGL(k_epipolar_search_px, dim3(1), dim3(1), 0, c->stream, c->rec, c->P, c->d_refs, c->d_pairs, c->W,
                       c->H, x, y, c->dprm, c->d_small, c->d_gmask, c->mrow);
    HIP_TRY(hipGetLastError());
    if ((rc = tables_staged(c))) return rc;
    return read_small(c, out, 5);
}

int sdm_search_range(sdm_ctx* c, int ref_slot, int nbr_slot, int x, int y, float mind, float maxd, float* umin,
                     float* umax)
{
    if (!c || !umin || !umax) return fail(SDM_EINVAL, "null argument");
    int rc = stage_tables(c, 1, &ref_slot, 1, &nbr_slot, nullptr, &mind, &maxd, true);
    if (rc) return rc;
    hipLaunchKernelGGL(k_search_range_px, dim3(1), dim3(1), 0, c->stream, c->d_refs, c->d_pairs, c->W, x, y, c->d_small);
    HIP_TRY(hipGetLastError());
    if ((rc = tables_staged(c))) return rc;
    float o[2];
    if ((rc = read_small(c, o, 2))) return rc;
    *umin = o[0];
    *umax = o[1];
    return SDM_OK;
}

int sdm_fuse(sdm_ctx* c, const float* rho, const float* sigma, int n, float out[3])
{
    if (!c || !out || (n > 0 && (!rho || !sigma))) return fail(SDM_EINVAL, "null argument");
    if (n < 0 || n > SDM_MAX_NEIGHBOURS) return fail(SDM_EINVAL, "n out of range");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) c->h_f2[i] = make_float2(rho[i], sigma[i]);
    HIP_TRY(hipMemcpyAsync(c->scratch, c->h_f2, sizeof(float2) * std::max(n, 1), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_fuse_px, dim3(1), dim3(1), 0, c->stream, c->scratch, n, c->dprm.lambdaN, c->d_small);
    HIP_TRY(hipGetLastError());
    return read_small(c, out, 3);
}

int sdm_pair_geometry(sdm_ctx* c, int ref_slot, int nbr_slot, float F12[9], float R21[9], float t21[3])
{
    if (!c) return fail(SDM_EINVAL, "null context");
    int rc = stage_tables(c, 1, &ref_slot, 1, &nbr_slot, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if ((rc = tables_staged(c))) return rc;
    PairConst pc;
    HIP_TRY(hipMemcpyAsync(&pc, c->d_pairs, sizeof(PairConst), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (F12) memcpy(F12, pc.F, sizeof(float) * 9);
    if (R21) {
        memcpy(R21, pc.Rx, sizeof(float) * 3);
        memcpy(R21 + 3, pc.Ry, sizeof(float) * 3);
        memcpy(R21 + 6, pc.Rz, sizeof(float) * 3);
    }
    if (t21) {
        t21[0] = pc.tx;
        t21[1] = pc.ty;
        t21[2] = pc.tz;
    }
    return SDM_OK;
}

// ---- host helpers of the class surface --------------------------------------------------------------------------------------
int sdm_stereo_search_constraints(const float* d, int n, float* min_depth, float* max_depth)
{
    if (!d || n <= 0 || !min_depth || !max_depth) return fail(SDM_EINVAL, "bad argument");
    // PM.cc:373: std::accumulate(..., 0.0) accumulates in double
    double acc = 0.0;
    for (int i = 0; i < n; i++) acc = acc + (double)d[i];
    float sum = (float)acc;
    float mean = sum / (float)n;
    double acc2 = 0.0;  // PM.cc:378: std::inner_product(..., 0.0) over float differences
    for (int i = 0; i < n; i++) {
        float diff = d[i] - mean;
        float pr = diff * diff;
        acc2 = acc2 + (double)pr;
    }
    float variance = (float)(acc2 / (double)n);
    float stdev = std::sqrt(variance);
    *max_depth = 1.0f / (mean + 2.0f * stdev);  // PM.cc:381
    *min_depth = 1.0f / (mean - 2.0f * stdev);  // PM.cc:382
    return SDM_OK;
}

float sdm_median_rot_in_plane(const int* mp1, const float* angle1, int n1, const int* mp2, const float* angle2, int n2)
{
    std::vector<float> rot;  // PM.cc:467-484
    for (int i = 0; i < n1; i++) {
        if (mp1[i] < 0) continue;
        for (int j = 0; j < n2; j++) {
            if (mp2[j] != mp1[i]) continue;
            if (angle1[i] < 0 || angle2[j] < 0) continue;
            rot.push_back(angle2[j] - angle1[i]);
        }
    }
    if (rot.empty()) return 0.f;  // PM.cc:174-177
    std::sort(rot.begin(), rot.end());
    return rot[(rot.size() - 1) / 2];
}

// ---- instrumentation ------------------------------------------------------------------------------------------------------------
int sdm_enable_stats(sdm_ctx* c, int on)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    c->stats_on = on != 0;
    return SDM_OK;
}

int sdm_selftest(sdm_ctx* c, int which, unsigned long long out[2])
{
    if (!c || !out) return fail(SDM_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(unsigned long long) * STATS_WORDS, c->stream));
    if (which == 0) {
        hipLaunchKernelGGL(k_selftest_div, dim3(4096), dim3(BLOCK), 0, c->stream, c->dprm.theta_var, c->dprm.inv_theta,
                           c->d_stats + 5);
    } else if (which == 1) {
        hipLaunchKernelGGL(k_selftest_chi, dim3(1024), dim3(BLOCK), 0, c->stream, 1024, c->d_stats + 5,
                           c->d_stats + 6);
    } else if (which == 2) {
        hipLaunchKernelGGL(k_selftest_cost, dim3(1024), dim3(BLOCK), 0, c->stream, c->dprm, 2048, c->d_stats + 5,
                           c->d_stats + 6);
    } else if (which == 4) {
        hipLaunchKernelGGL(k_selftest_fusion_terms, dim3(4096), dim3(BLOCK), 0, c->stream, 2048, c->d_stats + 5,
                           c->d_stats + 6);
    } else if (which == 5) {
        hipLaunchKernelGGL(k_selftest_quot, dim3(4096), dim3(BLOCK), 0, c->stream, 8192, c->d_stats + 5,
                           c->d_stats + 6);
    } else if (which == 7 || which == 9) {
        // scratch: one 3x2 {rho,sigma} patch and one constant block per thread
        const int blocks = 1024;
        float2* d_patch = nullptr;
        PairConst* d_pc = nullptr;
        HIP_TRY(hipMalloc(&d_patch, sizeof(float2) * 6 * blocks * BLOCK));
        if (hipMalloc(&d_pc, sizeof(PairConst) * blocks * BLOCK) != hipSuccess) {
            (void)hipFree(d_patch);
            return fail(SDM_EHIP, "selftest scratch allocation failed");
        }
        hipLaunchKernelGGL(k_selftest_k4, dim3(blocks), dim3(BLOCK), 0, c->stream, 2048, d_patch, d_pc, c->d_stats + 5,
                           c->d_stats + 6, which == 9 ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(out, c->d_stats + 5, sizeof(unsigned long long) * 2, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(d_patch);
        (void)hipFree(d_pc);
        if (e != hipSuccess) return fail(SDM_EHIP, hipGetErrorString(e));
        HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(unsigned long long) * STATS_WORDS, c->stream));
        return SDM_OK;
    } else if (which == 6) {
        hipLaunchKernelGGL(k_selftest_rcp, dim3(4096), dim3(BLOCK), 0, c->stream, c->d_stats + 5, c->d_stats + 6);
    } else if (which == 8) {
        hipLaunchKernelGGL(k_selftest_scan_ids, dim3(4096), dim3(BLOCK), 0, c->stream, c->d_stats + 5, c->d_stats + 6);
    } else if (which == 3) {
        hipLaunchKernelGGL(k_selftest_gates, dim3(4096), dim3(BLOCK), 0, c->stream, c->dprm, c->d_stats + 5, c->d_stats + 6);
    } else {
        return fail(SDM_EINVAL, "unknown selftest");
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->d_stats + 5, sizeof(unsigned long long) * 2, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(unsigned long long) * STATS_WORDS, c->stream));
    return SDM_OK;
}

int sdm_enable_timing(sdm_ctx* c, int on)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    c->timing_on = on != 0;
    return SDM_OK;
}

int sdm_get_timing(sdm_ctx* c, double ms_total[SDM_NUM_STAGES], long long launches[SDM_NUM_STAGES], int reset)
{
    if (!c || !ms_total || !launches) return fail(SDM_EINVAL, "null argument");
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < SDM_NUM_STAGES; i++) {
        ms_total[i] = 0.0;
        launches[i] = 0;
    }
    for (size_t i = 0; i < c->spans_used; i++) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->spans[i].a, c->spans[i].b));
        ms_total[c->spans[i].stage] += (double)ms;
        launches[c->spans[i].stage]++;
    }
    if (reset) c->spans_used = 0;
    return SDM_OK;
}

int sdm_set_ingest_overlap(sdm_ctx* c, int on)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (on && c->ing_bufs < sdm_ctx::ING_BUFS_MAX) {  // three 64-keyframe blocks' worth of chunk buffers
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->up_stream) HIP_TRY(hipStreamSynchronize(c->up_stream));
        const int rc = alloc_ingest_bufs(c, sdm_ctx::ING_BUFS_MAX);
        if (rc) return rc;
        c->ing_bufs = sdm_ctx::ING_BUFS_MAX;
    }
    c->ingest_overlap = on != 0;  // (switching it off keeps the buffers: they are in the ring's rotation)
    return SDM_OK;
}

int sdm_set_scan_mode(sdm_ctx* c, int mode)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (mode < 0 || mode > 2) return fail(SDM_EINVAL, "scan mode must be 0, 1 or 2");
    c->scan_mode = mode;
    set_dev_params(c);
    return SDM_OK;
}

int sdm_get_stats(sdm_ctx* c, sdm_stats* out, int reset)
{
    if (!c || !out) return fail(SDM_EINVAL, "null argument");
    unsigned long long v[STATS_WORDS];
    HIP_TRY(hipMemcpyAsync(v, c->d_stats, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    out->searches = (long long)v[0];
    out->candidates = (long long)v[1];
    out->gate_pass = (long long)v[2];
    out->hypotheses = (long long)v[3];
    out->fused = (long long)v[4];
    out->mask_waves = (long long)v[5];
    out->mask_steps = (long long)v[6];
    out->mask_row_mismatch = (long long)v[7];
    out->open_pixels = (long long)v[8];
    out->table_stagings = c->table_stagings;
    if (reset) c->table_stagings = 0;
    if (reset) HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(v), c->stream));
    return SDM_OK;
}

}  // extern "C"
