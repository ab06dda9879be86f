// sdm_comm.h -- the path's ONE exchange step between GPUs, native in the C ABI.
//
// InterKeyFrameDepthChecking of keyframe k (PM.cc:628-799) reads the FINISHED {rho,sigma} maps of k's
// covisible neighbours (the reference gates on that at PM.cc:292-298).  With keyframes sharded in
// contiguous blocks over the GPUs of a node, those maps cross GPUs once per pass, between K3 and K4:
//   halo       point-to-point ncclSend/ncclRecv of exactly the maps a peer's K4 reads (xGMI is
//              point-to-point: 2 x (N/2) x 8P bytes per rank over two direct links, independent of world
//              size), issued on a second stream behind an event so it overlaps the interior keyframes' K1-K3;
//   all-gather ncclAllGather of every rank's block (BASELINE.json's wording), in place in the depth pool when
//              slot == global keyframe index, or into an engine-owned buffer from which the needed maps are
//              copied to local slots.
// RCCL is resolved with dlopen at the first sdm_comm_* call: a single-GPU user never loads it, and inside a
// process that already holds torch's librccl.so.1 the same library instance is used.
// Included by sdm_engine.hip after sdm_ctx is defined (one translation unit).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
};
RcclApi g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return SDM_OK;
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return fail(SDM_ECOMM, std::string("cannot load RCCL (librccl.so.1): ") + dlerror());
    RcclApi a;
    a.handle = h;
#define SDM_RCCL_SYM(field, name)                                                         \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                        \
    if (!a.field) return fail(SDM_ECOMM, std::string("RCCL symbol missing: ") + name)
    SDM_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    SDM_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    SDM_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    SDM_RCCL_SYM(GetErrorString, "ncclGetErrorString");
    SDM_RCCL_SYM(GroupStart, "ncclGroupStart");
    SDM_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    SDM_RCCL_SYM(Send, "ncclSend");
    SDM_RCCL_SYM(Recv, "ncclRecv");
    SDM_RCCL_SYM(AllGather, "ncclAllGather");
    SDM_RCCL_SYM(AllReduce, "ncclAllReduce");
    SDM_RCCL_SYM(CommCount, "ncclCommCount");
    SDM_RCCL_SYM(CommUserRank, "ncclCommUserRank");
#undef SDM_RCCL_SYM
    g_rccl = a;
    return SDM_OK;
}

#define RCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r__ = (expr);                                                              \
        if (r__ != ncclSuccess)                                                                 \
            return fail(SDM_ECOMM, std::string(#expr) + ": " + g_rccl.GetErrorString(r__));     \
    } while (0)

// second stream + the two events that order it against the compute stream
int comm_streams(sdm_ctx* c)
{
    if (c->comm_stream) return SDM_OK;
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_maps_ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_xchg_done, hipEventDisableTiming));
    return SDM_OK;
}

void comm_release(sdm_ctx* c)
{
    // nothing of this context may still be running on the communicator when it is destroyed: the exchange stream and
    // the compute stream (the all-gather forms run on it; hipStreamSynchronize(nullptr) covers a caller's null stream)
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm && c->own_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
    c->own_comm = false;
    c->world = 1;
    c->rank = 0;
    if (c->comm_stream) {
        (void)hipStreamSynchronize(c->comm_stream);
        (void)hipStreamDestroy(c->comm_stream);
        c->comm_stream = nullptr;
    }
    if (c->ev_maps_ready) (void)hipEventDestroy(c->ev_maps_ready);
    if (c->ev_xchg_done) (void)hipEventDestroy(c->ev_xchg_done);
    c->ev_maps_ready = c->ev_xchg_done = nullptr;
    (void)hipFree(c->gather_buf);
    c->gather_buf = nullptr;
    c->gather_slots = 0;
    (void)hipFree(c->d_agree);
    c->d_agree = nullptr;
    (void)hipFree(c->stage_buf);
    c->stage_buf = nullptr;
    c->stage_slots = 0;
    (void)hipFree(c->d_xchg_mismatch);
    c->d_xchg_mismatch = nullptr;
    c->xchg_entries = 0;
    c->xchg_pending = false;
    c->ag_open = false;
    c->ag_pieces.clear();
}

// Up to COPY_BATCH whole {rho,sigma} maps moved by ONE launch (blockIdx.y = map): the exchange's packing and
// fetch steps move ~N maps per pass, and N back-to-back hipMemcpyAsync calls cost more in launch gaps than in bytes.
constexpr int COPY_BATCH = 32;
struct CopyBatch {
    const float4* src[COPY_BATCH];
    float4* dst[COPY_BATCH];
};
__global__ __launch_bounds__(BLOCK) void k_copy_maps(CopyBatch b, long long n16 /* 16-byte units per map */)
{
    const float4* __restrict__ s = b.src[blockIdx.y];
    float4* __restrict__ d = b.dst[blockIdx.y];
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += (long long)gridDim.x * BLOCK) d[i] = s[i];
}
// maps are P float2: 16-byte units when P is even (every supported W is), plain copies otherwise
int copy_maps(sdm_ctx* c, int n, const float2* const* src, float2* const* dst, hipStream_t stream)
{
    if ((c->P & 1) != 0) {
        for (int i = 0; i < n; i++)
            HIP_TRY(hipMemcpyAsync(dst[i], src[i], sizeof(float2) * c->P, hipMemcpyDeviceToDevice, stream));
        return SDM_OK;
    }
    const long long n16 = (long long)c->P / 2;
    const unsigned gx = (unsigned)std::min<long long>((n16 + BLOCK - 1) / BLOCK, 256);
    for (int first = 0; first < n; first += COPY_BATCH) {
        const int m = std::min(COPY_BATCH, n - first);
        CopyBatch b;
        for (int i = 0; i < COPY_BATCH; i++) {
            b.src[i] = reinterpret_cast<const float4*>(src[first + (i < m ? i : 0)]);
            b.dst[i] = reinterpret_cast<float4*>(dst[first + (i < m ? i : 0)]);
        }
        hipLaunchKernelGGL(k_copy_maps, dim3(gx, (unsigned)m), dim3(BLOCK), 0, stream, b, n16);
        HIP_TRY(hipGetLastError());
    }
    return SDM_OK;
}

// ---- compact transport (sdm_exchange_compact) ---------------------------------------------------------------------------
// A reconstructed map is zero outside its keyframe's active-pixel list (PM.cc:201), and a rank that reads another rank's
// map holds that keyframe's image -- hence the same list -- as part of its input halo.  So a map can cross ranks as the
// {rho,sigma} of its list entries, in list order: ~19 % of the pixels on the App. D scene, a fixed `entries` per map on the
// wire (the list's tail is padding).  Sender: k_pack_lists gathers them; receiver: k_unpack_lists scatters them through
// ITS list into a plane that is (or is first made) zero elsewhere.
// A packed map is `entries` values followed by a header (XCHG_HEADER float2, one cache line): the sender's list length and the
// 64-bit hash of its list (the sum over the listed pixels of a mix of (y << 16 | x), kept per slot since the list was built:
// sdm_ingest.h).  A receiver whose list of that keyframe differs in length or hash (its image differs from the sender's) does
// not scatter the map -- it would land on the wrong pixels -- and counts the mismatch (sdm_exchange_mismatches).
constexpr int XCHG_HEADER = 8;
struct ListBatch {
    int slot[COPY_BATCH];
    float2* buf[COPY_BATCH];
};
__global__ __launch_bounds__(BLOCK) void k_pack_lists(ListBatch b, const float2* __restrict__ pool, long long plane, int W,
                                                      const unsigned* __restrict__ act, const int* __restrict__ act_count,
                                                      const unsigned long long* __restrict__ act_hash, int entries)
{
    const int slot = b.slot[blockIdx.y];
    float2* __restrict__ out = b.buf[blockIdx.y];
    const int n = act_count[slot];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long h = act_hash[slot];
        out[entries] = make_float2(__int_as_float(n), __uint_as_float((unsigned)h));
        out[entries + 1] = make_float2(__uint_as_float((unsigned)(h >> 32)), 0.0f);
    }
    const unsigned* __restrict__ list = act + (long long)slot * plane;
    const float2* __restrict__ map = pool + (long long)slot * plane;
    // the list's entries, then zeros up to `entries`: the payload is a function of the map alone (the NumPy statement of the
    // format, shard.pack_compact, pads with zeros as well: byte-for-byte equal payloads)
    for (int t = blockIdx.x * BLOCK + threadIdx.x; t < entries; t += gridDim.x * BLOCK) {
        float2 v = make_float2(0.f, 0.f);
        if (t < n) {
            const unsigned xy = list[t];
            v = map[(int)(xy >> 16) * W + (int)(xy & 0xffffu)];
        }
        out[t] = v;
    }
}
__global__ __launch_bounds__(BLOCK) void k_unpack_lists(ListBatch b, float2* __restrict__ pool, long long plane, int W,
                                                        const unsigned* __restrict__ act, const int* __restrict__ act_count,
                                                        const unsigned long long* __restrict__ act_hash, int entries,
                                                        unsigned* __restrict__ mismatches)
{
    const int slot = b.slot[blockIdx.y];
    const float2* __restrict__ in = b.buf[blockIdx.y];
    const int n = act_count[slot];
    const unsigned long long h = act_hash[slot];
    const float2 h0 = in[entries], h1 = in[entries + 1];
    if (__float_as_int(h0.x) != n || __float_as_uint(h0.y) != (unsigned)h ||
        __float_as_uint(h1.x) != (unsigned)(h >> 32)) {  // the same for every thread of the map
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(mismatches, 1u);
        return;
    }
    const unsigned* __restrict__ list = act + (long long)slot * plane;
    float2* __restrict__ map = pool + (long long)slot * plane;
    for (int t = blockIdx.x * BLOCK + threadIdx.x; t < n; t += gridDim.x * BLOCK) {
        const unsigned xy = list[t];
        map[(int)(xy >> 16) * W + (int)(xy & 0xffffu)] = in[t];
    }
}
int list_kernel(sdm_ctx* c, bool pack, int n, const int* slots, float2* const* bufs, hipStream_t stream)
{
    const unsigned gx = (unsigned)std::min<long long>(((long long)c->xchg_entries + BLOCK - 1) / BLOCK, 256);
    for (int first = 0; first < n; first += COPY_BATCH) {
        const int m = std::min(COPY_BATCH, n - first);
        ListBatch b;
        for (int i = 0; i < COPY_BATCH; i++) {
            b.slot[i] = slots[first + (i < m ? i : 0)];
            b.buf[i] = bufs[first + (i < m ? i : 0)];
        }
        if (pack)
            hipLaunchKernelGGL(k_pack_lists, dim3(gx, (unsigned)m), dim3(BLOCK), 0, stream, b, c->pool, c->P, c->W, c->d_act,
                               c->d_act_count, c->d_act_hash, c->xchg_entries);
        else
            hipLaunchKernelGGL(k_unpack_lists, dim3(gx, (unsigned)m), dim3(BLOCK), 0, stream, b, c->pool, c->P, c->W, c->d_act,
                               c->d_act_count, c->d_act_hash, c->xchg_entries, c->d_xchg_mismatch);
        HIP_TRY(hipGetLastError());
    }
    return SDM_OK;
}
// the slots whose maps leave in compact form: reconstructed here under the current lambdaG (zero outside the list), and the
// list fits the wire format
int check_compact_sources(sdm_ctx* c, int n, const int* slots)
{
    int rc = sync_counts(c);
    if (rc) return rc;
    for (int i = 0; i < n; i++) {
        if (!(c->recon_lambdaG[slots[i]] == c->dprm.lambdaG) || !(c->act_lambdaG[slots[i]] == c->dprm.lambdaG))
            return fail(SDM_ESTATE, "compact exchange: a source map is not a pipeline map (zero outside its active list)");
        if (c->h_act_count[slots[i]] > c->xchg_entries)
            return fail(SDM_ESTATE, "compact exchange: a source keyframe's active list is longer than entries_per_map");
    }
    return SDM_OK;
}
// the slots that receive compact maps: the keyframe (image -> list) must be resident; the plane is zeroed first unless it
// already is zero outside the list.  `stream` is the exchange stream, ordered behind everything queued so far.
int prepare_compact_destinations(sdm_ctx* c, int n, const int* slots, hipStream_t stream)
{
    int rc;
    bool built = false;
    for (int i = 0; i < n; i++) {
        if ((rc = check_slot(c, slots[i], true))) return rc;
        if (!(c->act_lambdaG[slots[i]] == c->dprm.lambdaG)) {
            if ((rc = build_active(c, slots[i]))) return rc;
            built = true;
        }
    }
    if (built) HIP_TRY(hipStreamSynchronize(c->stream));  // first use only: the lists must exist before the exchange stream reads them
    if ((rc = sync_counts(c))) return rc;
    for (int i = 0; i < n; i++) {
        if (c->h_act_count[slots[i]] > c->xchg_entries)
            return fail(SDM_ESTATE, "compact exchange: a destination keyframe's active list is longer than entries_per_map");
        if (!(c->recon_lambdaG[slots[i]] == c->dprm.lambdaG))
            HIP_TRY(hipMemsetAsync(c->pool + (long long)slots[i] * c->P, 0, sizeof(float2) * c->P, stream));
    }
    return SDM_OK;
}
void mark_received(sdm_ctx* c, int slot)
{
    c->has_depth[slot] = 1;  // a peer's finished map (semidense_flag_, PM.cc:294)
    // whole map: this rank did not reconstruct it (no claim about its support); compact: values at this keyframe's list
    // pixels, zero elsewhere -- a pipeline map like the sender's
    c->recon_lambdaG[slot] = c->xchg_entries > 0 ? c->dprm.lambdaG : std::nanf("");
}
// floats2 per map on the wire and in the staging / landing buffers
long long xchg_stride(const sdm_ctx* c) { return c->xchg_entries > 0 ? (long long)c->xchg_entries + XCHG_HEADER : c->P; }

// staging (outgoing) and landing (incoming) buffers of at least this many maps of the current wire format
int ensure_xchg_buffers(sdm_ctx* c, long long stage_maps, long long gather_maps)
{
    const long long M = xchg_stride(c);
    if (c->stage_slots < stage_maps) {
        if (c->comm_stream) HIP_TRY(hipStreamSynchronize(c->comm_stream));
        (void)hipFree(c->stage_buf);
        c->stage_buf = nullptr;
        c->stage_slots = 0;
        HIP_TRY(hipMalloc((void**)&c->stage_buf, sizeof(float2) * (size_t)M * (size_t)stage_maps));
        c->stage_slots = (int)stage_maps;
    }
    if (c->gather_slots < gather_maps) {
        HIP_TRY(hipStreamSynchronize(c->stream));  // earlier fetch copies may still read the old buffer
        if (c->comm_stream) HIP_TRY(hipStreamSynchronize(c->comm_stream));
        (void)hipFree(c->gather_buf);
        c->gather_buf = nullptr;
        c->gather_slots = 0;
        HIP_TRY(hipMalloc((void**)&c->gather_buf, sizeof(float2) * (size_t)M * (size_t)gather_maps));
        c->gather_slots = gather_maps;
    }
    return SDM_OK;
}

int check_xchg_slots(sdm_ctx* c, int n, const int* peer, const int* slot, const char* what)
{
    if (n < 0 || (n > 0 && (!peer || !slot))) return fail(SDM_EINVAL, std::string("bad ") + what + " list");
    for (int i = 0; i < n; i++) {
        if (slot[i] < 0 || slot[i] >= c->cfg.max_keyframes) return fail(SDM_EINVAL, std::string(what) + " slot out of range");
        // (a one-rank communicator -- the hardware rehearsal of sdm_comm_init -- has only itself to talk to)
        if (peer[i] < 0 || peer[i] >= c->world || (peer[i] == c->rank && c->world > 1))
            return fail(SDM_EINVAL, std::string(what) + " peer out of range (or self)");
    }
    return SDM_OK;
}

// a slot may be received into once, and never while it is also being sent (the transfers of one group are unordered)
int check_xchg_overlap(sdm_ctx* c, int n_send, const int* send_slot, int n_recv, const int* recv_slot)
{
    std::vector<char> seen((size_t)c->cfg.max_keyframes, 0);
    for (int i = 0; i < n_send; i++) seen[send_slot[i]] = 1;  // one map may go to several peers
    for (int i = 0; i < n_recv; i++) {
        if (seen[recv_slot[i]] == 1) return fail(SDM_EINVAL, "slot listed both as send and recv");
        if (seen[recv_slot[i]] == 2) return fail(SDM_EINVAL, "duplicate recv slot");
        seen[recv_slot[i]] = 2;
    }
    return SDM_OK;
}

}  // namespace

extern "C" {

int sdm_comm_unique_id(unsigned char id[SDM_COMM_ID_BYTES])
{
    static_assert(SDM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    if (!id) return fail(SDM_EINVAL, "null id");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId u;
    RCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, SDM_COMM_ID_BYTES);
    return SDM_OK;
}

int sdm_comm_init(sdm_ctx* c, const unsigned char id[SDM_COMM_ID_BYTES], int world, int rank)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (world < 1 || rank < 0 || rank >= world) return fail(SDM_EINVAL, "bad world/rank");
    if (c->comm || c->world != 1) return fail(SDM_ESTATE, "context already has a communicator");
    // world 1: nothing to exchange, no communicator, every exchange call is a no-op or a device copy -- unless
    // SDM_COMM_SINGLE_RANK_RCCL=1 asks for a real one-rank communicator: then every RCCL call of this file runs for real
    // (an all-gather over one rank, sends to itself), which is how the transport is rehearsed on a one-GPU box
    // (tests/test_gpu_comm.py)
    const char* rehearse = getenv("SDM_COMM_SINGLE_RANK_RCCL");
    if (world == 1 && !(rehearse && atoi(rehearse) == 1)) return SDM_OK;
    if (!id) return fail(SDM_EINVAL, "null id");
    int rc = load_rccl();
    if (rc) return rc;
    if ((rc = comm_streams(c))) return rc;
    ncclUniqueId u;
    memcpy(u.internal, id, SDM_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    HIP_TRY(hipSetDevice(c->cfg.device));
    RCCL_TRY(g_rccl.CommInitRank(&comm, world, u, rank));
    c->comm = comm;
    c->own_comm = true;
    c->world = world;
    c->rank = rank;
    return SDM_OK;
}

int sdm_comm_attach(sdm_ctx* c, void* nccl_comm)
{
    if (!c || !nccl_comm) return fail(SDM_EINVAL, "null argument");
    if (c->comm || c->world != 1) return fail(SDM_ESTATE, "context already has a communicator");
    int rc = load_rccl();
    if (rc) return rc;
    int world = 0, rank = 0;
    RCCL_TRY(g_rccl.CommCount((ncclComm_t)nccl_comm, &world));
    RCCL_TRY(g_rccl.CommUserRank((ncclComm_t)nccl_comm, &rank));
    if ((rc = comm_streams(c))) return rc;
    c->comm = nccl_comm;
    c->own_comm = false;
    c->world = world;
    c->rank = rank;
    return SDM_OK;
}

int sdm_comm_destroy(sdm_ctx* c)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (c->stream) HIP_TRY(hipStreamSynchronize(c->stream));
    comm_release(c);
    return SDM_OK;
}

int sdm_comm_info(sdm_ctx* c, int* world, int* rank)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    return SDM_OK;
}

// Starts the halo exchange: everything queued on the context's stream so far (the boundary keyframes'
// K1-K3) is finished before the maps leave; the transfers run on the exchange stream, so work queued
// on the context's stream after this call (the interior keyframes) overlaps them.
int sdm_exchange_halo_begin(sdm_ctx* c, int n_send, const int* send_peer, const int* send_slot, int n_recv,
                            const int* recv_peer, const int* recv_slot)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (c->xchg_pending) return fail(SDM_ESTATE, "an exchange is already in flight: call sdm_exchange_wait first");
    if (!c->comm) {
        if (n_send || n_recv) return fail(SDM_EINVAL, "world size 1 has no peers");
        return SDM_OK;
    }
    int rc;
    if ((rc = check_xchg_slots(c, n_send, send_peer, send_slot, "send"))) return rc;
    if ((rc = check_xchg_slots(c, n_recv, recv_peer, recv_slot, "recv"))) return rc;
    if ((rc = check_xchg_overlap(c, n_send, send_slot, n_recv, recv_slot))) return rc;
    for (int i = 0; i < n_send; i++)
        if (!c->has_depth[send_slot[i]]) return fail(SDM_ESTATE, "send slot has no reconstructed depth map");
    HIP_TRY(hipSetDevice(c->cfg.device));
    const bool compact = c->xchg_entries > 0;
    const long long M = xchg_stride(c);
    if (compact) {
        if ((rc = check_compact_sources(c, n_send, send_slot))) return rc;
        if ((rc = ensure_xchg_buffers(c, n_send, n_recv))) return rc;
    }
    HIP_TRY(hipEventRecord(c->ev_maps_ready, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_maps_ready, 0));
    if (compact) {
        if ((rc = prepare_compact_destinations(c, n_recv, recv_slot, c->comm_stream))) return rc;
        std::vector<float2*> bufs((size_t)n_send);
        for (int i = 0; i < n_send; i++) bufs[i] = c->stage_buf + (long long)i * M;
        if ((rc = list_kernel(c, true, n_send, send_slot, bufs.data(), c->comm_stream))) return rc;
    }
    const size_t count = (size_t)M * 2;  // floats per map {rho,sigma} on the wire
    ncclComm_t comm = (ncclComm_t)c->comm;
    // one group: all sends and receives of this rank progress together (no ordering deadlock between peers).  A
    // failing call must not leave the thread's group open (every later RCCL call of the thread, torch's included,
    // would be queued into it and never launched): remember the first failure, close the group, then report it.
    RCCL_TRY(g_rccl.GroupStart());
    ncclResult_t first = ncclSuccess;
    const char* what = "";
    for (int i = 0; i < n_send && first == ncclSuccess; i++) {
        const float2* src = compact ? c->stage_buf + (long long)i * M : c->pool + (long long)send_slot[i] * c->P;
        first = g_rccl.Send(src, count, ncclFloat, send_peer[i], comm, c->comm_stream);
        what = "ncclSend";
    }
    for (int i = 0; i < n_recv && first == ncclSuccess; i++) {
        float2* dst = compact ? c->gather_buf + (long long)i * M : c->pool + (long long)recv_slot[i] * c->P;
        first = g_rccl.Recv(dst, count, ncclFloat, recv_peer[i], comm, c->comm_stream);
        what = "ncclRecv";
    }
    const ncclResult_t ended = g_rccl.GroupEnd();
    if (first != ncclSuccess) return fail(SDM_ECOMM, std::string(what) + ": " + g_rccl.GetErrorString(first));
    if (ended != ncclSuccess) return fail(SDM_ECOMM, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ended));
    if (compact) {
        std::vector<float2*> bufs((size_t)n_recv);
        for (int i = 0; i < n_recv; i++) bufs[i] = c->gather_buf + (long long)i * M;
        if ((rc = list_kernel(c, false, n_recv, recv_slot, bufs.data(), c->comm_stream))) return rc;
    }
    HIP_TRY(hipEventRecord(c->ev_xchg_done, c->comm_stream));
    for (int i = 0; i < n_recv; i++) mark_received(c, recv_slot[i]);
    c->xchg_pending = true;
    return SDM_OK;
}

// Work queued on the context's stream after this call (K4) sees the received maps.  No host wait.
int sdm_exchange_wait(sdm_ctx* c)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (!c->xchg_pending) return SDM_OK;
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_xchg_done, 0));
    c->xchg_pending = false;
    return SDM_OK;
}

int sdm_exchange_halo(sdm_ctx* c, int n_send, const int* send_peer, const int* send_slot, int n_recv,
                      const int* recv_peer, const int* recv_slot)
{
    int rc = sdm_exchange_halo_begin(c, n_send, send_peer, send_slot, n_recv, recv_peer, recv_slot);
    if (rc) return rc;
    return sdm_exchange_wait(c);
}

// All-gather of every rank's block of `count` maps starting at local slot `first_slot`.
//  n_fetch < 0   in place: the pool holds world*count slots, slot == global keyframe index, and this
//                rank's block sits at first_slot == rank*count (the SURVEY §8b `sdm_allgather` sketch);
//  n_fetch >= 0  gathered into an engine-owned buffer [world][count] maps; map fetch_index[i] of that
//                sequence (= owner_rank*count + position) is then copied into local slot dst_slot[i].
// Stream-ordered on the context's stream; no host wait.
int sdm_allgather_depth(sdm_ctx* c, int first_slot, int count, int n_fetch, const int* fetch_index,
                        const int* dst_slot)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (count < 1 || first_slot < 0 || first_slot + count > c->cfg.max_keyframes)
        return fail(SDM_EINVAL, "block out of range");
    if (n_fetch > 0 && (!fetch_index || !dst_slot)) return fail(SDM_EINVAL, "null fetch list");
    if (c->xchg_entries > 0)
        return fail(SDM_ESTATE, "the one-shot all-gather moves whole maps: sdm_exchange_compact(ctx, 0) first, or use the pieces");
    const long long total = (long long)c->world * count;
    for (int i = 0; i < n_fetch; i++) {
        if (fetch_index[i] < 0 || fetch_index[i] >= total) return fail(SDM_EINVAL, "fetch index out of range");
        if (dst_slot[i] < 0 || dst_slot[i] >= c->cfg.max_keyframes) return fail(SDM_EINVAL, "fetch slot out of range");
        if (dst_slot[i] >= first_slot && dst_slot[i] < first_slot + count)
            return fail(SDM_EINVAL, "fetch would overwrite this rank's own block");
    }
    if (n_fetch > 0) {
        std::vector<char> seen((size_t)c->cfg.max_keyframes, 0);
        for (int i = 0; i < n_fetch; i++) {
            if (seen[dst_slot[i]]) return fail(SDM_EINVAL, "duplicate fetch slot");
            seen[dst_slot[i]] = 1;
        }
    }
    if (n_fetch < 0 && (total > c->cfg.max_keyframes || first_slot != c->rank * count))
        return fail(SDM_EINVAL, "in-place all-gather needs slot == global keyframe index");
    for (int r = 0; r < count; r++)
        if (!c->has_depth[first_slot + r]) return fail(SDM_ESTATE, "block slot has no reconstructed depth map");
    if (!c->comm) return SDM_OK;  // this rank's block is all there is
    HIP_TRY(hipSetDevice(c->cfg.device));
    const size_t block_floats = (size_t)count * (size_t)c->P * 2;
    ncclComm_t comm = (ncclComm_t)c->comm;
    if (n_fetch < 0) {
        RCCL_TRY(g_rccl.AllGather(c->pool + (long long)first_slot * c->P, c->pool, block_floats, ncclFloat, comm,
                                  c->stream));
        for (long long s = 0; s < total; s++)
            if (s < first_slot || s >= first_slot + count) {
                c->has_depth[s] = 1;
                c->recon_lambdaG[s] = std::nanf("");
            }
        return SDM_OK;
    }
    if (c->gather_slots < total) {
        (void)hipFree(c->gather_buf);
        c->gather_buf = nullptr;
        c->gather_slots = 0;
        HIP_TRY(hipMalloc((void**)&c->gather_buf, sizeof(float2) * (size_t)c->P * (size_t)total));
        c->gather_slots = total;
    }
    RCCL_TRY(g_rccl.AllGather(c->pool + (long long)first_slot * c->P, c->gather_buf, block_floats, ncclFloat, comm,
                              c->stream));
    std::vector<const float2*> srcs((size_t)n_fetch);
    std::vector<float2*> dsts((size_t)n_fetch);
    for (int i = 0; i < n_fetch; i++) {
        srcs[i] = c->gather_buf + (long long)fetch_index[i] * c->P;
        dsts[i] = c->pool + (long long)dst_slot[i] * c->P;
    }
    int rc = copy_maps(c, n_fetch, srcs.data(), dsts.data(), c->stream);
    if (rc) return rc;
    for (int i = 0; i < n_fetch; i++) {
        c->has_depth[dst_slot[i]] = 1;
        c->recon_lambdaG[dst_slot[i]] = std::nanf("");
    }
    return SDM_OK;
}

// ---- all-gather in pieces, overlapped with the reconstruction ------------------------------------------------------
// Every rank contributes the same number of maps (`maps_per_rank`), in pieces: a piece is a list of local slots whose
// K1-K3 have just been queued; it is gathered on the exchange stream (behind an event) while the next keyframes are
// reconstructed on the compute stream -- the same two-event scheme as the halo form.  Two uses:
//   * the whole block in a few sub-blocks (BASELINE.json's literal "all-gather of the per-keyframe maps");
//   * only the keyframes that some other rank's K4 reads (the boundary keyframes of an index-local covisibility graph),
//     padded to a common count -- the same collective over a third of the bytes, issued before the interior keyframes
//     are reconstructed.
// Piece i = `count` maps lands in the gather buffer at [piece base][rank][count] (piece base = world * maps before it);
// _finish makes the compute stream wait for the last piece and copies map fetch_index[i] = owner_rank * maps_per_rank +
// position (position = index in the owner's contribution order) into local slot dst_slot[i].  A piece of consecutive
// slots is gathered straight from the depth pool; any other list is packed into a staging buffer first (device copies
// on the exchange stream).  With world == 1 the "gather" is a device copy, so that the bookkeeping and the fetch
// addressing run (and are tested) on one GPU, too.
int sdm_allgather_begin(sdm_ctx* c, int maps_per_rank)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (c->ag_open) return fail(SDM_ESTATE, "an all-gather is already open: call sdm_allgather_finish first");
    if (c->xchg_pending) return fail(SDM_ESTATE, "an exchange is in flight: call sdm_exchange_wait first");
    if (maps_per_rank < 1) return fail(SDM_EINVAL, "maps_per_rank < 1");
    HIP_TRY(hipSetDevice(c->cfg.device));
    int rc = comm_streams(c);
    if (rc) return rc;
    const long long total = (long long)c->world * maps_per_rank;
    if ((rc = ensure_xchg_buffers(c, 0, total))) return rc;
    // the previous pass's fetch copies (compute stream) must have left the buffer before new pieces land in it
    HIP_TRY(hipEventRecord(c->ev_maps_ready, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_maps_ready, 0));
    c->ag_open = true;
    c->ag_count = maps_per_rank;
    c->ag_covered = 0;
    c->ag_pieces.clear();
    c->ag_contributed.assign((size_t)c->cfg.max_keyframes, 0);
    return SDM_OK;
}

int sdm_allgather_piece(sdm_ctx* c, int count, const int* slots)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (!c->ag_open) return fail(SDM_ESTATE, "sdm_allgather_begin first");
    if (count < 1 || !slots || c->ag_covered + count > c->ag_count)
        return fail(SDM_EINVAL, "piece exceeds the maps_per_rank announced to sdm_allgather_begin");
    bool contiguous = true;
    for (int i = 0; i < count; i++) {
        if (slots[i] < 0 || slots[i] >= c->cfg.max_keyframes) return fail(SDM_EINVAL, "piece slot out of range");
        if (!c->has_depth[slots[i]]) return fail(SDM_ESTATE, "piece slot has no reconstructed depth map");
        contiguous = contiguous && slots[i] == slots[0] + i;
    }
    HIP_TRY(hipSetDevice(c->cfg.device));
    int rc;
    const bool compact = c->xchg_entries > 0;
    const long long M = xchg_stride(c);
    if (compact && (rc = check_compact_sources(c, count, slots))) return rc;
    if ((compact || !contiguous) && (rc = ensure_xchg_buffers(c, c->ag_count, 0))) return rc;
    HIP_TRY(hipEventRecord(c->ev_maps_ready, c->stream));  // this piece's K1-K3 are queued on the compute stream
    HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_maps_ready, 0));
    const int offset = c->ag_covered;
    const float2* src = c->pool + (long long)slots[0] * c->P;
    if (compact) {  // the listed pixels of every map of the piece, in list order
        std::vector<float2*> bufs((size_t)count);
        for (int i = 0; i < count; i++) bufs[i] = c->stage_buf + (long long)(offset + i) * M;
        if ((rc = list_kernel(c, true, count, slots, bufs.data(), c->comm_stream))) return rc;
        src = c->stage_buf + (long long)offset * M;
    } else if (!contiguous) {
        std::vector<const float2*> srcs((size_t)count);
        std::vector<float2*> dsts((size_t)count);
        for (int i = 0; i < count; i++) {
            srcs[i] = c->pool + (long long)slots[i] * c->P;
            dsts[i] = c->stage_buf + (long long)(offset + i) * c->P;
        }
        if ((rc = copy_maps(c, count, srcs.data(), dsts.data(), c->comm_stream))) return rc;
        src = c->stage_buf + (long long)offset * c->P;
    }
    const size_t piece_floats = (size_t)count * (size_t)M * 2;
    float2* dst = c->gather_buf + (long long)c->world * offset * M;  // pieces before this one hold world*offset maps
    if (!c->comm) {
        HIP_TRY(hipMemcpyAsync(dst, src, piece_floats * sizeof(float), hipMemcpyDeviceToDevice, c->comm_stream));
    } else {
        RCCL_TRY(g_rccl.AllGather(src, dst, piece_floats, ncclFloat, (ncclComm_t)c->comm, c->comm_stream));
    }
    for (int i = 0; i < count; i++) c->ag_contributed[slots[i]] = 1;
    c->ag_pieces.push_back({offset, count});
    c->ag_covered = offset + count;
    return SDM_OK;
}

int sdm_allgather_finish(sdm_ctx* c, int n_fetch, const int* fetch_index, const int* dst_slot)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (!c->ag_open) return fail(SDM_ESTATE, "sdm_allgather_begin first");
    if (c->ag_covered != c->ag_count) return fail(SDM_ESTATE, "the pieces do not add up to maps_per_rank yet");
    if (n_fetch < 0 || (n_fetch > 0 && (!fetch_index || !dst_slot))) return fail(SDM_EINVAL, "bad fetch list");
    const long long total = (long long)c->world * c->ag_count;
    std::vector<char> seen((size_t)c->cfg.max_keyframes, 0);
    for (int i = 0; i < n_fetch; i++) {
        if (fetch_index[i] < 0 || fetch_index[i] >= total) return fail(SDM_EINVAL, "fetch index out of range");
        if (dst_slot[i] < 0 || dst_slot[i] >= c->cfg.max_keyframes) return fail(SDM_EINVAL, "fetch slot out of range");
        if (c->ag_contributed[dst_slot[i]]) return fail(SDM_EINVAL, "fetch would overwrite a map this rank contributed");
        if (seen[dst_slot[i]]) return fail(SDM_EINVAL, "duplicate fetch slot");
        seen[dst_slot[i]] = 1;
    }
    HIP_TRY(hipSetDevice(c->cfg.device));
    // the fetch copies run on the EXCHANGE stream behind the last piece; the compute stream only waits for them.  What
    // the caller queued on the compute stream before this call (the interior keyframes' K1-K3 and K4) neither reads nor
    // writes the destination slots' maps, so the copies overlap it.
    const bool compact = c->xchg_entries > 0;
    const long long M = xchg_stride(c);
    std::vector<const float2*> srcs((size_t)n_fetch);
    std::vector<float2*> dsts((size_t)n_fetch);
    for (int i = 0; i < n_fetch; i++) {
        const int owner = fetch_index[i] / c->ag_count, pos = fetch_index[i] - owner * c->ag_count;
        const float2* src = nullptr;
        for (const auto& pc : c->ag_pieces)
            if (pos >= pc.offset && pos < pc.offset + pc.count)
                src = c->gather_buf + ((long long)c->world * pc.offset + (long long)owner * pc.count + (pos - pc.offset)) * M;
        srcs[i] = src;
        dsts[i] = c->pool + (long long)dst_slot[i] * c->P;
    }
    int rc;
    if (compact) {
        if ((rc = prepare_compact_destinations(c, n_fetch, dst_slot, c->comm_stream))) return rc;
        std::vector<float2*> bufs((size_t)n_fetch);
        for (int i = 0; i < n_fetch; i++) bufs[i] = const_cast<float2*>(srcs[i]);
        if ((rc = list_kernel(c, false, n_fetch, dst_slot, bufs.data(), c->comm_stream))) return rc;
    } else if ((rc = copy_maps(c, n_fetch, srcs.data(), dsts.data(), c->comm_stream))) {
        return rc;
    }
    HIP_TRY(hipEventRecord(c->ev_xchg_done, c->comm_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_xchg_done, 0));
    for (int i = 0; i < n_fetch; i++) mark_received(c, dst_slot[i]);
    c->ag_open = false;
    return SDM_OK;
}

// Wire format of the maps that cross ranks (halo and all-gather-in-pieces forms; the one-shot sdm_allgather_depth always
// moves whole maps).  entries_per_map = 0: whole maps, 8P bytes each.  > 0: the {rho,sigma} of the first entries_per_map
// entries of the keyframe's active-pixel list, in list order -- everything a reconstructed map holds (it is zero outside
// the list), 8 * entries_per_map bytes.  Every rank must set the SAME value, no shorter than the longest list among the
// keyframes it sends or receives (sdm_active_count; the call that would send or receive a longer one fails with
// SDM_ESTATE before anything is posted -- agree on the value across ranks first, as bench.py does).  The receiver must
// hold the keyframe (its image, hence its list) in the destination slot: true for the input halo of a sharded sequence.
int sdm_exchange_compact(sdm_ctx* c, int entries_per_map)
{
    if (!c) return fail(SDM_EINVAL, "null context");
    if (entries_per_map < 0 || entries_per_map > c->P) return fail(SDM_EINVAL, "entries_per_map out of range");
    if (c->xchg_pending || c->ag_open) return fail(SDM_ESTATE, "an exchange is in flight");
    if (entries_per_map == c->xchg_entries) return SDM_OK;
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->comm_stream) HIP_TRY(hipStreamSynchronize(c->comm_stream));
    (void)hipFree(c->gather_buf);  // sized in maps of the old format
    c->gather_buf = nullptr;
    c->gather_slots = 0;
    (void)hipFree(c->stage_buf);
    c->stage_buf = nullptr;
    c->stage_slots = 0;
    if (entries_per_map > 0 && !c->d_xchg_mismatch) {
        HIP_TRY(hipMalloc((void**)&c->d_xchg_mismatch, sizeof(unsigned)));
        HIP_TRY(hipMemset(c->d_xchg_mismatch, 0, sizeof(unsigned)));
    }
    c->xchg_entries = entries_per_map;
    return SDM_OK;
}

// The compact wire format through HOST memory, one map per call (host-blocking): what k_pack_lists would put on the wire for
// `slot` -> out[2 * (entries_per_map + 8)] floats, and the receiving side for a payload that arrived by any other route
// (*refused = 1: the list it was packed with differs from this slot's).  The same kernels and checks as the RCCL forms --
// this is how the format crosses a process boundary on a box with one GPU (shard.py's staged transport over gloo:
// tests/test_gpu_shard.py, bench.py's rehearsal) and how a host framework with its own transport would use it.
int sdm_compact_pack_host(sdm_ctx* c, int slot, float* out)
{
    int rc = check_slot(c, slot, true);
    if (rc) return rc;
    if (!out) return fail(SDM_EINVAL, "null buffer");
    if (c->xchg_entries <= 0) return fail(SDM_ESTATE, "sdm_exchange_compact(ctx, entries_per_map > 0) first");
    if (!c->has_depth[slot]) return fail(SDM_ESTATE, "slot has no reconstructed depth map");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if ((rc = check_compact_sources(c, 1, &slot))) return rc;
    if ((rc = ensure_xchg_buffers(c, 1, 0))) return rc;
    float2* buf = c->stage_buf;
    if ((rc = list_kernel(c, true, 1, &slot, &buf, c->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out, buf, sizeof(float2) * (size_t)xchg_stride(c), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SDM_OK;
}

int sdm_compact_unpack_host(sdm_ctx* c, int slot, const float* in, int* refused)
{
    int rc = check_slot(c, slot, true);
    if (rc) return rc;
    if (!in || !refused) return fail(SDM_EINVAL, "null argument");
    if (c->xchg_entries <= 0) return fail(SDM_ESTATE, "sdm_exchange_compact(ctx, entries_per_map > 0) first");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if ((rc = ensure_xchg_buffers(c, 0, 1))) return rc;
    {
        // the payload's header is in host memory: a payload packed with another list is refused BEFORE the destination plane
        // is touched (prepare_compact_destinations zeroes a plane that is not a pipeline map yet)
        if (!(c->act_lambdaG[slot] == c->dprm.lambdaG) && (rc = build_active(c, slot))) return rc;
        if ((rc = sync_counts(c))) return rc;
        unsigned long long h = 0;
        HIP_TRY(hipMemcpyAsync(&h, c->d_act_hash + slot, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const float* hdr = in + 2 * (size_t)c->xchg_entries;
        int hn;
        unsigned hlo, hhi;
        memcpy(&hn, hdr, 4);
        memcpy(&hlo, hdr + 1, 4);
        memcpy(&hhi, hdr + 2, 4);
        if (hn != c->h_act_count[slot] || hlo != (unsigned)h || hhi != (unsigned)(h >> 32)) {
            const unsigned one = 1u;  // counted like a refusal on the RCCL path (sdm_exchange_mismatches)
            unsigned cur = 0;
            HIP_TRY(hipMemcpy(&cur, c->d_xchg_mismatch, sizeof(unsigned), hipMemcpyDeviceToHost));
            cur += one;
            HIP_TRY(hipMemcpy(c->d_xchg_mismatch, &cur, sizeof(unsigned), hipMemcpyHostToDevice));
            *refused = 1;
            return SDM_OK;
        }
    }
    if ((rc = prepare_compact_destinations(c, 1, &slot, c->stream))) return rc;
    unsigned before = 0, after = 0;
    HIP_TRY(hipMemcpyAsync(&before, c->d_xchg_mismatch, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    float2* buf = c->gather_buf;
    HIP_TRY(hipMemcpyAsync(buf, in, sizeof(float2) * (size_t)xchg_stride(c), hipMemcpyHostToDevice, c->stream));
    if ((rc = list_kernel(c, false, 1, &slot, &buf, c->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(&after, c->d_xchg_mismatch, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *refused = after != before ? 1 : 0;
    if (!*refused) mark_received(c, slot);
    return SDM_OK;
}

// Would these slots' maps be accepted as compact sources right now (pipeline maps under the current lambdaG: zero outside
// their active lists)?  The query form of the check the compact sends make before anything is posted: a driver folds it
// into its per-pass wire-format agreement, so that a rank whose send sources do not qualify (a map restored with
// sdm_upload_depth, say) makes ALL ranks use whole maps for the pass instead of failing alone with its peers' receives
// already posted.  List lengths are not part of the answer (they are what sdm_comm_all_max agrees on).
int sdm_compact_sources_ready(sdm_ctx* c, int n, const int* slots, int* ready)
{
    if (!c || !ready || (n > 0 && !slots)) return fail(SDM_EINVAL, "null argument");
    *ready = 1;
    for (int i = 0; i < n; i++) {
        if (slots[i] < 0 || slots[i] >= c->cfg.max_keyframes) return fail(SDM_EINVAL, "slot out of range");
        if (!(c->recon_lambdaG[slots[i]] == c->dprm.lambdaG) || !(c->act_lambdaG[slots[i]] == c->dprm.lambdaG)) *ready = 0;
    }
    return SDM_OK;
}

// Compact maps whose sender's list (length or hash) differed from the receiver's (they were NOT scattered; the destination plane
// keeps what it held): the count since the last call, after waiting for everything queued.  0 on a healthy job -- the lists
// are a function of the keyframe's image and lambdaG, which sender and receiver share.
int sdm_exchange_mismatches(sdm_ctx* c, int* count)
{
    if (!c || !count) return fail(SDM_EINVAL, "null argument");
    *count = 0;
    if (!c->d_xchg_mismatch) return SDM_OK;
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->comm_stream) HIP_TRY(hipStreamSynchronize(c->comm_stream));
    unsigned v = 0;
    HIP_TRY(hipMemcpy(&v, c->d_xchg_mismatch, sizeof(unsigned), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(c->d_xchg_mismatch, 0, sizeof(unsigned)));
    *count = (int)v;
    return SDM_OK;
}

// Go / no-go across the ranks before a collective pass: every rank contributes its local verdict, all receive the
// minimum.  A rank that cannot take part in the exchange it planned (a keyframe without an image, a failed upload)
// says so HERE, so that its peers skip the pass instead of waiting for transfers that never come.  Host-blocking.
int sdm_comm_all_ok(sdm_ctx* c, int local_ok, int* all_ok)
{
    if (!c || !all_ok) return fail(SDM_EINVAL, "null argument");
    *all_ok = local_ok ? 1 : 0;
    if (!c->comm) return SDM_OK;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (!c->d_agree) HIP_TRY(hipMalloc((void**)&c->d_agree, sizeof(int)));
    const int v = local_ok ? 1 : 0;
    HIP_TRY(hipMemcpyAsync(c->d_agree, &v, sizeof(int), hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(g_rccl.AllReduce(c->d_agree, c->d_agree, 1, ncclInt, ncclMin, (ncclComm_t)c->comm, c->stream));
    int out = 0;
    HIP_TRY(hipMemcpyAsync(&out, c->d_agree, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *all_ok = out;
    return SDM_OK;
}

int sdm_comm_all_max(sdm_ctx* c, int local_value, int* all_max)
{
    if (!c || !all_max) return fail(SDM_EINVAL, "null argument");
    *all_max = local_value;
    if (!c->comm) return SDM_OK;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (!c->d_agree) HIP_TRY(hipMalloc((void**)&c->d_agree, sizeof(int)));
    HIP_TRY(hipMemcpyAsync(c->d_agree, &local_value, sizeof(int), hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(g_rccl.AllReduce(c->d_agree, c->d_agree, 1, ncclInt, ncclMax, (ncclComm_t)c->comm, c->stream));
    int out = 0;
    HIP_TRY(hipMemcpyAsync(&out, c->d_agree, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *all_max = out;
    return SDM_OK;
}

// Maps written into the depth pool from outside the engine (an ext_depth_pool filled by the host framework's
// own collective): marks the slots as holding finished depth maps (kf->semidense_flag_, PM.cc:292-298).
int sdm_mark_depth_present(sdm_ctx* c, int n, const int* slots)
{
    if (!c || (n > 0 && !slots)) return fail(SDM_EINVAL, "null argument");
    for (int i = 0; i < n; i++) {
        if (slots[i] < 0 || slots[i] >= c->cfg.max_keyframes) return fail(SDM_EINVAL, "slot out of range");
        c->has_depth[slots[i]] = 1;
        c->recon_lambdaG[slots[i]] = std::nanf("");
    }
    return SDM_OK;
}

}  // extern "C"
