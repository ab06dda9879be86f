// sdm_ingest.h -- keyframe ingest for a BATCH of keyframes: the pre-processing PM.cc assumes on
// KeyFrame (GradImg / GradTheta / I_stddev), the search records, the active-pixel lists and the
// per-slot metadata, in three launches over (keyframe, tile) whatever the batch size:
//   k_prepass_batch   image tile (+ halo) in LDS -> Scharr/32 gradient, magnitude, fastAtan2 phase -> the 16-byte search
//                     records directly (no gradient planes in HBM), per-tile integer sums for I_stddev, the lambdaG gate
//                     of PM.cc:201 as one lane mask per 64-pixel row segment, a hash of the gated pixel set, and the
//                     zero-fill of the slot's depth / checked / point planes (a fresh KeyFrame's maps);
//   k_prepass_finish  one workgroup per keyframe: I_stddev (PM.cc:457) from the tile sums, exclusive scan of the row
//                     segments' popcounts (raster order), list length, list hash, the slot's KfMeta;
//   k_list_write      one wave per row segment: (y << 16 | x) of the gated pixels at their raster-order position.
// Round 3 issued twelve launches and four memsets per keyframe (k_gradient, k_istd_finish, k_pack, k_active_count /
// _scan / _write, k_set_meta, ...), each launch-latency sized; results here are the same bit for bit
// (tests/test_gpu_ingest.py: batch vs one keyframe per call vs the oracle).
// The frame ingest before it (cv::undistort + cvtColor as the fork applies them, src/Tracking.cc:244-271,
// src/Modeler/Modeler.cc:154-155) is k_ingest_batch, one launch per batch as well.
#pragma once
#include "sdm_device.h"

namespace sdm {

constexpr int TILE_W = 64;
constexpr int TILE_H = 16;
constexpr int TILE_PX = TILE_W * TILE_H;  // 1024
constexpr int BLOCK = 256;
constexpr int PX_PER_THREAD = TILE_PX / BLOCK;  // 4

struct IngestItem {      // one keyframe of an ingest batch (device table, staged with one copy per batch)
    const uint8_t* img;  // gray image in device memory, H*W
    const uint8_t* src;  // colour / distorted frame in device memory (k_ingest_batch writes `img` from it), or null
    int slot;
    int keep_istd;       // k_prepass_finish leaves the slot's I_stddev alone
    KfMeta meta;         // stored into the slot's metadata (I_stddev is filled in on the device)
};
// the item tables of the chunks one k_prepass_batch launch covers: keyframe kf of the launch = p[kf / per][kf % per]
// (one chunk: p[0] and per >= its size; several chunks in one launch: all but the last hold `per` keyframes)
constexpr int CHUNK_TABLES = 4;
struct ChunkTables {
    const IngestItem* p[CHUNK_TABLES];
    int per;
};
__device__ __forceinline__ const IngestItem& chunk_item(const ChunkTables& t, int kf)
{
    const int ch = kf / t.per;
    return t.p[ch][kf - ch * t.per];
}

// ---- K(-1): image ingest -- what the fork does to a camera frame before the path sees it --------------------
// src/Tracking.cc:266-271 undistorts the (colour) frame with cv::undistort(im, imu, mK, mDistCoef) and hands it to
// Modeler::AddFrameImage, which keeps it 3-channel (src/Modeler/Modeler.cc:1496-1514); the Modeler converts it to
// gray with cvtColor(CV_RGB2GRAY) where it uses it (src/Modeler/Modeler.cc:154-155); Tracking's own gray image
// (for ORB) comes from cvtColor(RGB/BGR/RGBA/BGRA -> GRAY), src/Tracking.cc:244-257.
// One thread per OUTPUT pixel, as cv::undistort does it: the distorted source position in double
// (initUndistortRectifyMap with R = I and the same camera matrix), rounded to the 1/32-pixel fixed-point map
// (CV_16SC2 + CV_16UC1, INTER_BITS = 5), bilinear remap in 15-bit fixed point with a constant zero border, then
// the 8-bit RGB->gray fixed-point weights 4899/9617/1868 >> 14.  OpenCV is absent from the image: this is the
// published algorithm restated from memory -- PARITY UNPINNED; the oracle states the same
// arithmetic and the two agree bit for bit.
struct IngestParams {
    double fx, fy, cx, cy;
    double k1, k2, p1, p2, k3;
    int undistort;  // 0: dist == NULL (copy / grey-convert only)
    int channels;   // 1, 3 or 4 interleaved bytes per pixel
    int r_idx, g_idx, b_idx;  // byte index of R, G, B inside a pixel
};

__device__ __forceinline__ int ingest_gray(int r, int g, int b)
{
    return (r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14;  // RGB2Gray<uchar>: R2Y, G2Y, B2Y, yuv_shift = 14
}

// blockIdx.y = keyframe of the batch (all frames of a batch share the camera: IngestParams)
__global__ __launch_bounds__(BLOCK) void k_ingest_batch(const IngestItem* __restrict__ items, int W, int H, IngestParams q)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= W * H) return;
    const uint8_t* __restrict__ src = items[blockIdx.y].src;
    uint8_t* __restrict__ gray = const_cast<uint8_t*>(items[blockIdx.y].img);
    const int v = i / W, u = i - v * W;
    const int ch = q.channels;
    if (!q.undistort) {
        const uint8_t* px = src + (size_t)i * ch;
        gray[i] = (uint8_t)(ch == 1 ? px[0] : ingest_gray(px[q.r_idx], px[q.g_idx], px[q.b_idx]));
        return;
    }
    const double x = ((double)u - q.cx) / q.fx, y = ((double)v - q.cy) / q.fy;
    const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
    const double kr = 1 + ((q.k3 * r2 + q.k2) * r2 + q.k1) * r2;
    const double xd = x * kr + q.p1 * _2xy + q.p2 * (r2 + 2 * x2);
    const double yd = y * kr + q.p1 * (r2 + 2 * y2) + q.p2 * _2xy;
    const double us = q.fx * xd + q.cx, vs = q.fy * yd + q.cy;
    // saturate_cast<int>(double) = round half to even, clamped; a NaN position lands outside the image
    double fu = rint(us * 32.0), fv = rint(vs * 32.0);
    if (!(fu > -2147483648.0)) fu = -2147483648.0;
    if (!(fv > -2147483648.0)) fv = -2147483648.0;
    if (fu > 2147483647.0) fu = 2147483647.0;
    if (fv > 2147483647.0) fv = 2147483647.0;
    const int iu = (int)fu, iv = (int)fv;
    const int sx = iu >> 5, sy = iv >> 5, a = iu & 31, b = iv & 31;
    const int w00 = (32 - a) * (32 - b) * 32, w01 = a * (32 - b) * 32, w10 = (32 - a) * b * 32, w11 = a * b * 32;
    int acc[3] = {0, 0, 0};
    const int idx[3] = {ch == 1 ? 0 : q.r_idx, ch == 1 ? 0 : q.g_idx, ch == 1 ? 0 : q.b_idx};
    const int nc = ch == 1 ? 1 : 3;
    for (int t = 0; t < 4; t++) {
        const int yy = sy + (t >> 1), xx = sx + (t & 1);
        const int w = t == 0 ? w00 : (t == 1 ? w01 : (t == 2 ? w10 : w11));
        if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;  // BORDER_CONSTANT, value 0
        const uint8_t* px = src + ((size_t)yy * W + xx) * ch;
        for (int c = 0; c < nc; c++) acc[c] += w * (int)px[idx[c]];
    }
    int val[3];
    for (int c = 0; c < nc; c++) val[c] = (acc[c] + (1 << 14)) >> 15;  // FixedPtCast<int, uchar, INTER_REMAP_COEF_BITS>
    gray[i] = (uint8_t)(ch == 1 ? val[0] : ingest_gray(val[0], val[1], val[2]));
}

// ---- active-pixel lists -----------------------------------------------------------------------------------------
// The reference skips every pixel with GradImg < lambdaG (PM.cc:201), ~80 % of an image, and every later stage only
// ever touches the survivors.  That set depends on the keyframe's own image only, so it is built ONCE when the
// keyframe is uploaded: act[] holds (y << 16 | x) of the inset pixels that pass the gate, in raster order.
// A 64x16 tile is 16 row segments of 64 pixels = one wave-wide ballot each; segment (y, tx) of an image has raster
// rank y * tiles_x + tx, and the list is the concatenation of the segments' set bits in that order.
__device__ __forceinline__ bool act_gate(float grad, int x, int y, int W, int H, float lambdaG)
{
    return (x >= 2 && x < W - 2 && y >= 2 && y < H - 2) &&  // PM.cc:198-199
           !(grad < lambdaG);                                // PM.cc:201
}
// Hash of the gated pixel SET (the list is that set in raster order, so equal sets <=> equal lists <=> equal gate masks of all
// 64-pixel row segments): the sum over the row segments with a non-empty mask of a 64-bit mix of the mask and the segment's
// first pixel (y << 16 | x0).  One term per wave and row, from a ballot -- wave-uniform, so it is scalar-unit work (a term per
// PIXEL, the rounds 2-4 definition, was a fifth of the pre-pass's vector instructions: two 64-bit multiplies per lane).
// Sender and receiver of a compact map compare length and hash (sdm_comm.h); shard.list_hash is the numpy statement.
__device__ __forceinline__ unsigned long long seg_hash_term(unsigned long long mask, unsigned y, unsigned x0)
{
    if (mask == 0ull) return 0ull;
    unsigned long long z = mask ^ ((unsigned long long)((y << 16) | x0) * 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;  // SplitMix64 finaliser
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

constexpr int PART_WORDS = 4;  // per-tile partials: sum of im, sum of im^2, list hash, "a GradTheta value outside [0,360]"
constexpr int PRE_HALO_W = TILE_W + 2;
constexpr int PRE_HALO_H = TILE_H + 3;  // the record of row y also holds GradImg(y+1): one more gradient row per tile

// a tile's partials -> part_tile[PART_WORDS]: s1, s2 = the thread's sums of im and im^2 (32 bits hold a tile's: 1024 * 255^2),
// hsum / bad = the WAVE's hash terms and flag (uniform)
__device__ __forceinline__ void tile_partials(unsigned s1, unsigned s2, unsigned long long hsum, bool bad,
                                              unsigned long long* red /* [PART_WORDS][BLOCK/64] */,
                                              unsigned long long* __restrict__ part_tile)
{
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_down(s1, o);
        s2 += __shfl_down(s2, o);
    }
    const int wv = (int)(threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0) {
        red[0 * (BLOCK / 64) + wv] = s1;
        red[1 * (BLOCK / 64) + wv] = s2;
        red[2 * (BLOCK / 64) + wv] = hsum;
        red[3 * (BLOCK / 64) + wv] = bad ? 1ull : 0ull;
    }
    __syncthreads();
    if (threadIdx.x < PART_WORDS) {
        unsigned long long tot = 0;
        for (int w = 0; w < BLOCK / 64; w++) tot += red[threadIdx.x * (BLOCK / 64) + w];
        part_tile[threadIdx.x] = tot;
    }
}

// sqrtf(x) for x = (sx^2 + sy^2) / 1024 with integers |sx|, |sy| <= 16 * 255: x is 0 or lies in [2^-10, 2^15), where
// x * rsq(x) plus one residual step is the IEEE square root bit for bit (sdm_device.h sqrt_exact: exhaustive over
// [2^-102, 2^128), profiles/r04_exact_ops.txt); 0 is selected, not branched on
__device__ __forceinline__ float sqrt_grad(float x)
{
#if SDM_K1_OPT & 0x10000
    const float r = __builtin_amdgcn_rsqf(x);
    const float y0 = x * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-y0, y0, x);
    const float y = __builtin_fmaf(e, h, y0);
    return x == 0.0f ? 0.0f : y;
#else
    return sqrtf(x);
#endif
}

// Scharr/32 gradient at tile pixel (lx, ly) from the staged image tile (3x3 neighbourhood rows ly..ly+2, columns
// lx..lx+2 of t): the arithmetic of the per-keyframe pre-pass, float for float
__device__ __forceinline__ void scharr32(const uint8_t (*t)[PRE_HALO_W], int lx, int ly, float& gx, float& gy, int& centre)
{
    const int a00 = t[ly][lx], a01 = t[ly][lx + 1], a02 = t[ly][lx + 2];
    const int a10 = t[ly + 1][lx], a11 = t[ly + 1][lx + 1], a12 = t[ly + 1][lx + 2];
    const int a20 = t[ly + 2][lx], a21 = t[ly + 2][lx + 1], a22 = t[ly + 2][lx + 2];
    const int sx = 3 * (a02 - a00) + 10 * (a12 - a10) + 3 * (a22 - a20);
    const int sy = 3 * (a20 - a00) + 10 * (a21 - a01) + 3 * (a22 - a02);
    gx = (float)sx * (1.0f / 32.0f);
    gy = (float)sy * (1.0f / 32.0f);
    centre = a11;
}

// One 64-pixel row segment's two words of the slot's gate bit planes (sdm_device.h scan_masked; every lane of the wave
// calls): sgate = the scan's gradient gate on this pixel as a CANDIDATE of another keyframe's search (PM.cc:411), theta its
// GradTheta.  Plane b < MASK_BINS holds the gated pixels whose angle falls into bin b -- and those whose angle is not in
// [0,360) at all --, planes MASK_BINS .. MASK_UNION-1 repeat the first bins, plane MASK_UNION holds all gated pixels.  Lane p
// stores plane p's two dwords (MASK_PLANES consecutive dwords per 32-column word).
__device__ __forceinline__ void write_gate_planes(unsigned* __restrict__ gmask, long long row_dword0, int tx, bool sgate, float theta,
                                                  bool store)
{
    // six ballots -- the gate, "gated with an angle in [0,360)", the four bits of the bin -- instead of one per plane: lane p
    // assembles plane p's word from them with its own constants (k-th bit of its bin set ? B_k : ~B_k)
    const bool in_range = (theta >= 0.0f) & (theta < 360.0f);
    const int bin = in_range ? mask_bin(theta) : 0;
    const int lane = (int)(threadIdx.x & 63u);
    const int mybin = lane < MASK_BINS ? lane : lane - MASK_BINS;
    static_assert(MASK_BINS == 16, "four bin bits");
    const unsigned long long G = __builtin_amdgcn_ballot_w64(sgate);
    const unsigned long long R = __builtin_amdgcn_ballot_w64(sgate & in_range);
    unsigned long long mine = R;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned long long Bk = __builtin_amdgcn_ballot_w64((bin >> k) & 1);
        const unsigned long long flip = ((mybin >> k) & 1) ? 0ull : ~0ull;
        mine &= Bk ^ flip;
    }
    mine |= G & ~R;  // an angle outside [0,360) belongs to every bin
    if (lane == MASK_UNION) mine = G;
    if (store && lane < MASK_PLANES) {
        gmask[row_dword0 + (long long)(2 * tx) * MASK_PLANES + lane] = (unsigned)mine;
        gmask[row_dword0 + (long long)(2 * tx + 1) * MASK_PLANES + lane] = (unsigned)(mine >> 32);
    }
}

// grid (tiles, keyframes).  ZERO: the slot receives a new keyframe -- its depth map, checked plane and point set start
// as zeros (a fresh KeyFrame's depth_map_ / depth_sigma_ / SemiDensePointSets_)
template <bool ZERO>
__global__ __launch_bounds__(BLOCK) void k_prepass_batch(const ChunkTables tabs, int W, int H, int tiles_x,
                                                         long long plane, float4* __restrict__ rec,
                                                         float2* __restrict__ pool, float* __restrict__ chk,
                                                         float* __restrict__ xyz, float lambdaG,
                                                         unsigned long long* __restrict__ part,
                                                         unsigned long long* __restrict__ seg_mask, int nseg,
                                                         unsigned* __restrict__ gmask, int mrow, int kf0,
                                                         IngestItem* __restrict__ gitems)
{
    __shared__ uint8_t t[PRE_HALO_H][PRE_HALO_W];
    __shared__ float gm[TILE_H + 1][TILE_W];
    __shared__ unsigned long long red[PART_WORDS * (BLOCK / 64)];
    const int kf = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const int gkf = kf0 + kf;  // the keyframe's index in its GROUP of chunks (k_prepass_finish / k_list_write run once per group)
    const IngestItem& item = chunk_item(tabs, kf);
    const uint8_t* __restrict__ im = item.img;
    const int slot = item.slot;
    if (tile == 0 && tid == 0) gitems[gkf] = item;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int tx0 = tx * TILE_W, ty0 = ty * TILE_H;
    const int lx = tid & (TILE_W - 1), wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: rows become scalar work)
    for (int hy = wv; hy < PRE_HALO_H; hy += BLOCK / 64) {  // replicated border; a wave per halo row
        const uint8_t* __restrict__ row = im + (long long)min(max(ty0 + hy - 1, 0), H - 1) * W;
        t[hy][lx] = row[min(max(tx0 + lx - 1, 0), W - 1)];
        if (lx < PRE_HALO_W - TILE_W) t[hy][TILE_W + lx] = row[min(max(tx0 + TILE_W + lx - 1, 0), W - 1)];
    }
    __syncthreads();
    float th[PX_PER_THREAD];
    unsigned s1 = 0u, s2 = 0u;
    unsigned long long hsum = 0ull;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < PX_PER_THREAD; i++) {
        const int ly = i * (BLOCK / 64) + wv;
        float gx, gy;
        int centre;
        scharr32(t, lx, ly, gx, gy, centre);
        const float xx = gx * gx, yy = gy * gy;
        gm[ly][lx] = sqrt_grad(xx + yy);
        th[i] = fast_atan2_deg(gy, gx);
        if (tx0 + lx < W && ty0 + ly < H) {  // exact integer sums for I_stddev (any order gives the same total)
            s1 += (unsigned)centre;
            s2 += (unsigned)(centre * centre);
        }
    }
    if (wv == 0) {  // gradient row TILE_H of the tile = row 0 of the tile below: GradImg(y+1) of the last record row
        float gx, gy;
        int centre;
        scharr32(t, lx, TILE_H, gx, gy, centre);
        const float xx = gx * gx, yy = gy * gy;
        gm[TILE_H][lx] = sqrt_grad(xx + yy);
    }
    __syncthreads();
    const long long base = (long long)slot * plane;
#pragma unroll
    for (int i = 0; i < PX_PER_THREAD; i++) {
        const int ly = i * (BLOCK / 64) + wv;
        const int x = tx0 + lx, y = ty0 + ly;
        const bool in = x < W && y < H;
        bool gate = false, sgate = false, odd = false;
        if (in) {
            const bool below = y + 1 < H;
            const unsigned bits = (unsigned)t[ly + 1][lx + 1] | ((below ? (unsigned)t[ly + 2][lx + 1] : 0u) << 8);
            float4 r;
            r.x = gm[ly][lx];
            r.y = th[i];
            r.z = below ? gm[ly + 1][lx] : 0.0f;
            r.w = __uint_as_float(bits);
            const long long o = base + (long long)y * W + x;
            rec[o] = r;
            if (ZERO) {
                pool[o] = make_float2(0.f, 0.f);
                chk[o] = 0.f;
            }
            gate = act_gate(r.x, x, y, W, H, lambdaG);
            sgate = !(r.x < lambdaG);  // the scan's gate on this pixel as a CANDIDATE of another keyframe's search, PM.cc:411
            odd = !(r.y >= 0.0f && r.y <= 360.0f);  // never from fastAtan2; kept for symmetry with k_pack
        }
        bad = bad || __builtin_amdgcn_ballot_w64(odd) != 0ull;
        if (ZERO && xyz && y < H) {  // the segment's 3 * 64 floats as dense dword stores
            float* __restrict__ z = xyz + (base + (long long)y * W + tx0) * 3;
            const int nz = 3 * min(TILE_W, W - tx0);
#pragma unroll
            for (int k = 0; k < 3; k++)
                if (lx + TILE_W * k < nz) z[lx + TILE_W * k] = 0.f;
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(gate);
        hsum += seg_hash_term(m, (unsigned)y, (unsigned)tx0);
        if (lx == 0 && y < H) seg_mask[(long long)gkf * nseg + (long long)y * tiles_x + tx] = m;
        write_gate_planes(gmask, ((long long)slot * H + y) * MASK_PLANES * mrow, tx, sgate, th[i], y < H);
    }
    tile_partials(s1, s2, hsum, bad, red, part + ((long long)gkf * gridDim.x + tile) * PART_WORDS);
}

// the same gate masks and hash from a slot's RECORDS (lists rebuilt under another lambdaG, or records packed from the
// caller's own planes: sdm_upload_keyframe); partials 0, 1, 3 are not produced
__global__ __launch_bounds__(BLOCK) void k_gate_batch(const IngestItem* __restrict__ items, int W, int H, int tiles_x,
                                                      long long plane, const float4* __restrict__ rec, float lambdaG,
                                                      unsigned long long* __restrict__ part,
                                                      unsigned long long* __restrict__ seg_mask, int nseg,
                                                      unsigned* __restrict__ gmask, int mrow, int kf0,
                                                      IngestItem* __restrict__ gitems)
{
    __shared__ unsigned long long red[PART_WORDS * (BLOCK / 64)];
    const int kf = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const int gkf = kf0 + kf;
    const int slot = items[kf].slot;
    if (tile == 0 && tid == 0) gitems[gkf] = items[kf];
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int lx = tid & (TILE_W - 1), wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float4* __restrict__ r = rec + (long long)slot * plane;
    unsigned long long hsum = 0ull;
#pragma unroll
    for (int i = 0; i < PX_PER_THREAD; i++) {
        const int x = tx * TILE_W + lx, y = ty * TILE_H + i * (BLOCK / 64) + wv;
        bool gate = false, sgate = false;
        float theta = 0.0f;
        if (x < W && y < H) {
            const float4 rr = r[(long long)y * W + x];
            gate = act_gate(rr.x, x, y, W, H, lambdaG);
            sgate = !(rr.x < lambdaG);
            theta = rr.y;
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(gate);
        hsum += seg_hash_term(m, (unsigned)y, (unsigned)(tx * TILE_W));
        if (lx == 0 && y < H) seg_mask[(long long)gkf * nseg + (long long)y * tiles_x + tx] = m;
        write_gate_planes(gmask, ((long long)slot * H + y) * MASK_PLANES * mrow, tx, sgate, theta, y < H);
    }
    tile_partials(0u, 0u, hsum, false, red, part + ((long long)gkf * gridDim.x + tile) * PART_WORDS);
}

// One workgroup per keyframe of a group of chunks (items = the group's table, filled by the chunks' first kernels).  The list
// length also goes straight into the host's pinned mirror (host_count: device-visible host memory; the host reads it behind an
// event recorded after this kernel).  full != 0 (a keyframe was uploaded): I_stddev = population sigma of im (PM.cc:457) from
// the tile sums, the slot's metadata and its theta flag are written; otherwise only the list (length, hash, offsets).
constexpr int FIN_BLOCK = 1024;
__global__ __launch_bounds__(FIN_BLOCK) void k_prepass_finish(const IngestItem* __restrict__ items, int W, int H, int ntiles,
                                                              int nseg, const unsigned long long* __restrict__ part,
                                                              const unsigned long long* __restrict__ seg_mask,
                                                              int* __restrict__ seg_off, KfMeta* __restrict__ meta,
                                                              int* __restrict__ act_count, int* __restrict__ host_count,
                                                              int* __restrict__ theta_bad,
                                                              unsigned long long* __restrict__ act_hash, int full)
{
    __shared__ unsigned long long red[PART_WORDS][FIN_BLOCK / 64];
    __shared__ int wtot[FIN_BLOCK / 64];
    const int kf = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const IngestItem it = items[kf];
    unsigned long long v[PART_WORDS] = {0ull, 0ull, 0ull, 0ull};
    for (int i = tid; i < ntiles; i += FIN_BLOCK)
#pragma unroll
        for (int k = 0; k < PART_WORDS; k++) v[k] += part[((long long)kf * ntiles + i) * PART_WORDS + k];
#pragma unroll
    for (int k = 0; k < PART_WORDS; k++) {
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o);
        if (lane == 0) red[k][wv] = v[k];
    }
    // exclusive scan of the segments' popcounts in raster order: thread t owns segments [t*per, (t+1)*per)
    const unsigned long long* __restrict__ sm = seg_mask + (long long)kf * nseg;
    int* __restrict__ so = seg_off + (long long)kf * nseg;
    const int per = (nseg + FIN_BLOCK - 1) / FIN_BLOCK;
    const int s0 = tid * per, s1 = min(s0 + per, nseg);
    int mine = 0;
    for (int s = s0; s < s1; s++) mine += __popcll(sm[s]);
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
    for (int w = 0; w < FIN_BLOCK / 64; w++) {
        if (w == wv) wbase = total;
        total += wtot[w];
    }
    int run = wbase + incl - mine;
    for (int s = s0; s < s1; s++) {
        so[s] = run;
        run += __popcll(sm[s]);
    }
    if (tid == 0) {
        unsigned long long tot[PART_WORDS];
        for (int k = 0; k < PART_WORDS; k++) {
            tot[k] = 0;
            for (int w = 0; w < FIN_BLOCK / 64; w++) tot[k] += red[k][w];
        }
        act_count[it.slot] = total;
        host_count[it.slot] = total;
        act_hash[it.slot] = tot[2];
        if (full) {
            KfMeta m = it.meta;
            if (it.keep_istd) {
                m.I_stddev = meta[it.slot].I_stddev;
            } else {
                const double n = (double)W * (double)H;
                const double mean = (double)tot[0] / n;
                double var = (double)tot[1] / n - mean * mean;
                if (var < 0) var = 0;
                m.I_stddev = (float)sqrt(var);
            }
            meta[it.slot] = m;
            theta_bad[it.slot] = tot[3] ? 1 : 0;
        }
    }
}

// one wave per LIST_SEGS consecutive row segments: the set bits of their masks, at the segments' raster-order offsets
constexpr int LIST_SEGS = 16;
__global__ __launch_bounds__(BLOCK) void k_list_write(const IngestItem* __restrict__ items, int tiles_x, int nseg,
                                                      long long plane, const unsigned long long* __restrict__ seg_mask,
                                                      const int* __restrict__ seg_off, unsigned* __restrict__ act)
{
    const int kf = blockIdx.y;
    const int s0 = (blockIdx.x * (BLOCK / 64) + (int)(threadIdx.x >> 6)) * LIST_SEGS, lane = threadIdx.x & 63;
    if (s0 >= nseg) return;
    const int mine = min(s0 + (lane & (LIST_SEGS - 1)), nseg - 1);
    const unsigned long long mm = seg_mask[(long long)kf * nseg + mine];
    const int mo = seg_off[(long long)kf * nseg + mine];
    unsigned* __restrict__ out = act + (long long)items[kf].slot * plane;
    int y = s0 / tiles_x, tx = s0 - y * tiles_x;
#pragma unroll
    for (int j = 0; j < LIST_SEGS; j++) {
        if (s0 + j < nseg) {  // wave-uniform
            const unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mm >> 32), j) << 32) |
                                         (unsigned)__builtin_amdgcn_readlane((int)mm, j);
            const int off = __builtin_amdgcn_readlane(mo, j);
            if ((m >> lane) & 1ull)
                out[off + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned)y << 16) | (unsigned)(tx * TILE_W + lane);
        }
        if (++tx == tiles_x) {
            tx = 0;
            y++;
        }
    }
}

// pack im/grad/theta planes into the 16-byte search records (layout: sdm_device.h): the caller's own planes
// (sdm_upload_keyframe).  theta_bad: set when any GradTheta value lies outside [0,360] (or is NaN): the slot's pairs then
// keep the per-candidate precondition of the closed-form angle gates (PairConst::clean).
__global__ __launch_bounds__(BLOCK) void k_pack(const uint8_t* __restrict__ im, const float* __restrict__ grad,
                                                const float* __restrict__ theta, int W, int H,
                                                float4* __restrict__ rec, int* __restrict__ theta_bad)
{
    int idx = blockIdx.x * BLOCK + threadIdx.x;
    bool bad = false;
    if (idx < W * H) {
        int y = idx / W;
        bool below = (y + 1 < H);
        unsigned bits = (unsigned)im[idx] | ((below ? (unsigned)im[idx + W] : 0u) << 8);
        float4 r;
        r.x = grad[idx];
        r.y = theta[idx];
        r.z = below ? grad[idx + W] : 0.0f;
        r.w = __uint_as_float(bits);
        rec[idx] = r;
        bad = !(r.y >= 0.0f && r.y <= 360.0f);
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(theta_bad, 1);
}

__global__ __launch_bounds__(BLOCK) void k_unpack(const float4* __restrict__ rec, int n, uint8_t* __restrict__ im,
                                                  float* __restrict__ grad, float* __restrict__ theta)
{
    int idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    float4 r = rec[idx];
    im[idx] = (uint8_t)(__float_as_uint(r.w) & 0xffu);
    grad[idx] = r.x;
    theta[idx] = r.y;
}

// per-slot metadata written in stream order (80 bytes as a kernel argument: no copy, no host synchronisation)
__global__ void k_set_meta(KfMeta* __restrict__ dst, KfMeta m, int keep_istd)
{
    if (keep_istd) m.I_stddev = dst->I_stddev;
    *dst = m;
}

}  // namespace sdm
