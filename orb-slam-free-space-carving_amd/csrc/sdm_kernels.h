// sdm_kernels.h -- gfx950 kernels of the ProbabilityMapping path (included by sdm_engine.hip).
//
// Common shape of K1..K4: one 256-thread workgroup per 64x16 pixel tile of one reference keyframe.
// The workgroup compacts the tile's ACTIVE pixels (the ~20 % that pass the reference's per-pixel
// gate) into LDS in raster order, so waves are not spent on skipped pixels; thread t then owns
// active pixel t and loops the neighbour index j uniformly, which keeps the per-(ref,nbr)
// constants in scalar registers and the scan lengths of adjacent lanes similar.  Results are
// staged in an LDS copy of the tile and written back with coalesced 8-byte stores.
//
// Block -> (reference, tile) mapping is XCD-aware: blocks b and b+8 share an XCD (round-robin
// dispatch), so tile t of EVERY reference keyframe is given to XCD t%8; consecutive reference
// keyframes read the same neighbour-image region from that XCD's L2 instead of re-fetching it.
#pragma once
#include "sdm_device.h"
#include "sdm_ingest.h"

namespace sdm {

#ifndef SDM_INTRA_COMPACT
#define SDM_INTRA_COMPACT 1  // K2/K3 on pipeline maps through a compact result array (0: through a scratch plane, the round-2 form)
#endif

struct TileGeom {
    int W, H, tiles_x, tiles_y, ntiles;
};

__host__ __device__ inline TileGeom make_geom(int W, int H)
{
    TileGeom g;
    g.W = W;
    g.H = H;
    g.tiles_x = (W + TILE_W - 1) / TILE_W;
    g.tiles_y = (H + TILE_H - 1) / TILE_H;
    g.ntiles = g.tiles_x * g.tiles_y;
    return g;
}
__host__ inline int grid_blocks(const TileGeom& g, int n_ref) { return 8 * ((g.ntiles + 7) / 8) * n_ref; }

// XCD-aware decode; returns false for padding blocks.
__device__ __forceinline__ bool decode_block(const TileGeom& g, int n_ref, int& ref, int& tx0, int& ty0)
{
    int b = blockIdx.x;
    int xcd = b & 7, i = b >> 3;
    int tpx = (g.ntiles + 7) >> 3;
    ref = i / tpx;
    int tile = (i - ref * tpx) * 8 + xcd;
    if (ref >= n_ref || tile >= g.ntiles) return false;
    int ty = tile / g.tiles_x;
    tx0 = (tile - ty * g.tiles_x) * TILE_W;
    ty0 = ty * TILE_H;
    return true;
}

// Ordered compaction of up to 4 flags per thread (local index L = i*256 + tid) into act[].
// wsum: 16 ints of LDS.  Returns the number of active pixels.  Contains one __syncthreads().
__device__ __forceinline__ int block_compact(const bool (&f)[PX_PER_THREAD], unsigned short* act, int* wsum)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pre[PX_PER_THREAD];
#pragma unroll
    for (int i = 0; i < PX_PER_THREAD; i++) {
        unsigned long long m = __ballot(f[i]);
        pre[i] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[i * 4 + wave] = __popcll(m);
    }
    __syncthreads();
    int run = 0, total = 0;
    int off[PX_PER_THREAD];
#pragma unroll
    for (int i = 0; i < PX_PER_THREAD; i++) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w == wave) off[i] = run;
            run += wsum[i * 4 + w];
        }
    }
    total = run;
#pragma unroll
    for (int i = 0; i < PX_PER_THREAD; i++)
        if (f[i]) act[off[i] + pre[i]] = (unsigned short)(i * BLOCK + tid);
    return total;
}

#ifndef SDM_K23_GROUP
#define SDM_K23_GROUP 0  // the same order for the intra-keyframe list kernels (k_intra_compact / _commit / _list)
#endif
#ifndef SDM_K23_BAND
#define SDM_K23_BAND 4
#endif
#ifndef SDM_K4_BAND
#define SDM_K4_BAND 4  // chunks per band (0: one contiguous band per XCD)
#endif
#ifndef SDM_K4_GROUP
#define SDM_K4_GROUP 8  // n > 0: K4's blocks ordered by image band (per XCD) and groups of n reference keyframes; 0: chunk c of
                        // every keyframe on XCD c mod 8, reference fastest (rounds 1-3)
#endif
// ---- block -> (reference keyframe, list chunk) of the list kernels ----------------------------------------------------------
// Blocks b and b + 8 share an XCD (and its L2).  GROUP == 0: chunk c of EVERY keyframe on XCD c mod 8, reference fastest -- the
// keyframes that read one region of a neighbour image at the same time share an L2 fill (K1).  GROUP > 0: XCD x takes BANDS of
// `BAND` consecutive chunks (bands dealt round robin) -- vertically adjacent chunks read the same rows of a map -- and inside a
// band groups of GROUP consecutive keyframes, reference fastest, then chunk, then group (K2-K4, whose chunk spans an image
// row).  band_grid_per_ref() is the grid the host launches per keyframe.
template <int GROUP, int BAND>
__device__ __forceinline__ bool decode_list_block(int b, int n_ref, int max_chunks, int& ref, int& chunk)
{
    const int i8 = b >> 3;
    if (GROUP == 0) {
        const int cl = i8 / n_ref;
        ref = i8 - cl * n_ref;
        chunk = cl * 8 + (b & 7);
    } else {
        const int cpx = (max_chunks + 7) >> 3;  // chunks per XCD and keyframe
        const int band = (BAND > 0 && BAND < cpx) ? BAND : cpx;
        const int per_band = n_ref * band;
        const int kb = i8 / per_band;
        const int ib = i8 - kb * per_band;
        const int per_group = GROUP * band;
        const int g = ib / per_group;
        const int gsz = min(GROUP, n_ref - g * GROUP);  // the last group may be short
        const int tt = ib - g * per_group;
        const int cc = tt / gsz;
        ref = g * GROUP + (tt - cc * gsz);
        chunk = (kb * 8 + (b & 7)) * band + cc;
    }
    return chunk < max_chunks;
}
__host__ __device__ inline int band_grid_per_ref(int max_chunks, int group, int band)
{
    const int cpx = (max_chunks + 7) / 8;
    if (group == 0 || band <= 0 || band >= cpx) return 8 * cpx;
    return 8 * ((cpx + band - 1) / band) * band;
}

// ---- K4's approximate projection: per-pair error-bound constants (PairConst::pb) -------------------------------------
// K4 needs the projection (xj, yj) of PM.cc:677-680 only for the bounds test of PM.cc:695 and floor(): an approximation
// decides both unless it lies within its error bound of an integer (0 and cols-1 / rows-1 are integers).  With
// d = 2^-23, dp = RN(1/rho), r~ = v_rcp_f32(t2~) and the approximate chain
//     t_i~ = fma(n_i, dp, T_i)     u~ = nfx t0~ + ncx t2~ (the reference's operations)     xj~ = u~ r~
// a term-by-term comparison with the reference chain RN(RN(n_i / rho) + T_i), ..., RN(u / t2) gives
//     |t_i~ - t_i| <= d A_i,  A_i = 2 |n_i| dp + |T_i|        |u~ - u| <= 3 d B,  B = nfx A_0 + ncx A_2
//     |xj~ - xj|  <= d (3 B + |xj| A_2) |r~| (1 + d) + 2 d |xj|
// and with the per-pair bounds N_i >= |n_i| (over the whole reference image), C1 = max(nfx N_0 + ncx N_2, nfy N_1 + ncy N_2),
// C2 = max(nfx |tx| + ncx |tz|, nfy |ty| + ncy |tz|), mc = min(ncx, ncy):  B <= 2 C1 dp + C2,  A_2 <= (2 C1 dp + C2) / mc.  So
//     H = (pb[0] dp + pb[1]) |r~|,  pb[0] = 8 d C1, pb[1] = 4 d C2       (>= the absolute part, 25 % to spare)
//     G = H pb[2] + 3 d,            pb[2] = 1.25 / (4 mc)                (>= the part proportional to |xj|, 25 % to spare)
//     eps = H + G |xj~|
// bounds |xj~ - xj| (the second-order term G eps is inside the spare: G >= 0.16 makes H >= 0.5 because mc >= 1 is
// required, and then every value is within eps of an integer).  A lane whose xj~ or yj~ is within eps of an integer -- or
// NaN, or beyond 2^23 -- takes the exact projection (inter_project); Inf / NaN constants (a pair that fails the sanity
// test below) flag every lane.  sdm_selftest(7) compares cell and validity of unflagged lanes with the exact chain.
__device__ inline void k4_proj_bounds(PairConst& pc, float X0max, float X1max)
{
    const float d = 0x1p-23f, s = 1.0f + 0x1p-18f;
    const float N0 = (fabsf(pc.Rx[0]) * X0max + fabsf(pc.Rx[1]) * X1max + fabsf(pc.Rx[2])) * s;
    const float N1 = (fabsf(pc.Ry[0]) * X0max + fabsf(pc.Ry[1]) * X1max + fabsf(pc.Ry[2])) * s;
    const float N2 = (fabsf(pc.Rz[0]) * X0max + fabsf(pc.Rz[1]) * X1max + fabsf(pc.Rz[2])) * s;
    const float fx = fabsf(pc.nfx), fy = fabsf(pc.nfy), cx = fabsf(pc.ncx), cy = fabsf(pc.ncy);
    const float C1 = fmaxf(fx * N0 + cx * N2, fy * N1 + cy * N2) * s;
    const float C2 = fmaxf(fx * fabsf(pc.tx) + cx * fabsf(pc.tz), fy * fabsf(pc.ty) + cy * fabsf(pc.tz)) * s;
    const float mc = fminf(cx, cy);
    // every magnitude the chain can produce stays far inside the float range (rho itself is inside [2^-40, 2^41) or the
    // lane is slow anyway): no overflow, no denormal reciprocal
    const bool sane = (C1 < 0x1p60f) & (C2 < 0x1p60f) & (mc >= 1.0f) & (mc < 0x1p30f) & (fx < 0x1p30f) & (fy < 0x1p30f);
    pc.pb[0] = sane ? 8.0f * d * C1 * s : __builtin_inff();  // (a NaN operand fails `sane`)
    pc.pb[1] = sane ? 4.0f * d * C2 * s : __builtin_inff();
    pc.pb[2] = sane ? 0.3125f / mc * s : __builtin_inff();
}

// ---- per-batch constant tables ----------------------------------------------------------------------
// One thread per (reference, neighbour): the host work of PM.cc:170-195 (F12, R21, t21) done on
// device from the resident keyframe metadata, so a batch needs only slot indices from the host.
__global__ void k_pair_setup(const KfMeta* __restrict__ meta, const int* __restrict__ ref_slots,
                             const int* __restrict__ nbr_slots, const float* __restrict__ rot,
                             const float* __restrict__ mind, const float* __restrict__ maxd,
                             const int* __restrict__ act_counts, const int* __restrict__ theta_bad, int n_ref, int n,
                             int W, int H, RefConst* __restrict__ refs, PairConst* __restrict__ pairs)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_ref * n) return;
    int r = idx / n, j = idx - r * n;
    KfMeta m1 = meta[ref_slots[r]];
    KfMeta m2 = meta[nbr_slots[idx]];
    PairConst pc;
    float R21[9], t21[3];
    pair_geometry(m1, m2, pc.F, R21, t21);
    for (int i = 0; i < 3; i++) {
        pc.Rx[i] = R21[i];
        pc.Ry[i] = R21[3 + i];
        pc.Rz[i] = R21[6 + i];
    }
    pc.tx = t21[0];
    pc.ty = t21[1];
    pc.tz = t21[2];
    pc.rot = rot ? rot[idx] : 0.0f;
    pc.istd = m2.I_stddev;
    pc.nbr_slot = nbr_slots[idx];
    pc.nfx = m2.fx;
    pc.nfy = m2.fy;
    pc.ncx = m2.cx;
    pc.ncy = m2.cy;
    // th_pi, theta2 in [0,360] and rot in [-360,360] => th_pi + rot in [-360,720] => one wrap lands in [0,360] => d2, d3
    // in [-360,360]: the closed-form gates of the scan hold for every candidate of this pair (sdm_device.h)
    pc.clean = ((theta_bad[ref_slots[r]] == 0 && theta_bad[nbr_slots[idx]] == 0 && pc.rot >= -360.0f && pc.rot <= 360.0f) ? 1 : 0) |
               (line_quot_safe(pc.F) ? 2 : 0);
    {  // bit 2: a long search range at the principal point (xp = (0, 0, 1): the ray dot products are the rows' last entries)
        float umin, umax;
        search_range(m1.fx, m1.cx, pc.Rx[2], pc.Rz[2], pc.tx, pc.tz, mind ? mind[r] : 0.f, maxd ? maxd[r] : 0.f, W, umin, umax);
        if (!(umax - umin < (float)MASK_HINT_L)) pc.clean |= 4;  // (NaN ends: cannot tell)
    }
    // K4's approximate projection (k4_proj_bounds): the largest |xp0|, |xp1| any pixel of the reference image can have
    {
        const float s = 1.0f + 0x1p-20f;
        const float X0 = fmaxf(fabsf(m1.cx), fabsf((float)(W - 1) - m1.cx)) / fabsf(m1.fx) * s;
        const float X1 = fmaxf(fabsf(m1.cy), fabsf((float)(H - 1) - m1.cy)) / fabsf(m1.fy) * s;
        k4_proj_bounds(pc, X0, X1);
    }
    pairs[idx] = pc;
    if (j == 0) {
        RefConst rc;
        rc.slot = ref_slots[r];
        rc.fx = m1.fx;
        rc.fy = m1.fy;
        rc.cx = m1.cx;
        rc.cy = m1.cy;
        rc.mind = mind ? mind[r] : 0.f;
        rc.maxd = maxd ? maxd[r] : 0.f;
        rc.act_count = act_counts[ref_slots[r]];
        refs[r] = rc;
    }
}

// RefConst table for stages without neighbours (K2/K3/K5 on slots)
__global__ void k_ref_setup(const KfMeta* __restrict__ meta, const int* __restrict__ ref_slots,
                            const int* __restrict__ act_counts, int n_ref, RefConst* __restrict__ refs)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_ref) return;
    KfMeta m = meta[ref_slots[r]];
    RefConst rc;
    rc.slot = ref_slots[r];
    rc.fx = m.fx;
    rc.fy = m.fy;
    rc.cx = m.cx;
    rc.cy = m.cy;
    rc.mind = 0.f;
    rc.maxd = 0.f;
    rc.act_count = act_counts[ref_slots[r]];
    refs[r] = rc;
}

// zero the depth maps of a batch's reference keyframes (a fresh depth_map_/depth_sigma_)
__global__ __launch_bounds__(BLOCK) void k_zero_maps(float2* __restrict__ pool, long long plane,
                                                     const int* __restrict__ slots, int n_ref)
{
    long long idx = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= plane * n_ref) return;
    int r = (int)(idx / plane);
    long long p = idx - (long long)r * plane;
    pool[(long long)slots[r] * plane + p] = make_float2(0.f, 0.f);
}

// ---- K1: epipolar search + hypothesis fusion, PM.cc:197-231 --------------------------------------------
// Workgroup = 64 consecutive ACTIVE pixels of one reference keyframe x 4 waves.  Wave w searches
// the neighbours j = w, w+4, w+8, ... for all 64 pixels, so j -- and with it the per-(ref,nbr)
// constant block -- is wave-uniform (scalar registers) and adjacent lanes scan adjacent epipolar
// segments of the same neighbour image.  Splitting the neighbours over the waves keeps the LDS
// footprint at n*64 hypotheses (10 KB at n = 20), which lets 8 waves per SIMD stay resident to
// hide the gather latency; interleaving j (not blocking it) balances the waves, because the scan
// length grows with the baseline and neighbours are ordered by covisibility.
// LDS: float2 {rho, 1/sigma^2}[n][64] in neighbour order (rho = +Inf marks "no hypothesis", PM.cc:216), sigma[n][64],
// and the per-hypothesis compatible-set sizes (bytes): 13 B x n x 64 + 2 KB (18.3 KB at n = 20).
#ifndef SDM_K1_WAVES
#define SDM_K1_WAVES 4
#endif
constexpr int K1_PX = 64;                // active pixels per workgroup
constexpr int K1_WAVES = SDM_K1_WAVES;   // neighbour stripes
constexpr int K1_BLOCK = K1_PX * K1_WAVES;
constexpr int K1_OUTLIER_ROWS = 4;  // rows outside the first hypothesis' set that the fusion bound may still test
__host__ __device__ inline size_t k1_lds_bytes(int n)
{
    const size_t nn = (size_t)(n > 0 ? n : 1);
    // the per-row counters (bytes, 4 rows per word) double as the outlier rows' counters of the fusion bound: at least
    // K1_OUTLIER_ROWS words per pixel
    const size_t cw = (nn + 3) / 4 > (size_t)K1_OUTLIER_ROWS ? (nn + 3) / 4 : (size_t)K1_OUTLIER_ROWS;
    return (sizeof(float2) + sizeof(float)) * (size_t)K1_PX * nn + sizeof(unsigned) * (size_t)K1_PX * cw +
           sizeof(unsigned long long) * K1_BLOCK + sizeof(unsigned long long) * K1_PX + sizeof(unsigned) * K1_PX + 16;
}

// Pixels whose fusion neither bound settles are not finished by their workgroup: they are appended -- hypotheses and all --
// to a per-launch list, and k_fuse_open runs the all-pairs count for 64 OPEN pixels per workgroup.  A clean workgroup
// never pays the pair loop because one of its 64 pixels is open.
constexpr int STATS_WORDS = 16;  // sdm_stats counters (the self-tests borrow words 4 .. 7)
struct OpenList {
    unsigned* count;          // this launch's counters: [0] open pixels seen so far (decides the quota below), [1] the first
                              // list reservation that did not fit (else ~0u), [2] list entries reserved.  Reservations are
                              // handed out in order, so the entries below min([2], [1]) are exactly the ones that were
                              // written; a workgroup whose reservation did not fit finished its pixels itself
    unsigned* next;           // the counters of the NEXT launch, reset by k_fuse_open (no memset between launches)
    unsigned capacity;        // entries the arrays hold (a multiple of 64)
    unsigned quota;           // workgroups that find fewer than `quota` open pixels counted before theirs do not defer:
                              // they count in place.  A launch with only a handful of open pixels (clean data) then leaves the list
                              // empty and k_fuse_open returns at once -- a non-empty list costs its launch ~15 us of
                              // latency (one workgroup's serial pair loop), however short it is.
    unsigned inplace_min;     // a workgroup with at least this many open pixels (of 64) counts in place as well: its pair loop
                              // runs with most lanes busy, so handing the pixels over would only add the list traffic
                              // (data on which nothing fuses cleanly -- i.i.d.-noise images -- has all 64 open)
    long long* pix;           // [capacity] element index into the depth pool
    unsigned long long* vm;   // [capacity] accepted-hypothesis mask
    float2* hyp;              // [capacity/64][n][64] {rho, sigma}
};

// ---- the all-pairs count of InverseDepthHypothesisFusion and the fusion over the winning row (K1 and K1b) -------------------
// ChiTest is symmetric bit for bit (the squared difference and the float sum of the two quotients commute), so every
// unordered pair {a,b} is tested once and credited to both rows' set sizes: byte counters in LDS, row a in byte a&3 of word
// [a>>2][64] (a is wave-uniform, so the shift is a scalar; a 32-bit add of 1 << 8*(a&3) cannot carry over: sizes <= 64).
// Rows are dealt to the four waves in a zig-zag (w, 7-w, 8+w, 15-w, ...) that balances the triangular pair counts.
__device__ __forceinline__ void k1_count_all_pairs(const float2* hyp, const float* sgm, unsigned* cnt, unsigned long long vm,
                                                   int n, int p, int w)
{
    for (int i = 0; K1_WAVES * i < n; i++) {
        const int a = K1_WAVES * i + ((i & 1) ? (K1_WAVES - 1 - w) : w);
        if (a >= n || !((vm >> a) & 1ull)) continue;
        const float2 ha = hyp[a * K1_PX + p];
        const float sa = sgm[a * K1_PX + p];
        // the self pair: 0/s2 + 0/s2 is 0 (< 5.99) unless s2 = sigma*sigma is 0 or NaN (0/0)
        unsigned c = (sa * sa > 0.0f) ? 1u : 0u;
        for (int bb = a + 1; bb < n; bb++) {
            const float2 hb = hyp[bb * K1_PX + p];
            if (chi_test_lazy(ha, hb, sa, &sgm[bb * K1_PX + p])) {
                c++;
                atomicAdd(&cnt[(bb >> 2) * K1_PX + p], 1u << (8 * (bb & 3)));
            }
        }
        atomicAdd(&cnt[(a >> 2) * K1_PX + p], c << (8 * (a & 3)));
    }
}
// the first row with the largest count (PM.cc:616: strict '>'), its membership re-derived for GetFusion overload B
// (PM.cc:947-970, hypothesis order); false when that set has fewer than lambdaN members (PM.cc:623)
__device__ __forceinline__ bool k1_fuse_counted(const float2* hyp, const float* sgm, const unsigned* cnt, unsigned long long vm,
                                                int n, int p, int lambdaN, float2& result)
{
    unsigned best = 0;
    int besta = 0;
    for (int a = 0; a < n; a++) {
        if (!((vm >> a) & 1ull)) continue;
        const unsigned c = (cnt[(a >> 2) * K1_PX + p] >> (8 * (a & 3))) & 0xffu;
        if (c > best) {  // first largest set wins
            best = c;
            besta = a;
        }
    }
    if ((int)best < lambdaN) return false;  // PM.cc:623
    const float2 ha = hyp[besta * K1_PX + p];
    const float sa = sgm[besta * K1_PX + p];
    float pjsj = 0.f, rsj = 0.f;
    for (int bb = 0; bb < n; bb++) {
        if (!((vm >> bb) & 1ull)) continue;
        const float2 hb = hyp[bb * K1_PX + p];
        const bool in = (bb == besta) ? (sa * sa > 0.0f) : chi_test_lazy(ha, hb, sa, &sgm[bb * K1_PX + p]);
        if (in) fusion_accum(hb.x, sgm[bb * K1_PX + p], pjsj, rsj);
    }
    result = make_float2(pjsj / rsj, sqrtf(1 / rsj));  // PM.cc:225-226
    return true;
}

// 8 waves per SIMD (64 vector registers): with the search constants in vector registers (SDM_K1_OPT bit 8) hipcc would
// otherwise take 66 and lose a wave; the registers it spills instead live in the exact-division fallback of the matching
// cost, which runs for about one candidate in 10^5
#ifndef SDM_K1_EU_MIN
#define SDM_K1_EU_MIN 8
#endif
#ifndef SDM_K1_LB
#define SDM_K1_LB __launch_bounds__(K1_BLOCK) __attribute__((amdgpu_waves_per_eu(SDM_K1_EU_MIN, 8)))
#endif
// MASK: the instantiation whose searches may take the scan over the gate bit planes (long ranges, sdm_device.h scan_masked);
// the host launches it only for calls that have a pair with a long range at the principal point (stage_tables)
template <bool STATS, bool MASK>
__global__ SDM_K1_LB void k_search_fuse(const float4* __restrict__ rec, long long plane,
                                                       const RefConst* __restrict__ refs,
                                                       const PairConst* __restrict__ pairs, int n_ref, int n,
                                                       int W, int H, int max_chunks, DevParams prm,
                                                       const unsigned* __restrict__ act, float2* __restrict__ pool,
                                                       unsigned long long* __restrict__ stats, OpenList open_list,
                                                       const unsigned* __restrict__ gmask, int mrow)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // [n][64] {rho, 1/sigma^2 (NaN = take the exact path)}: all a pair test reads; sigma itself is needed only by
    // the exact path, the self pair and the final fusion sum
    float2* hyp = reinterpret_cast<float2*>(smem_raw);
    float* sgm = reinterpret_cast<float*>(hyp + (size_t)n * K1_PX);  // [n][64] sigma
    // compatible-set sizes (<= n <= 64), one byte each: row a lives in byte a&3 of word [a>>2][64].  a is
    // wave-uniform, so the shift is a scalar; a 32-bit LDS atomic add of 1<<8*(a&3) cannot carry over.
    unsigned* cnt = reinterpret_cast<unsigned*>(sgm + (size_t)n * K1_PX);
    const int cnt_words = max((n + 3) >> 2, K1_OUTLIER_ROWS);
    unsigned long long* pmask = reinterpret_cast<unsigned long long*>(cnt + (size_t)cnt_words * K1_PX);  // [4][64]
    unsigned long long* set0 = pmask + K1_BLOCK;  // [64] members (other than itself) of the FIRST hypothesis' compatible set
    unsigned* cnt0 = reinterpret_cast<unsigned*>(set0 + K1_PX);  // [64] size of that set (without itself)
    unsigned* xbase = cnt0 + K1_PX;  // [1] first open-list entry of this workgroup (or ~0u)

    // XCD-aware decode (blocks b and b+8 share an XCD): chunk c of EVERY reference keyframe runs on
    // XCD c % 8, reference index fastest, so the ~n keyframes that read the same region of a
    // neighbour image are resident on that XCD at the same time and share one L2 fill.
    const int b = blockIdx.x;
    const int i8 = b >> 3;
    const int cl = i8 / n_ref;
    const int ref = i8 - cl * n_ref;
    const int chunk = cl * 8 + (b & 7);
    if (chunk >= max_chunks) return;
    const RefConst rc = refs[ref];
    if (chunk * K1_PX >= rc.act_count) return;
    const int tid = threadIdx.x, p = tid & (K1_PX - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = chunk * K1_PX + p;
    const bool on = t < rc.act_count;
    const float4* __restrict__ rrec = rec + (long long)rc.slot * plane;

    int x = 2, y = 2;
    float pixel = 0.f, grad1 = 0.f, th_pi = 0.f, xp0 = 0.f, xp1 = 0.f;
    if (on) {
        unsigned xy = act[(long long)rc.slot * plane + t];
        x = (int)(xy & 0xffffu);
        y = (int)(xy >> 16);
        float4 r = rrec[y * W + x];
        pixel = (float)(int)(__float_as_uint(r.w) & 0xffu);  // PM.cc:202
        grad1 = r.x;
        th_pi = r.y;                       // PM.cc:214
        xp0 = ((float)x - rc.cx) / rc.fx;  // PM.cc:862
        xp1 = ((float)y - rc.cy) / rc.fy;
    }
    SearchStats st = {0, 0, 0};
    MaskStats ms = {0, 0, 0};
    MaskView mv;  // the neighbour's gate bit planes (sdm_device.h scan_masked); base per neighbour below
    mv.row_pitch = (unsigned)mrow * MASK_WORD_BYTES;  // mrow: 32-bit words per image row
    const PairConst* __restrict__ pcs = pairs + (long long)ref * n;
    const float rcvb[4] = {rc.fx, rc.cx, rc.mind, rc.maxd};
    const float* __restrict__ rcv = rcvb;
    // Hypotheses go to LDS in neighbour order.  "No hypothesis" (PM.cc:216) is stored as rho = +Inf,
    // sigma = 1: against any real hypothesis the squared difference is +Inf, so the compatibility
    // test fails on its fast path and the pair loops need no validity checks.  Each wave also keeps
    // the validity bits of its own neighbours; the four partial masks are OR-ed after the barrier.
    unsigned long long mymask = 0;
    for (int j = w; j < n; j += K1_WAVES) {
        const PairConst* __restrict__ pc = pcs + j;
        const float4* __restrict__ nrec = rec + (long long)pc->nbr_slot * plane;
        float2 h = make_float2(__builtin_inff(), 1.0f);
#if SDM_K1_OPT & 0x1000
        {  // all 64 lanes together (wave-uniform scan plan); lanes past the end of the list search nothing
            float rho, sigma, bu, bv;
            const float* __restrict__ cv = reinterpret_cast<const float*>(pc);
            bool ok = epipolar_search<STATS, true>(nrec, W, H, cv, rcv, pc->clean, on, x, y, pixel, grad1, th_pi, xp0, xp1, prm,
                                                   rho, sigma, bu, bv, &st);
            if (ok && __float_as_uint(rho) < 0x7f800000u) {  // PM.cc:216: 1/rho > 0  <=>  rho in [+0, +Inf) (denormals on)
                h = make_float2(rho, sigma);
                mymask |= 1ull << j;
            }
        }
#else
        if (on) {
            float rho, sigma, bu, bv;
            const float* __restrict__ cv = reinterpret_cast<const float*>(pc);
            mv.base = (MASK && gmask) ? reinterpret_cast<const char*>(gmask + (long long)pc->nbr_slot * H * MASK_PLANES * mrow) : nullptr;
            bool ok = epipolar_search<STATS, false, MASK>(nrec, W, H, cv, rcv, pc->clean, true, x, y, pixel, grad1, th_pi, xp0, xp1, prm, rho,
                                             sigma, bu, bv, &st, mv, &ms);
            if (ok && __float_as_uint(rho) < 0x7f800000u) {  // PM.cc:216: 1/rho > 0  <=>  rho in [+0, +Inf) (denormals on)
                h = make_float2(rho, sigma);
                mymask |= 1ull << j;
            }
        }
#endif
        hyp[j * K1_PX + p] = make_float2(h.x, safe_rcp_sq(h.y));
        sgm[j * K1_PX + p] = h.y;
    }
    for (int q = w; q < cnt_words; q += K1_WAVES) cnt[q * K1_PX + p] = 0u;
    if (w == 0) {
        set0[p] = 0ull;
        cnt0[p] = 0u;
    }
    pmask[tid] = mymask;
    __syncthreads();

    // InverseDepthHypothesisFusion, PM.cc:598-626: the first hypothesis whose compatible set is largest (strict '>',
    // PM.cc:616), then GetFusion over that set.
    unsigned long long vm = 0;
#pragma unroll
    for (int q = 0; q < K1_WAVES; q++) vm |= pmask[q * K1_PX + p];
#if SDM_ABLATE == 1
    const int nh = 0;
    if (vm == 0x123456789ull) pool[0] = make_float2(1.f, 1.f);
#else
    const int nh = __popcll(vm);
#endif
    const bool go = nh > prm.lambdaN;  // PM.cc:221
    // Branch-and-bound.  Let a0 be the FIRST accepted hypothesis, S0 its compatible set and O the accepted hypotheses
    // outside S0.  No row precedes a0, so S0 is the answer iff no row has a LARGER set:
    //  (1) O empty: a set holds at most the nh accepted hypotheses, so |S0| = nh is maximal -- nh-1 tests instead of
    //      nh(nh-1)/2;
    //  (2) O small: a row b in S0 can only exceed |S0| by being compatible with some o in O, so it suffices to test the
    //      rows o in O against everything: if no o is compatible with a member of S0 (then |set(b)| <= |S0| for every b in
    //      S0) and no o's own set is larger than S0, S0 stands -- (1 + |O|) rows instead of nh.
    // The four waves share each row's tests (b = w, w+4, ...).  Pixels neither case settles are handed to k_fuse_open
    // (the all-pairs count with 64 open pixels per workgroup); only when that list is full does this workgroup run the
    // pair loop itself.
    const int a0 = (int)__ffsll((long long)vm) - 1;  // first accepted hypothesis (-1 if none)
    bool self0 = false;
    if (go) {  // case (1) only needs the SIZE of row a0's set: one counter add per wave
        const float2 ha = hyp[a0 * K1_PX + p];
        const float sa = sgm[a0 * K1_PX + p];
        self0 = sa * sa > 0.0f;  // the self pair: 0/s2 + 0/s2 is 0 (< 5.99) unless s2 = sigma*sigma is 0 or NaN (0/0)
        unsigned c = 0;
        for (int bb = w; bb < n; bb += K1_WAVES) {
            const float2 hb = hyp[bb * K1_PX + p];  // "no hypothesis" rows hold rho = +Inf: never compatible
            if (bb != a0 && chi_test_lazy(ha, hb, sa, &sgm[bb * K1_PX + p])) c++;
        }
        if (c) atomicAdd(&cnt0[p], c);
    }
    __syncthreads();
#if SDM_ABLATE == 9  // diagnostic build: never take a shortcut (every fusing pixel runs the all-pairs count)
    bool settled = false;
    const bool few = false;
    const unsigned long long S0 = 0ull;
    const int s0 = 0;
#else
    bool settled = go && self0 && (int)cnt0[p] + 1 == nh;  // case (1)
    // case (2) needs the MEMBERS of S0: row a0 once more, for the workgroups that still have an open pixel (none on
    // clean data).  Lanes are pixels in every wave, so the ballots below have the same value in all four waves.
    const bool more = go && !settled && self0;
    unsigned long long S0 = settled ? vm : 0ull;
    if (__builtin_amdgcn_ballot_w64(more) != 0ull) {
        if (more) {
            const float2 ha = hyp[a0 * K1_PX + p];
            const float sa = sgm[a0 * K1_PX + p];
            unsigned long long m = 0;
            for (int bb = w; bb < n; bb += K1_WAVES) {
                const float2 hb = hyp[bb * K1_PX + p];
                if (bb != a0 && chi_test_lazy(ha, hb, sa, &sgm[bb * K1_PX + p])) m |= 1ull << bb;
            }
            if (m) atomicOr(&set0[p], m);
        }
        __syncthreads();
        if (more) S0 = set0[p] | (1ull << a0);
    }
    const int s0 = __popcll(S0);
    const unsigned long long O = vm & ~S0;
    // case (2) candidates: a0 compatible with itself, at most K1_OUTLIER_ROWS rows outside its set
    const bool few = more && __popcll(O) <= K1_OUTLIER_ROWS;
    if (__builtin_amdgcn_ballot_w64(few) != 0ull) {
        unsigned* orow = cnt;  // [K1_OUTLIER_ROWS][64]: set size of the i-th outlier row, + 0x10000 per member of S0 it is
                               // compatible with (the pair loop's counters are not in use yet; re-zeroed below)
        unsigned long long rem = few ? O : 0ull;
        for (int i = 0; i < K1_OUTLIER_ROWS && __builtin_amdgcn_ballot_w64(rem != 0ull) != 0ull; i++) {
            if (rem != 0ull) {
                const int o = (int)__ffsll((long long)rem) - 1;
                rem &= rem - 1ull;
                const float2 ho = hyp[o * K1_PX + p];
                const float so = sgm[o * K1_PX + p];
                unsigned c = (w == 0 && so * so > 0.0f) ? 1u : 0u;  // its self pair, counted once
                for (int bb = w; bb < n; bb += K1_WAVES) {
                    const float2 hb = hyp[bb * K1_PX + p];
                    if (bb != o && chi_test_lazy(ho, hb, so, &sgm[bb * K1_PX + p])) c += ((S0 >> bb) & 1ull) ? 0x10001u : 1u;
                }
                if (c) atomicAdd(&orow[i * K1_PX + p], c);
            }
        }
        __syncthreads();
        if (few) {
            bool stands = true;
            const int nO = __popcll(O);
            for (int i = 0; i < K1_OUTLIER_ROWS; i++) {
                const unsigned v = orow[i * K1_PX + p];
                if (i < nO) stands = stands && (v >> 16) == 0u && (int)(v & 0xffffu) <= s0;
            }
            settled = stands;
        }
        __syncthreads();  // every wave has read its rows before they become the pair loop's counters again
        for (int q = w; q < K1_OUTLIER_ROWS; q += K1_WAVES) cnt[q * K1_PX + p] = 0u;
    }
#endif
    // ---- pixels still open: defer them to k_fuse_open, or (list full) count all pairs here
    const bool open = go && !settled;
    const unsigned long long open_mask = __builtin_amdgcn_ballot_w64(open);  // the same in all four waves
    bool deferred = false;
    if (open_mask != 0ull) {
        if (tid == 0) {
            const unsigned cntw = (unsigned)__popcll(open_mask);
            unsigned first = 0xFFFFFFFFu;  // count in place
            // (a dense workgroup touches no counter at all: on data where every workgroup is dense, 300 k workgroups adding to
            // the same three words cost more than their pair loops)
            if (cntw < open_list.inplace_min && atomicAdd(&open_list.count[0], cntw) >= open_list.quota) {
                const unsigned base = atomicAdd(&open_list.count[2], cntw);  // entries base .. base+cntw-1 of the list
                if (base <= open_list.capacity && cntw <= open_list.capacity - base)
                    first = base;
                else
                    atomicMin(&open_list.count[1], base);
            }
            xbase[0] = first;
        }
        __syncthreads();  // also orders the counter re-zeroing above before the pair loop below
        const unsigned base = xbase[0];
        deferred = base != 0xFFFFFFFFu;
        if (deferred) {
            if (open) {
                // entry e of the list lives in block e/64, lane e%64: [block][hypothesis][lane] keeps k_fuse_open's loads
                // coalesced.  The four waves share the rows.
                const unsigned e = base + (unsigned)__popcll(open_mask & ((1ull << p) - 1ull));
                float2* __restrict__ dst = open_list.hyp + ((size_t)(e >> 6) * n) * K1_PX + (e & 63u);
                for (int j = w; j < n; j += K1_WAVES) dst[(size_t)j * K1_PX] = make_float2(hyp[j * K1_PX + p].x, sgm[j * K1_PX + p]);
                if (w == 0) {
                    open_list.pix[e] = (long long)rc.slot * plane + y * W + x;
                    open_list.vm[e] = vm;
                }
            }
        } else {
            if (open) k1_count_all_pairs(hyp, sgm, cnt, vm, n, p, w);
            __syncthreads();
        }
    }

    unsigned long long n_fused = 0;
    if (w == 0 && on && !(open && deferred)) {
        float2 result = make_float2(0.f, 0.f);  // a fresh depth_map_/depth_sigma_ entry (not fused)
        if (settled) {
            // the first hypothesis' set S0 stands (PM.cc:623 still asks for lambdaN members): GetFusion overload B over its
            // members, in hypothesis order, PM.cc:947-970
            if (s0 >= prm.lambdaN) {
                float pjsj = 0.f, rsj = 0.f;
                for (int bb = 0; bb < n; bb++) {
                    if (!((S0 >> bb) & 1ull)) continue;
                    fusion_accum(hyp[bb * K1_PX + p].x, sgm[bb * K1_PX + p], pjsj, rsj);
                }
                result = make_float2(pjsj / rsj, sqrtf(1 / rsj));  // PM.cc:225-226
                n_fused = 1;
            }
        } else if (go) {
            n_fused = k1_fuse_counted(hyp, sgm, cnt, vm, n, p, prm.lambdaN, result) ? 1 : 0;
        }
        // every listed pixel is written (fused value or zero); pixels outside the list are zero already
        pool[(long long)rc.slot * plane + y * W + x] = result;
    }
    if (STATS) {
        unsigned long long v[9] = {st.searches, st.candidates, st.gate_pass,
                                   (w == 0 && on) ? (unsigned long long)nh : 0ull, n_fused, ms.waves, ms.steps, ms.row_mismatch,
                                   (w == 0 && open) ? 1ull : 0ull};
#pragma unroll
        for (int k = 0; k < 9; k++) {
            unsigned long long s = v[k];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
            if ((tid & 63) == 0 && s) atomicAdd(&stats[k], s);
        }
    }
}

// ---- K1b: the all-pairs count for the pixels K1 left open, 64 of them per workgroup ---------------------------------------
// Grid-stride over the blocks of the launch's open list (the entry count lives on the device: no host read-back).  Same
// LDS layout and the same counting / fusion code as K1's in-place fallback, so the result is the same bit for bit.
template <bool STATS>
__global__ __launch_bounds__(K1_BLOCK) void k_fuse_open(OpenList open_list, int n, DevParams prm, float2* __restrict__ pool,
                                                        long long pool_elems, unsigned long long* __restrict__ stats)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float2* hyp = reinterpret_cast<float2*>(smem_raw);               // [n][64] {rho, 1/sigma^2}
    float* sgm = reinterpret_cast<float*>(hyp + (size_t)n * K1_PX);  // [n][64] sigma
    unsigned* cnt = reinterpret_cast<unsigned*>(sgm + (size_t)n * K1_PX);
    const int cnt_words = (n + 3) >> 2;
    // entries 0 .. total-1 were written (see OpenList::count); never more than the arrays hold
    const unsigned total = min(min(open_list.count[2], open_list.count[1]), open_list.capacity);
    const int tid = threadIdx.x, p = tid & (K1_PX - 1);
    if (blockIdx.x == 0 && tid == 0) {  // the next launch's counters: nobody uses them before this kernel has finished
        open_list.next[0] = 0u;
        open_list.next[1] = 0xFFFFFFFFu;
        open_list.next[2] = 0u;
    }
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long n_fused = 0;
    for (unsigned blk = blockIdx.x; blk * (unsigned)K1_PX < total; blk += gridDim.x) {
        const unsigned e = blk * K1_PX + p;
        const bool on = e < total;
        const float2* __restrict__ src = open_list.hyp + ((size_t)blk * n) * K1_PX + p;
        for (int j = w; j < n; j += K1_WAVES) {
            const float2 h = on ? src[(size_t)j * K1_PX] : make_float2(__builtin_inff(), 1.0f);
            hyp[j * K1_PX + p] = make_float2(h.x, safe_rcp_sq(h.y));
            sgm[j * K1_PX + p] = h.y;
        }
        for (int q = w; q < cnt_words; q += K1_WAVES) cnt[q * K1_PX + p] = 0u;
        const unsigned long long vm = on ? open_list.vm[e] : 0ull;
        __syncthreads();
        if (on) k1_count_all_pairs(hyp, sgm, cnt, vm, n, p, w);
        __syncthreads();
        if (w == 0 && on) {
            float2 result = make_float2(0.f, 0.f);
            if (k1_fuse_counted(hyp, sgm, cnt, vm, n, p, prm.lambdaN, result)) n_fused++;
            const long long px = open_list.pix[e];
            if (px >= 0 && px < pool_elems) pool[px] = result;  // (an entry is always in range; a store that is not never leaves)
        }
        __syncthreads();  // the LDS block is reused by the next list block
    }
    if (STATS) {
        for (int o = 32; o > 0; o >>= 1) n_fused += __shfl_down(n_fused, o);
        if ((tid & 63) == 0 && n_fused) atomicAdd(&stats[4], n_fused);
    }
}

// ---- shared halo loader for the 3x3 stencil kernels -----------------------------------------------------
constexpr int HALO_W = TILE_W + 2;
constexpr int HALO_H = TILE_H + 2;
__device__ __forceinline__ void load_halo(const float2* __restrict__ in, int W, int H, int tx0, int ty0,
                                          float2* tile)
{
    for (int i = threadIdx.x; i < HALO_W * HALO_H; i += BLOCK) {
        int hy = i / HALO_W, hx = i - hy * HALO_W;
        int x = tx0 + hx - 1, y = ty0 + hy - 1;
        float2 v = make_float2(0.f, 0.f);
        if (x >= 0 && x < W && y >= 0 && y < H) v = in[y * W + x];
        tile[i] = v;
    }
}

// ---- K2: IntraKeyFrameDepthChecking, PM.cc:486-547 -------------------------------------------------------
// jobs[i] = {in offset, out offset} (float2 elements) of reference i.
__global__ __launch_bounds__(BLOCK) void k_intra_check(const float2* __restrict__ in_base, float2* __restrict__ out_base,
                                                       const long long* __restrict__ in_off,
                                                       const long long* __restrict__ out_off, int n_ref, TileGeom g)
{
    __shared__ float2 tile[HALO_W * HALO_H];
    __shared__ float2 outv[TILE_PX];
    __shared__ unsigned short act[TILE_PX];
    __shared__ int wsum[16];
    int ref, tx0, ty0;
    if (!decode_block(g, n_ref, ref, tx0, ty0)) return;
    const int tid = threadIdx.x, W = g.W, H = g.H;
    const float2* __restrict__ in = in_base + in_off[ref];
    float2* __restrict__ out = out_base + out_off[ref];
    load_halo(in, W, H, tx0, ty0, tile);
    __syncthreads();
    bool f[PX_PER_THREAD];
#pragma unroll
    for (int i = 0; i < PX_PER_THREAD; i++) {
        int L = i * BLOCK + tid;
        int lx = L & (TILE_W - 1), ly = L >> 6;
        int x = tx0 + lx, y = ty0 + ly;
        float2 c = tile[(ly + 1) * HALO_W + lx + 1];
        outv[L] = c;  // depth_map_new = depth_map.clone(), PM.cc:488-489
        bool inset = (x >= 2 && x < W - 2 && y >= 2 && y < H - 2);
        f[i] = inset && (gt_1em6(c.x));  // PM.cc:497
    }
    const int nAct = block_compact(f, act, wsum);
    __syncthreads();
    for (int base = 0; base < nAct; base += BLOCK) {
        const int t = base + tid;
        if (t >= nAct) continue;
        const int L = act[t];
        const int lx = L & (TILE_W - 1), ly = L >> 6;
        const float2 c = tile[(ly + 1) * HALO_W + lx + 1];
        // compatible_neighbor_ho in raster order, then itself (PM.cc:504-522); GetFusion B streamed
        float pjsj = 0.f, rsj = 0.f, tmin = 0.f;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            if (k == 4) continue;
            float2 v = tile[(ly + k / 3) * HALO_W + lx + (k % 3)];
            if (gt_1em6(v.x) && chi_test(v.x, c.x, v.y, c.y)) {  // PM.cc:510-512
                if (cnt == 0) tmin = v.y;
                fusion_accum(v.x, v.y, pjsj, rsj);
                if ((double)v.y * (double)v.y < (double)tmin * (double)tmin) tmin = v.y;  // PM.cc:958
                cnt++;
            }
        }
        if (cnt == 0) tmin = c.y;
        fusion_accum(c.x, c.y, pjsj, rsj);
        if ((double)c.y * (double)c.y < (double)tmin * (double)tmin) tmin = c.y;
        cnt++;
        if (cnt >= 3)
            outv[L] = make_float2(pjsj / rsj, tmin);  // PM.cc:530-531 (sigma := min sigma)
        else
            outv[L] = make_float2(0.f, 0.f);  // PM.cc:535-536
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PX_PER_THREAD; i++) {
        int L = i * BLOCK + tid;
        int x = tx0 + (L & (TILE_W - 1)), y = ty0 + (L >> 6);
        if (x < W && y < H) out[y * W + x] = outv[L];
    }
}

// ---- K3: IntraKeyFrameDepthGrowing, PM.cc:549-596 --------------------------------------------------------
// grad is read with element stride gstride from grad_base + grad_off[ref] (4 for the record plane).
__global__ __launch_bounds__(BLOCK) void k_intra_grow(const float2* __restrict__ in_base, float2* __restrict__ out_base,
                                                      const long long* __restrict__ in_off,
                                                      const long long* __restrict__ out_off,
                                                      const float* __restrict__ grad_base,
                                                      const long long* __restrict__ grad_off, int gstride, int n_ref,
                                                      TileGeom g, float lambdaG)
{
    __shared__ float2 tile[HALO_W * HALO_H];
    __shared__ float2 outv[TILE_PX];
    __shared__ unsigned short act[TILE_PX];
    __shared__ int wsum[16];
    int ref, tx0, ty0;
    if (!decode_block(g, n_ref, ref, tx0, ty0)) return;
    const int tid = threadIdx.x, W = g.W, H = g.H;
    const float2* __restrict__ in = in_base + in_off[ref];
    float2* __restrict__ out = out_base + out_off[ref];
    const float* __restrict__ grad = grad_base + grad_off[ref];
    load_halo(in, W, H, tx0, ty0, tile);
    __syncthreads();
    bool f[PX_PER_THREAD];
#pragma unroll
    for (int i = 0; i < PX_PER_THREAD; i++) {
        int L = i * BLOCK + tid;
        int lx = L & (TILE_W - 1), ly = L >> 6;
        int x = tx0 + lx, y = ty0 + ly;
        float2 c = tile[(ly + 1) * HALO_W + lx + 1];
        outv[L] = c;
        bool inset = (x >= 2 && x < W - 2 && y >= 2 && y < H - 2);
        bool cand = inset && (lt_1em6(c.x));  // PM.cc:560
        if (cand) cand = !(grad[(long long)(y * W + x) * gstride] < lambdaG);  // PM.cc:562
        f[i] = cand;
    }
    const int nAct = block_compact(f, act, wsum);
    __syncthreads();
    for (int base = 0; base < nAct; base += BLOCK) {
        const int t = base + tid;
        if (t >= nAct) continue;
        const int L = act[t];
        const int lx = L & (TILE_W - 1), ly = L >> 6;
        const float2 c = tile[(ly + 1) * HALO_W + lx + 1];
        float pjsj = 0.f, rsj = 0.f, smin = 0.f;  // GetFusion overload A, PM.cc:926-945
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            if (k == 4) continue;
            float2 v = tile[(ly + k / 3) * HALO_W + lx + (k % 3)];
            if (chi_test(v.x, c.x, v.y, c.y)) {  // PM.cc:571
                if (cnt == 0) smin = v.y;
                fusion_accum(v.x, v.y, pjsj, rsj);
                if (v.y < smin) smin = v.y;  // PM.cc:939
                cnt++;
            }
        }
        if (cnt >= 2) outv[L] = make_float2(pjsj / rsj, smin);  // PM.cc:581-587
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PX_PER_THREAD; i++) {
        int L = i * BLOCK + tid;
        int x = tx0 + (L & (TILE_W - 1)), y = ty0 + (L >> 6);
        if (x < W && y < H) out[y * W + x] = outv[L];
    }
}

// ---- K2/K3, pipeline form ---------------------------------------------------------------------------------
// Inside SemiDenseRecon the map K2 reads was just written by K1: it is zero outside the keyframe's
// active-pixel list, and K2/K3 only ever modify listed pixels (K2: pixels with rho > 1e-6; K3:
// pixels with GradImg >= lambdaG).  So both passes run one thread per list entry with the 3x3
// neighbourhood gathered straight from the (L2-resident) map.  K2 reads the K1 map (zero outside the list) and
// writes the scratch plane at listed pixels only; K3 reads that plane at listed pixels only (a neighbour outside
// the list is a zero by construction, see intra_grow_pixel<true>) and writes the K1 map.  jobs: src/dst offsets.
__device__ __forceinline__ float2 intra_check_pixel(const float2* __restrict__ in, int W, int x, int y, float2 c)
{
    float pjsj = 0.f, rsj = 0.f, tmin = 0.f;  // GetFusion overload B streamed, PM.cc:947-970
    int cnt = 0;
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        if (k == 4) continue;
        v[k < 4 ? k : k - 1] = in[(y + k / 3 - 1) * W + x + (k % 3) - 1];
    }
    const float rc = safe_rcp_sq(c.y);
#pragma unroll
    for (int k = 0; k < 8; k++) {  // raster order, PM.cc:504-521
        if (gt_1em6(v[k].x) && chi_test_fast(v[k].x, c.x, v[k].y, c.y, safe_rcp_sq(v[k].y), rc)) {
            if (cnt == 0) tmin = v[k].y;
            fusion_accum(v[k].x, v[k].y, pjsj, rsj);
            if ((double)v[k].y * (double)v[k].y < (double)tmin * (double)tmin) tmin = v[k].y;
            cnt++;
        }
    }
    if (cnt == 0) tmin = c.y;
    fusion_accum(c.x, c.y, pjsj, rsj);  // itself, last: PM.cc:522
    if ((double)c.y * (double)c.y < (double)tmin * (double)tmin) tmin = c.y;
    cnt++;
    return (cnt >= 3) ? make_float2(pjsj / rsj, tmin) : make_float2(0.f, 0.f);  // PM.cc:524-536
}

// LISTED: `in` holds valid data only at pixels of the keyframe's active list (the scratch plane K2's list kernel
// wrote); any other neighbour is a zero of the pipeline map and is substituted instead of read, so that plane never
// has to be cleared.  rrec / H / lambdaG restate the list's membership rule (act_flag).
template <bool LISTED>
__device__ __forceinline__ float2 intra_grow_pixel(const float2* __restrict__ in, int W, int x, int y, float2 c,
                                                   const float4* __restrict__ rrec = nullptr, int H = 0,
                                                   float lambdaG = 0.f)
{
    // sigma_p == 0 (every pixel the pipeline left unsupported): ChiTest(.., sigma_p) is Delta^2/0 = Inf, or 0/0 = NaN
    // when Delta = 0 -- never below 5.99, so nothing can grow here and the 3x3 gather is skipped
    if (c.y == 0.0f) return c;
    float pjsj = 0.f, rsj = 0.f, smin = 0.f;  // GetFusion overload A, PM.cc:926-945
    int cnt = 0;
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        if (k == 4) continue;
        const int xn = x + (k % 3) - 1, yn = y + k / 3 - 1;
        float2 t = make_float2(0.f, 0.f);
        if (!LISTED || ((xn >= 2 && xn < W - 2 && yn >= 2 && yn < H - 2) && !(rrec[yn * W + xn].x < lambdaG)))
            t = in[yn * W + xn];
        v[k < 4 ? k : k - 1] = t;
    }
    const float rc = safe_rcp_sq(c.y);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (chi_test_fast(v[k].x, c.x, v[k].y, c.y, safe_rcp_sq(v[k].y), rc)) {  // PM.cc:571
            if (cnt == 0) smin = v[k].y;
            fusion_accum(v[k].x, v[k].y, pjsj, rsj);
            if (v[k].y < smin) smin = v[k].y;
            cnt++;
        }
    }
    return (cnt >= 2) ? make_float2(pjsj / rsj, smin) : c;  // PM.cc:581-587
}

template <bool GROW>
__global__ __launch_bounds__(BLOCK) void k_intra_list(const float2* __restrict__ src_base, float2* __restrict__ dst_base,
                                                      const long long* __restrict__ src_off,
                                                      const long long* __restrict__ dst_off,
                                                      const RefConst* __restrict__ refs, int first, int n_ref, int W,
                                                      int max_chunks, long long plane, const unsigned* __restrict__ act,
                                                      const float4* __restrict__ rec, int H, float lambdaG)
{
    int r, chunk;
    if (!decode_list_block<SDM_K23_GROUP, SDM_K23_BAND>(blockIdx.x, n_ref, max_chunks, r, chunk)) return;
    const RefConst rc = refs[first + r];
    const int t = chunk * BLOCK + threadIdx.x;
    if (t >= rc.act_count) return;
    const unsigned xy = act[(long long)rc.slot * plane + t];
    const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
    const float2* __restrict__ in = src_base + src_off[first + r];
    float2* __restrict__ out = dst_base + dst_off[first + r];
    const float2 c = in[y * W + x];
    float2 o = c;
    if (GROW) {
        if (lt_1em6(c.x))  // PM.cc:560 (gradient gate = list)
            o = intra_grow_pixel<true>(in, W, x, y, c, rec + (long long)rc.slot * plane, H, lambdaG);
    } else {
        if (gt_1em6(c.x)) o = intra_check_pixel(in, W, x, y, c);  // PM.cc:497
    }
    out[y * W + x] = o;
}

// ---- K2 + K3 on pipeline maps without a second plane ---------------------------------------------------------------------
// The list kernels above go pool -> scratch plane -> pool: four sparse passes (8-byte gathers / scatters at list pixels, ~20
// cache lines per wave instruction), and what bounds them is the number of line misses a CU keeps in flight.  Here K2
// writes its result COMPACTLY, indexed by list position (`out[r][t]`, 4 lines per wave instruction), a commit pass reads
// that back and scatters it into the pool (the Jacobi order is kept: every K2 read of the pool precedes every write), and
// K3 is reduced to what it really is on pipeline maps -- nothing, except for listed pixels that hold rho < 1e-6 WITH a
// non-zero sigma (PM.cc:560; sigma_p = 0 can never grow): K2 appends those to a (normally empty) list and
// one small kernel grows them, reads before writes.  Two sparse passes instead of four.
struct GrowList {
    unsigned* count;            // [0] candidates seen by this call's K2
    unsigned* next;             // the other call's counter, re-zeroed by k_grow
    unsigned capacity;          // entries of pix / val; beyond it the grow kernels walk the keyframes' whole lists instead
    long long* pix;             // pool index (slot * plane + y * W + x)
    float2* val;
};
// CHECK: run IntraKeyFrameDepthChecking (else the value passes through); DETECT: append K3's candidates
template <bool CHECK, bool DETECT>
__global__ __launch_bounds__(BLOCK) void k_intra_compact(const float2* __restrict__ pool, float2* __restrict__ out_base,
                                                         const long long* __restrict__ pool_off,
                                                         const long long* __restrict__ out_off,
                                                         const RefConst* __restrict__ refs, int first, int n_ref, int W,
                                                         int max_chunks, long long plane, const unsigned* __restrict__ act,
                                                         GrowList gl)
{
    int r, chunk;
    if (!decode_list_block<SDM_K23_GROUP, SDM_K23_BAND>(blockIdx.x, n_ref, max_chunks, r, chunk)) return;
    const RefConst rc = refs[first + r];
    const int t = chunk * BLOCK + threadIdx.x;
    const bool on = t < rc.act_count;
    float2 o = make_float2(0.f, 0.f);
    long long pix = 0;
    if (on) {
        const unsigned xy = act[(long long)rc.slot * plane + t];
        const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
        const float2* __restrict__ in = pool + pool_off[first + r];
        const float2 c = in[y * W + x];
        o = c;
        if (CHECK && gt_1em6(c.x)) o = intra_check_pixel(in, W, x, y, c);  // PM.cc:497
        if (CHECK) out_base[out_off[first + r] + t] = o;
        pix = pool_off[first + r] + y * W + x;
    }
    if (DETECT) {
        // PM.cc:560 (the gradient gate is the list) and the sigma_p = 0 exit of intra_grow_pixel
        const bool cand = on && lt_1em6(o.x) && !(o.y == 0.0f);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(cand);
        if (m != 0ull) {  // rare: one counter add per wave
            unsigned base = 0;
            if ((threadIdx.x & 63) == 0) base = atomicAdd(&gl.count[0], (unsigned)__popcll(m));
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (cand) {
                const unsigned e = base + (unsigned)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
                if (e < gl.capacity) gl.pix[e] = pix;
            }
        }
    }
}

// compact K2 results -> pool (every listed pixel)
__global__ __launch_bounds__(BLOCK) void k_intra_commit(float2* __restrict__ pool, const float2* __restrict__ in_base,
                                                        const long long* __restrict__ pool_off,
                                                        const long long* __restrict__ in_off,
                                                        const RefConst* __restrict__ refs, int first, int n_ref, int W,
                                                        int max_chunks, long long plane, const unsigned* __restrict__ act)
{
    int r, chunk;
    if (!decode_list_block<SDM_K23_GROUP, SDM_K23_BAND>(blockIdx.x, n_ref, max_chunks, r, chunk)) return;
    const RefConst rc = refs[first + r];
    const int t = chunk * BLOCK + threadIdx.x;
    if (t >= rc.act_count) return;
    const unsigned xy = act[(long long)rc.slot * plane + t];
    const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
    pool[pool_off[first + r] + y * W + x] = in_base[in_off[first + r] + t];
}

// K3 for the candidates: ONE workgroup (launched after the commit), so that "every read before any write" is a
// __syncthreads() -- phase 1 computes the grown values from the pool (K2's committed output; the pool is zero outside the
// lists, which is what the list form's "neighbours outside the list count as zero" means), phase 2 stores them.  With no
// candidate -- every launch on real data -- it costs one empty launch.  List overflow (count > capacity): walk every listed
// pixel of the batch [first, first + n_ref) instead and keep the values in the compact array.  Also re-zeroes the counter
// the NEXT call's K2 appends to.
constexpr int GROW_BLOCK = 1024;
__global__ __launch_bounds__(GROW_BLOCK) void k_grow(float2* __restrict__ pool, GrowList gl, int W, long long plane,
                                                     float2* __restrict__ cmp_base, const long long* __restrict__ pool_off,
                                                     const long long* __restrict__ cmp_off,
                                                     const RefConst* __restrict__ refs, int first, int n_ref,
                                                     const unsigned* __restrict__ act)
{
    const unsigned total = gl.count[0];
    if (threadIdx.x == 0) gl.next[0] = 0u;
    if (total == 0u) return;
    const bool listed = total <= gl.capacity;
    if (listed) {
        for (long long g = threadIdx.x; g < (long long)total; g += GROW_BLOCK) {
            const long long pix = gl.pix[g];
            const long long base = pix / plane * plane;
            const int p = (int)(pix - base);
            const int y = p / W, x = p - y * W;
            gl.val[g] = intra_grow_pixel<false>(pool + base, W, x, y, pool[pix]);
        }
    } else {
        for (int r = 0; r < n_ref; r++) {
            const RefConst rc = refs[first + r];
            const float2* __restrict__ in = pool + pool_off[first + r];
            for (long long t = threadIdx.x; t < rc.act_count; t += GROW_BLOCK) {
                const unsigned xy = act[(long long)rc.slot * plane + t];
                const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
                const float2 c = in[y * W + x];
                float2 o = c;
                if (lt_1em6(c.x) && !(c.y == 0.0f)) o = intra_grow_pixel<false>(in, W, x, y, c);
                cmp_base[cmp_off[first + r] + t] = o;
            }
        }
    }
    __threadfence();
    __syncthreads();  // every read of the pool above precedes every write below (Jacobi, PM.cc:551-552, 594-595)
    if (listed) {
        for (long long g = threadIdx.x; g < (long long)total; g += GROW_BLOCK) pool[gl.pix[g]] = gl.val[g];
    } else {
        for (int r = 0; r < n_ref; r++) {
            const RefConst rc = refs[first + r];
            float2* __restrict__ out = pool + pool_off[first + r];
            for (long long t = threadIdx.x; t < rc.act_count; t += GROW_BLOCK) {
                const unsigned xy = act[(long long)rc.slot * plane + t];
                const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
                out[y * W + x] = cmp_base[cmp_off[first + r] + t];  // unchanged for everything that was no candidate
            }
        }
    }
}

// ---- K4: InterKeyFrameDepthChecking, PM.cc:628-799 ------------------------------------------------------
struct __attribute__((packed, aligned(8))) Row2 {  // {rho,sigma} of two horizontally adjacent pixels
    float r0, s0, r1, s1;
};

// ---- One pixel of PM.cc:659-796: returns the new rho (0 = rejected, PM.cc:764) -------------------------------
// K4 spends its time in instruction issue, and with per-quotient guards the SCALAR pipe (one per CU: every
// exec-mask region and lane-mask operation goes through it, tools/ubench/salu.hip) was the longer pole: ~240
// scalar instructions per pixel-neighbour against ~330 vector ones.  So the per-neighbour body is straight-line:
//   * every quotient takes its reciprocal form unconditionally (Markstein: q = a*r, two FMA residual steps; 1/b as
//     v_rcp_f32 + one FMA step) -- bit-identical to the IEEE quotient whenever numerator and divisor magnitudes lie
//     in [2^-40, 2^41) (sdm_selftest(5)/(6); divisors with an all-ones significand included: round 4 dropped the
//     detector rounds 1-3 carried for them, tools/ubench/exact_ops.hip);
//   * instead of guarding each quotient, the running min/max of the operand magnitudes (as integers, so NaN and
//     Inf land above the window) are folded in with v_min3/v_max3;
//   * the four taps are predicated (operands of a tap that does not count are replaced by 1.0f, its terms are not
//     added); the 3.84 test is two multiplications and two comparisons, and a tap whose statistic is within
//     2^-13 of 3.84, or whose rho/sigma leave [2^-13, 2^13), raises the same flag;
//   * ONE test per neighbour: lanes whose flag is set redo that neighbour with the reference statement (plain
//     divisions, the double tap test).
struct K4Guard {
    unsigned hi, lo;  // max / min of |operand| bit patterns
};
constexpr unsigned K4_MAG_LO = QUOT_MAG_LO, K4_MAG_HI = QUOT_MAG_HI;  // the quotient window (sdm_device.h)
constexpr unsigned K4_TAP_LO = 114u << 23;          // 2^-13
constexpr unsigned K4_TAP_HI = (140u << 23) - 1u;  // just below 2^13
__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c) { return min(min(a, b), c); }
__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c) { return max(max(a, b), c); }
__device__ __forceinline__ void guard2(K4Guard& g, float a, float b)
{
    const unsigned ua = absbits(a), ub = absbits(b);
    g.hi = umax3(g.hi, ua, ub);
    g.lo = umin3(g.lo, ua, ub);
}
// the reference statement for one neighbour (PM.cc:677-755, 777-783), given the rows already fetched when the
// projection agreed (ra/rb are re-read here because an inexact fast projection may have addressed another pixel)
struct K4Sums {
    int kf_count;
    float sum_Jr, sum_JJ;
};
__device__ __noinline__ K4Sums inter_neighbour_exact(const float2* __restrict__ nb, const PairConst* __restrict__ pc,
                                                     int W, float colsm1, float rowsm1, float xp0, float xp1,
                                                     float depthp, float dp, K4Sums in)
{
    int kf_count = in.kf_count;
    float sum_Jr = in.sum_Jr, sum_JJ = in.sum_JJ;
    float t0 = row_dot_xp(pc->Rx, xp0, xp1) / depthp + pc->tx;  // PM.cc:678
    float t1 = row_dot_xp(pc->Ry, xp0, xp1) / depthp + pc->ty;
    float rzxp = row_dot_xp(pc->Rz, xp0, xp1);
    float t2 = rzxp / depthp + pc->tz;
    float u = pc->nfx * t0 + pc->ncx * t2;  // PM.cc:679
    float v = pc->nfy * t1 + pc->ncy * t2;
    float xj = u / t2, yj = v / t2;  // PM.cc:680
    float denom2 = depthp * pc->tz;
    float depthj = depthp / (rzxp + denom2);  // PM.cc:684-688
    if (!(xj >= 0 && xj < colsm1 && yj >= 0 && yj < rowsm1)) return in;  // PM.cc:695
    int x0 = (int)floorf(xj), y0 = (int)floorf(yj);
    const Row2 ra = *reinterpret_cast<const Row2*>(nb + y0 * W + x0);
    const Row2 rb = *reinterpret_cast<const Row2*>(nb + (y0 + 1) * W + x0);
    const float2 h[4] = {make_float2(ra.r0, ra.s0), make_float2(rb.r0, rb.s0), make_float2(ra.r1, ra.s1),
                         make_float2(rb.r1, rb.s1)};  // (y0,x0),(y1,x0),(y0,x1),(y1,x1): PM.cc:705-741
    int nj = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (!gt_1em6(h[k].x)) continue;
        float dd = depthj - h[k].x;
        float test = (float)(((double)dd * (double)dd) / ((double)h[k].y * (double)h[k].y));  // PM.cc:709
        if (!((double)test < 3.84)) continue;
        nj++;
        float djn = 1 / h[k].x;  // PM.cc:777-783
        float d2sigma = djn * djn * h[k].y;
        float J = (-rzxp) / d2sigma;                          // PM.cc:782
        float r0 = (djn - dp * rzxp - pc->tz) / d2sigma;    // PM.cc:783
        sum_Jr = sum_Jr + J * r0;
        sum_JJ = sum_JJ + J * J;
    }
    if (nj >= 1) kf_count++;  // PM.cc:755
    return K4Sums{kf_count, sum_Jr, sum_JJ};
}

// the straight-line form for one neighbour, in three steps so that the caller can have the NEXT neighbour's rows in flight
// while this one's taps are evaluated (SDM_K4_PIPE): projection -> row fetch -> taps; *slow is set for lanes whose result
// must not be used
struct K4Proj {
    unsigned off;  // byte offset of pixel (y0, x0) inside the neighbour's map (0 when the projection falls outside)
    bool valid;
    float den, rzxp;  // den = rzxp + rho tz: rho_j = rho / den (PM.cc:684-688), formed by inter_taps
    K4Guard g;
    float xj, yj;  // (read by sdm_selftest(7) only)
};
struct K4Rows {
    Row2 ra, rb;  // rows y0 and y0 + 1, columns x0 and x0 + 1
};
__device__ __forceinline__ K4Proj inter_project(const PairConst* __restrict__ pc, int W, float colsm1, float rowsm1, float xp0,
                                                float xp1, float depthp, float dp, const K4Guard& g0)
{
    K4Proj P;
    K4Guard g = g0;
    const float n0 = row_dot_xp(pc->Rx, xp0, xp1), n1 = row_dot_xp(pc->Ry, xp0, xp1);
    const float rzxp = row_dot_xp(pc->Rz, xp0, xp1);
    guard2(g, n0, n1);
    const float t0 = quot_fast(n0, depthp, dp) + pc->tx;  // PM.cc:678
    const float t1 = quot_fast(n1, depthp, dp) + pc->ty;
    const float t2 = quot_fast(rzxp, depthp, dp) + pc->tz;
    guard2(g, rzxp, t2);
    const float u = pc->nfx * t0 + pc->ncx * t2;  // PM.cc:679
    const float v = pc->nfy * t1 + pc->ncy * t2;
    guard2(g, u, v);
    const float r2 = rcp_fast(t2);
    const float xj = quot_fast(u, t2, r2), yj = quot_fast(v, t2, r2);  // PM.cc:680
    const float denom2 = depthp * pc->tz;
    P.den = rzxp + denom2;
    // PM.cc:695: 0 <= xj < cols-1 and 0 <= yj < rows-1 as ONE unsigned comparison per coordinate: x + 0.0f turns a -0
    // (which passes "x >= 0") into +0, a non-negative float orders like its bit pattern, negative values and NaN land
    // above every bound
    const unsigned bx = __float_as_uint(xj + 0.0f), by = __float_as_uint(yj + 0.0f);
    P.valid = (bx < __float_as_uint(colsm1)) & (by < __float_as_uint(rowsm1));
    // (int)xj truncates, which is floor for the valid (non-negative) coordinates; lanes that project outside fetch
    // pixel (0,0): the loads stay unconditional, their taps never count.  Byte offsets from the neighbour's plane
    // (8 bytes per pixel, < 2^30) keep the address arithmetic in 32 bits: one 24-bit multiply-add and a shift.
    const unsigned off = (__umul24((unsigned)cvt_i32_sat(yj), (unsigned)W) + (unsigned)cvt_i32_sat(xj)) << 3;
    P.off = P.valid ? off : 0u;
    P.rzxp = rzxp;
    P.xj = xj;
    P.yj = yj;
#if SDM_K4_DEPTHJ_QUOT
    {
        const unsigned ud = absbits(P.den);
        g.hi = max(g.hi, ud);
        g.lo = min(g.lo, ud);
    }
#endif
    P.g = g;
    return P;
}
// The same projection from the APPROXIMATE chain (see k4_proj_bounds): cell, validity and offset are the exact chain's
// unless *near is set (then the caller takes inter_project).  No quotient is formed here, so of the operand guards only
// rzxp's remains (it is the numerator of the taps' J).
#ifndef SDM_K4_APPROX
#define SDM_K4_APPROX 1
#endif
#ifndef SDM_K4_TAPABS
#define SDM_K4_TAPABS 0  // 1: |rho_n|, |sigma_n| in the taps' window test (rounds 1-3)
#endif
#ifndef SDM_K4_DEPTHJ_QUOT
#define SDM_K4_DEPTHJ_QUOT 1  // 1: rho_j's quotient in reciprocal form, its divisor in the operand window
#endif
#ifndef SDM_K4_TAPSEL
#define SDM_K4_TAPSEL 1  // 1: the taps' terms selected at the end instead of weighted; candidate-ness in the lane masks
#endif
#ifndef SDM_K4_NBRSKIP
#define SDM_K4_NBRSKIP 0  // 1: a neighbour in which no lane of the wave has a candidate tap skips the tap block (and rho_j's division)
#endif
#ifndef SDM_K4_TAPSKIP
#define SDM_K4_TAPSKIP 1  // 1: a tap that counts for no lane of the wave skips its Gauss-Newton terms (one ballot per tap)
#endif
#ifndef SDM_K4_CPRE
#define SDM_K4_CPRE 0  // 1: the next neighbour's constant block is requested (scalar loads) before this neighbour is evaluated
#endif
__device__ __forceinline__ K4Proj inter_project_approx(const PairConst* __restrict__ pc, int W, float colsm1, float rowsm1,
                                                       float xp0, float xp1, float depthp, float dp, const K4Guard& g0,
                                                       bool* near)
{
    K4Proj P;
    const float n0 = row_dot_xp(pc->Rx, xp0, xp1), n1 = row_dot_xp(pc->Ry, xp0, xp1);
    const float rzxp = row_dot_xp(pc->Rz, xp0, xp1);
    const float t0 = __builtin_fmaf(n0, dp, pc->tx), t1 = __builtin_fmaf(n1, dp, pc->ty), t2 = __builtin_fmaf(rzxp, dp, pc->tz);
    const float u = pc->nfx * t0 + pc->ncx * t2;
    const float v = pc->nfy * t1 + pc->ncy * t2;
    const float r2 = __builtin_amdgcn_rcpf(t2);
    const float xj = u * r2, yj = v * r2;
    const float Hh = __builtin_fmaf(pc->pb[0], dp, pc->pb[1]) * fabsf(r2);
    const float Gg = __builtin_fmaf(Hh, pc->pb[2], 3.0f * 0x1p-23f);
    const float ex = __builtin_fmaf(Gg, fabsf(xj), Hh), ey = __builtin_fmaf(Gg, fabsf(yj), Hh);
    // not "clearly inside one integer cell": within the bound of an integer, NaN anywhere, or too large to have a fraction
    *near = !(fabsf(xj - __builtin_rintf(xj)) > ex) | !(fabsf(yj - __builtin_rintf(yj)) > ey);
    const float denom2 = depthp * pc->tz;
    P.den = rzxp + denom2;
    const unsigned bx = __float_as_uint(xj + 0.0f), by = __float_as_uint(yj + 0.0f);
    P.valid = (bx < __float_as_uint(colsm1)) & (by < __float_as_uint(rowsm1));
    const unsigned off = (__umul24((unsigned)cvt_i32_sat(yj), (unsigned)W) + (unsigned)cvt_i32_sat(xj)) << 3;
    P.off = P.valid ? off : 0u;
    P.rzxp = rzxp;
    P.xj = xj;
    P.yj = yj;
    K4Guard g = g0;
    const unsigned uz = absbits(rzxp);
#if SDM_K4_DEPTHJ_QUOT
    // rho_j = rho / den in reciprocal form as well: den joins rzxp in the operand window (rho is in it already)
    const unsigned ud = absbits(P.den);
    g.hi = umax3(g.hi, uz, ud);
    g.lo = umin3(g.lo, uz, ud);
#else
    g.hi = max(g.hi, uz);
    g.lo = min(g.lo, uz);
#endif
    P.g = g;
    return P;
}
__device__ __forceinline__ K4Proj inter_project_any(const PairConst* __restrict__ pc, int W, float colsm1, float rowsm1,
                                                    float xp0, float xp1, float depthp, float dp, const K4Guard& g0)
{
#if SDM_K4_APPROX
    bool near;
    K4Proj P = inter_project_approx(pc, W, colsm1, rowsm1, xp0, xp1, depthp, dp, g0, &near);
    if (__builtin_expect(near, 0)) P = inter_project(pc, W, colsm1, rowsm1, xp0, xp1, depthp, dp, g0);
    return P;
#else
    return inter_project(pc, W, colsm1, rowsm1, xp0, xp1, depthp, dp, g0);
#endif
}
__device__ __forceinline__ K4Rows inter_fetch(const float2* __restrict__ nb, int W, unsigned off)
{
    const char* __restrict__ nbb = reinterpret_cast<const char*>(nb);
    K4Rows R;
    R.ra = *reinterpret_cast<const Row2*>(nbb + off);
    R.rb = *reinterpret_cast<const Row2*>(nbb + (size_t)W * 8 + off);
    return R;
}
__device__ __forceinline__ K4Sums inter_taps(const PairConst* __restrict__ pc, const K4Proj& P, const K4Rows& R, float depthp,
                                             float dp, K4Sums in, bool* slow)
{
    const float f0 = __uint_as_float(0x358637bdu);  // largest float below 1e-6 (gt_1em6)
    K4Guard g = P.g;
#if SDM_K4_DEPTHJ_QUOT
    const float rzxp = P.rzxp, depthj = quot_fast(depthp, P.den, rcp_fast(P.den));  // PM.cc:684-688; operands in P.g's window
#else
    const float rzxp = P.rzxp, depthj = depthp / P.den;  // PM.cc:684-688
#endif
    const Row2 ra = R.ra, rb = R.rb;
    const float hr[4] = {ra.r0, rb.r0, ra.r1, rb.r1};  // (y0,x0),(y1,x0),(y0,x1),(y1,x1): PM.cc:705-741
    const float hs[4] = {ra.s0, rb.s0, ra.s1, rb.s1};
    const float lim = P.valid ? f0 : __builtin_inff();  // rho_n > 1e-6 and the projection is inside
    float nsJr = in.sum_Jr, nsJJ = in.sum_JJ;
#if SDM_K4_TAPSEL
    bool anyc = false, ambb = false;
#else
    float njf = 0.0f;
    unsigned amb = 0;
#endif
    // candidate taps (rho_n > 1e-6, inside): rho_n and sigma_n must lie in [2^-13, 2^13).  That alone bounds
    // sigma^2, 1/rho_n and d2sigma = sigma/rho_n^2 inside the quotient window, so only r0's numerator is tracked besides.
    unsigned t_hi = K4_TAP_LO, t_lo = K4_TAP_LO;
    float rn[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool cand = hr[k] > lim;
#if SDM_K4_TAPSKIP >= 2
        rn[k] = 1.0f;
        if (__builtin_amdgcn_ballot_w64(cand) == 0ull) continue;  // wave-uniform: tap k is a candidate for no lane
#endif
        const float hx = cand ? hr[k] : 1.0f, sg = cand ? hs[k] : 1.0f;  // harmless operands otherwise
        // (no |.|: a candidate's rho_n is positive, and a negative sigma_n -- never produced by the pipeline -- has its sign
        // bit set: above the window as an unsigned number, so the lane takes the reference statement)
#if SDM_K4_TAPABS
        const unsigned uh = absbits(hx), us = absbits(sg);
#else
        const unsigned uh = __float_as_uint(hx), us = __float_as_uint(sg);
#endif
        t_hi = umax3(t_hi, uh, us);
        t_lo = umin3(t_lo, uh, us);
        // "test < 3.84" (PM.cc:709-710) without a division: with sigma in the window, dd^2 < 3.8397 sigma^2 and
        // dd^2 > 3.8403 sigma^2 (float products, relative error < 2^-22) are certain; in between the exact test
        // decides.  dd^2 = +Inf marks a tap that is not a candidate: never compatible, never ambiguous.
        const float dd = depthj - hr[k];
#if SDM_K4_TAPSEL
        // candidate-ness folded into the lane masks (scalar pipe) instead of a select on dd^2; the terms of a tap that does not
        // count are selected away at the end (no 0/1 weight, no substituted numerator: an incompatible tap whose numerator
        // leaves the operand window only sends its lane to the reference statement)
        const float dd2 = dd * dd;
        const float s2 = sg * sg;
        const bool c = cand & (dd2 < 3.8397f * s2);
        const bool sure_no = !cand | (dd2 > 3.8403f * s2);
        ambb |= !(c | sure_no);  // also NaN operands of a candidate
#if SDM_K4_TAPSKIP
        rn[k] = 1.0f;
        if (__builtin_amdgcn_ballot_w64(c) == 0ull) continue;  // wave-uniform: no lane's tap k counts
#endif
        const float djn = rcp_fast(hx);  // PM.cc:777-783
        const float d2sigma = djn * djn * sg;
        const float rd = rcp_fast(d2sigma);
        const float J = quot_fast(-rzxp, d2sigma, rd);      // PM.cc:782
        const float rnum = djn - dp * rzxp - pc->tz;        // PM.cc:783
        const float r0 = quot_fast(rnum, d2sigma, rd);
        rn[k] = rnum;
        const float sJr = nsJr + J * r0, sJJ = nsJJ + J * J;
        nsJr = c ? sJr : nsJr;
        nsJJ = c ? sJJ : nsJJ;
        anyc |= c;
    }
    const int nj = anyc ? 1 : 0;
    const unsigned amb = ambb ? 1u : 0u;
#else
        const float dd2 = cand ? dd * dd : __builtin_inff();
        const float s2 = sg * sg;
        const bool c = dd2 < 3.8397f * s2;
        const bool sure_no = dd2 > 3.8403f * s2;
        amb |= c ? 0u : (sure_no ? 0u : 1u);  // also NaN operands
#if SDM_K4_TAPSKIP
        rn[k] = 1.0f;
        if (__builtin_amdgcn_ballot_w64(c) == 0ull) continue;  // wave-uniform: no lane's tap k counts
#endif
        const float djn = rcp_fast(hx);  // PM.cc:777-783
        const float d2sigma = djn * djn * sg;
        const float rd = rcp_fast(d2sigma);
        const float J = quot_fast(-rzxp, d2sigma, rd);  // PM.cc:782
        float rnum = djn - dp * rzxp - pc->tz;          // PM.cc:783
        rnum = c ? rnum : 1.0f;
        const float r0 = quot_fast(rnum, d2sigma, rd);
        rn[k] = rnum;
        // a tap that does not count adds (finite) * 0 = +-0 to sums that are never -0 (they start at +0): the sums are the
        // reference's, and one select (the 0/1 weight) replaces three.  Finite: its operands are 1.0f or inside the tap
        // window, so |J| <= |rzxp| 2^39 and |r0| <= 2^39, and |rzxp| < 2^20 is part of the slow flag below.
        const float wgt = c ? 1.0f : 0.0f;
        nsJr = nsJr + (J * r0) * wgt;
        nsJJ = nsJJ + (J * J) * wgt;
        njf += wgt;
    }
    const int nj = (njf > 0.0f) ? 1 : 0;
#endif
    guard2(g, rn[0], rn[1]);
    guard2(g, rn[2], rn[3]);
    *slow = (g.lo < K4_MAG_LO) | (g.hi > K4_MAG_HI) | (amb != 0u) | (t_lo < K4_TAP_LO) |
            (t_hi > K4_TAP_HI) | (absbits(rzxp) >= 0x49800000u /* 2^20 */);
    return K4Sums{in.kf_count + nj, nsJr, nsJJ};  // PM.cc:755
}
__device__ __forceinline__ K4Sums inter_neighbour_fast(const float2* __restrict__ nb, const PairConst* __restrict__ pc,
                                                       int W, float colsm1, float rowsm1, float xp0, float xp1,
                                                       float depthp, float dp, const K4Guard& g0, K4Sums in,
                                                       bool* slow)
{
    const K4Proj P = inter_project_any(pc, W, colsm1, rowsm1, xp0, xp1, depthp, dp, g0);
    const K4Rows R = inter_fetch(nb, W, P.off);
#if SDM_K4_NBRSKIP
    {  // wave-uniform: no lane has a candidate tap (rho_n > 1e-6 at a valid projection, PM.cc:695, 705) in this neighbour --
       // the reference statement leaves count and sums alone as well, whatever the operand magnitudes
        const float lim = P.valid ? __uint_as_float(0x358637bdu) : __builtin_inff();
        const bool anyc = (R.ra.r0 > lim) | (R.rb.r0 > lim) | (R.ra.r1 > lim) | (R.rb.r1 > lim);
        if (__builtin_amdgcn_ballot_w64(anyc) == 0ull) {
            *slow = false;
            return in;
        }
    }
#endif
    return inter_taps(pc, P, R, depthp, dp, in, slow);
}

#ifndef SDM_K4_PIPE
#define SDM_K4_PIPE 0  // 1: the next neighbour's projection and row fetch are issued before this neighbour's taps are evaluated
#endif
__device__ __forceinline__ float inter_check_pixel(const float2* __restrict__ pool,
                                                   long long plane, const RefConst& rc,
                                                   const PairConst* __restrict__ pcs, int n, int W, int H, int x,
                                                   int y, float depthp, int lambdaN)
{
    const float colsm1 = (float)(W - 1), rowsm1 = (float)(H - 1);
    const float xp0 = ((float)x - rc.cx) / rc.fx, xp1 = ((float)y - rc.cy) / rc.fy;  // PM.cc:677
    const float dp = rcp_exact(depthp);                                                 // PM.cc:769
    // depthp is a divisor of every neighbour's three quotients: its checks are loop invariant
    K4Guard g0 = {absbits(depthp), absbits(depthp)};
    K4Sums acc = {0, 0.f, 0.f};
#if SDM_K4_PIPE
    K4Proj P = inter_project_any(pcs, W, colsm1, rowsm1, xp0, xp1, depthp, dp, g0);
    K4Rows R = inter_fetch(pool + (long long)pcs->nbr_slot * plane, W, P.off);
    for (int j = 0; j < n; j++) {
        const PairConst* __restrict__ pc = pcs + j;
        const float2* __restrict__ nb = pool + (long long)pc->nbr_slot * plane;
        // the next neighbour's rows are requested before this one's are consumed (the last round repeats its own: in cache)
        const PairConst* __restrict__ pcn = pcs + min(j + 1, n - 1);
        const K4Proj Pn = inter_project_any(pcn, W, colsm1, rowsm1, xp0, xp1, depthp, dp, g0);
        const K4Rows Rn = inter_fetch(pool + (long long)pcn->nbr_slot * plane, W, Pn.off);
        bool slow;
        const K4Sums fast = inter_taps(pc, P, R, depthp, dp, acc, &slow);
        if (__builtin_expect(slow, 0))
            acc = inter_neighbour_exact(nb, pc, W, colsm1, rowsm1, xp0, xp1, depthp, dp, acc);
        else
            acc = fast;
        P = Pn;
        R = Rn;
    }
#else
#if SDM_K4_CPRE
    PairConst nxt = pcs[0];
    for (int j = 0; j < n; j++) {
        const PairConst cur = nxt;
        if (j + 1 < n) nxt = pcs[j + 1];
        const float2* __restrict__ nb = pool + (long long)cur.nbr_slot * plane;
        bool slow;
        const K4Sums fast = inter_neighbour_fast(nb, &cur, W, colsm1, rowsm1, xp0, xp1, depthp, dp, g0, acc, &slow);
        if (__builtin_expect(slow, 0))
            acc = inter_neighbour_exact(nb, pcs + j, W, colsm1, rowsm1, xp0, xp1, depthp, dp, acc);
        else
            acc = fast;
    }
#else
    for (int j = 0; j < n; j++) {
        const PairConst* __restrict__ pc = pcs + j;
        const float2* __restrict__ nb = pool + (long long)pc->nbr_slot * plane;
        bool slow;
        const K4Sums fast = inter_neighbour_fast(nb, pc, W, colsm1, rowsm1, xp0, xp1, depthp, dp, g0, acc, &slow);
        if (__builtin_expect(slow, 0))
            acc = inter_neighbour_exact(nb, pc, W, colsm1, rowsm1, xp0, xp1, depthp, dp, acc);
        else
            acc = fast;
    }
#endif
#endif
    if (acc.kf_count < lambdaN) return 0.0f;      // PM.cc:764
    float dpDelta = (-acc.sum_Jr) / acc.sum_JJ;   // PM.cc:788-791
    return rcp_exact(dp + dpDelta);               // PM.cc:793
}

// Generic form: any depth map (e.g. uploaded by the caller).  64x16 tiles, in-tile compaction of
// the pixels that are not skipped by PM.cc:662, results staged in LDS, full-tile write-back.
__global__ __launch_bounds__(BLOCK) void k_inter_check(const float2* __restrict__ pool, long long plane,
                                                       const RefConst* __restrict__ refs,
                                                       const PairConst* __restrict__ pairs, int n_ref, int n,
                                                       TileGeom g, int lambdaN, float* __restrict__ chk)
{
    __shared__ float outv[TILE_PX];
    __shared__ unsigned short act[TILE_PX];
    __shared__ int wsum[16];
    int ref, tx0, ty0;
    if (!decode_block(g, n_ref, ref, tx0, ty0)) return;
    const int tid = threadIdx.x, W = g.W, H = g.H;
    const RefConst rc = refs[ref];
    const float2* __restrict__ cur = pool + (long long)rc.slot * plane;
    bool f[PX_PER_THREAD];
#pragma unroll
    for (int i = 0; i < PX_PER_THREAD; i++) {
        int L = i * BLOCK + tid;
        int x = tx0 + (L & (TILE_W - 1)), y = ty0 + (L >> 6);
        float r = (x < W && y < H) ? cur[y * W + x].x : 0.f;
        outv[L] = r;
        bool inset = (x >= 2 && x < W - 2 && y >= 2 && y < H - 2);  // PM.cc:659-660
        f[i] = inset && !(lt_1em6(r));                      // PM.cc:662
    }
    const int nAct = block_compact(f, act, wsum);
    __syncthreads();
    const PairConst* __restrict__ pcs = pairs + (long long)ref * n;
    for (int base = 0; base < nAct; base += BLOCK) {
        const int t = base + tid;
        if (t >= nAct) continue;
        const int L = act[t];
        const int x = tx0 + (L & (TILE_W - 1)), y = ty0 + (L >> 6);
        outv[L] = inter_check_pixel(pool, plane, rc, pcs, n, W, H, x, y, outv[L], lambdaN);
    }
    __syncthreads();
    float* __restrict__ out = chk + (long long)rc.slot * plane;
#pragma unroll
    for (int i = 0; i < PX_PER_THREAD; i++) {
        int L = i * BLOCK + tid;
        int x = tx0 + (L & (TILE_W - 1)), y = ty0 + (L >> 6);
        if (x < W && y < H) out[y * W + x] = outv[L];
    }
}

// one pixel of UpdateSemiDensePointSet (PM.cc:345-363): Pw = Twc * [Z(x-cx)/fx, Z(y-cy)/fy, Z, 1], Z = 1/rho
__device__ __forceinline__ void pointset_pixel(const KfMeta& m, int x, int y, float inv_d, float* __restrict__ o)
{
    if (lt_1em6(inv_d)) {  // PM.cc:345
        o[0] = 0.f;
        o[1] = 0.f;
        o[2] = 0.f;
        return;
    }
    float Rwc[9], Ow[3], tcw[3] = {m.Tcw[3], m.Tcw[7], m.Tcw[11]};  // Twc, src/KeyFrame.cc:70-84
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) Rwc[i * 3 + k] = m.Tcw[k * 4 + i];
    mat3_vec(Rwc, tcw, Ow);
    float Z = rcp_exact(inv_d);
    float X = Z * ((float)x - m.cx) / m.fx;
    float Y = Z * ((float)y - m.cy) / m.fy;
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = ((Rwc[i * 3 + 0] * X + Rwc[i * 3 + 1] * Y) + Rwc[i * 3 + 2] * Z) + (-Ow[i]) * 1.0f;
}

// Pipeline form: the depth map was produced by SemiDenseRecon, so every supported pixel is in the
// keyframe's active-pixel list (K1 writes only listed pixels, K2 only removes support, K3 grows
// only pixels with GradImg >= lambdaG).  One thread per list entry, no LDS; every listed pixel of the
// checked plane is written, the rest of it must already equal rho (= 0) -- else k_rho_copy runs first.
// XYZ: also back-project the checked value (the reference calls UpdateSemiDensePointSet right after
// InterKeyFrameDepthChecking, PM.cc:300-306): saves the second pass over the list and the checked plane.
#ifndef SDM_K4_BLOCK
#define SDM_K4_BLOCK 256  // threads per workgroup of the list kernel (64 / 128 / 256: A/B of the launch granularity, EXPERIMENTS.md)
#endif
constexpr int K4_BLOCK = SDM_K4_BLOCK;
template <bool XYZ>
__global__ __launch_bounds__(K4_BLOCK) void k_inter_check_list(const float2* __restrict__ pool, long long plane,
                                                            const RefConst* __restrict__ refs,
                                                            const PairConst* __restrict__ pairs, int n_ref, int n,
                                                            int W, int H, int max_chunks, int lambdaN,
                                                            const unsigned* __restrict__ act, float* __restrict__ chk,
                                                            const KfMeta* __restrict__ meta, float* __restrict__ xyz)
{
    int ref, chunk;
    if (!decode_list_block<SDM_K4_GROUP, SDM_K4_BAND>(blockIdx.x, n_ref, max_chunks, ref, chunk)) return;
    const RefConst rc = refs[ref];
    const int t = chunk * K4_BLOCK + threadIdx.x;
    if (t >= rc.act_count) return;
    const unsigned xy = act[(long long)rc.slot * plane + t];
    const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
    const long long o = (long long)rc.slot * plane + y * W + x;
    const float depthp = pool[o].x;
    float out = depthp;  // PM.cc:662: skipped pixels keep their value
    if (!(lt_1em6(depthp)))
        out = inter_check_pixel(pool, plane, rc, pairs + (long long)ref * n, n, W, H, x, y, depthp, lambdaN);
    chk[o] = out;
    if (XYZ) pointset_pixel(meta[rc.slot], x, y, out, xyz + o * 3);
}

// rho plane of the depth map -> checked plane
__global__ __launch_bounds__(BLOCK) void k_rho_copy(const float2* __restrict__ pool, float* __restrict__ chk,
                                                    long long plane, const int* __restrict__ slots, int n_ref)
{
    long long idx = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= plane * n_ref) return;
    int r = (int)(idx / plane);
    long long o = (long long)slots[r] * plane + (idx - (long long)r * plane);
    chk[o] = pool[o].x;
}

// chk -> depth map rho (the reference's in-place write, PM.cc:764/793)
__global__ __launch_bounds__(BLOCK) void k_commit(const float* __restrict__ chk, float2* __restrict__ pool,
                                                  long long plane, const int* __restrict__ slots, int n_ref)
{
    long long idx = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= plane * n_ref) return;
    int r = (int)(idx / plane);
    long long p = idx - (long long)r * plane;
    long long o = (long long)slots[r] * plane + p;
    pool[o].x = chk[o];
}

// depth map rho -> chk (so that sdm_pointset(source=1) is defined for never-checked slots is NOT
// needed; kept minimal on purpose)

// ---- K5: UpdateSemiDensePointSet, PM.cc:337-367 ------------------------------------------------------------
// src: rho read with element stride sstride from src_base + slot*plane*sstride.
__global__ __launch_bounds__(BLOCK) void k_pointset(const float* __restrict__ src_base, int sstride, long long plane,
                                                    const KfMeta* __restrict__ meta, const int* __restrict__ slots,
                                                    int n_ref, int W, int H, float* __restrict__ xyz)
{
    int r = blockIdx.y;
    int idx = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n_ref || idx >= W * H) return;
    int y = idx / W, x = idx - y * W;
    if (!(x >= 2 && x < W - 2 && y >= 2 && y < H - 2)) return;  // PM.cc:340-342
    const int slot = slots[r];
    const KfMeta m = meta[slot];
    // Twc = [Rcw^T | -Rcw^T tcw], src/KeyFrame.cc:70-84
    float Rwc[9], Ow[3], tcw[3] = {m.Tcw[3], m.Tcw[7], m.Tcw[11]};
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) Rwc[i * 3 + k] = m.Tcw[k * 4 + i];
    mat3_vec(Rwc, tcw, Ow);
    float inv_d = src_base[((long long)slot * plane + idx) * sstride];
    float* o = xyz + ((long long)slot * plane + idx) * 3;
    if (lt_1em6(inv_d)) {  // PM.cc:345
        o[0] = 0.f;
        o[1] = 0.f;
        o[2] = 0.f;
        return;
    }
    float Z = rcp_exact(inv_d);
    float X = Z * ((float)x - m.cx) / m.fx;
    float Y = Z * ((float)y - m.cy) / m.fy;
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = ((Rwc[i * 3 + 0] * X + Rwc[i * 3 + 1] * Y) + Rwc[i * 3 + 2] * Z) + (-Ow[i]) * 1.0f;
}

// K5, pipeline form: the source map is zero outside the keyframe's active list and the xyz plane is zero
// there already, so only listed pixels are (re)written.
__global__ __launch_bounds__(BLOCK) void k_pointset_list(const float* __restrict__ src_base, int sstride,
                                                         long long plane, const KfMeta* __restrict__ meta,
                                                         const RefConst* __restrict__ refs, int n_ref, int W,
                                                         int max_chunks, const unsigned* __restrict__ act,
                                                         float* __restrict__ xyz)
{
    int r, chunk;
    if (!decode_list_block<SDM_K23_GROUP, SDM_K23_BAND>(blockIdx.x, n_ref, max_chunks, r, chunk)) return;
    const RefConst rc = refs[r];
    const int t = chunk * BLOCK + threadIdx.x;
    if (t >= rc.act_count) return;
    const unsigned xy = act[(long long)rc.slot * plane + t];
    const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
    const long long idx = (long long)rc.slot * plane + y * W + x;
    pointset_pixel(meta[rc.slot], x, y, src_base[idx * sstride], xyz + idx * 3);
}

// ---- single-thread kernels behind the per-pixel C entry points ------------------------------------------------
__global__ void k_epipolar_search_px(const float4* __restrict__ rec, long long plane, const RefConst* refs,
                                     const PairConst* pairs, int W, int H, int x, int y, DevParams prm,
                                     float* __restrict__ out, const unsigned* __restrict__ gmask, int mrow)
{
    const RefConst rc = refs[0];
    const PairConst* pc = pairs;
    const float4* rrec = rec + (long long)rc.slot * plane;
    const float4* nrec = rec + (long long)pc->nbr_slot * plane;
    float4 r = rrec[y * W + x];
    float pixel = (float)(int)(__float_as_uint(r.w) & 0xffu);
    float xp0 = ((float)x - rc.cx) / rc.fx, xp1 = ((float)y - rc.cy) / rc.fy;
    float rho, sigma, bu, bv;
    SearchStats st = {0, 0, 0};
    const float* cv = reinterpret_cast<const float*>(pc);
    const float rcvb[4] = {rc.fx, rc.cx, rc.mind, rc.maxd};
    MaskStats ms = {0, 0, 0};
    MaskView mv;
    mv.base = gmask ? reinterpret_cast<const char*>(gmask + (long long)pc->nbr_slot * H * MASK_PLANES * mrow) : nullptr;
    mv.row_pitch = (unsigned)mrow * MASK_WORD_BYTES;  // mrow: 32-bit words per image row
    bool ok = epipolar_search<false, false, true>(nrec, W, H, cv, rcvb, pc->clean, true, x, y, pixel, r.x, r.y, xp0, xp1, prm, rho, sigma,
                                                  bu, bv, &st, mv, &ms);
    out[0] = rho;
    out[1] = sigma;
    out[2] = ok ? 1.f : 0.f;
    out[3] = bu;
    out[4] = bv;
}

__global__ void k_search_range_px(const RefConst* refs, const PairConst* pairs, int W, int x, int y,
                                  float* __restrict__ out)
{
    const RefConst rc = refs[0];
    float xp0 = ((float)x - rc.cx) / rc.fx, xp1 = ((float)y - rc.cy) / rc.fy;
    float rxxp = row_dot_xp(pairs->Rx, xp0, xp1), rzxp = row_dot_xp(pairs->Rz, xp0, xp1);
    float umin, umax;
    search_range(rc.fx, rc.cx, rxxp, rzxp, pairs->tx, pairs->tz, rc.mind, rc.maxd, W, umin, umax);
    out[0] = umin;
    out[1] = umax;
}

__global__ void k_fuse_px(const float2* __restrict__ hyp, int n, int lambdaN, float* __restrict__ out)
{
    float rho = 0.f, sigma = 0.f;
    bool ok = fuse_column(hyp, 1, n, lambdaN, rho, sigma);
    out[0] = ok ? rho : 0.f;
    out[1] = ok ? sigma : 0.f;
    out[2] = ok ? 1.f : 0.f;
}

// ---- arithmetic self-tests (tests/test_gpu_arith.py) ---------------------------------------------------------
// which = 0: div_theta fast path vs the plain double division for ALL 2^32 float bit patterns.
__global__ __launch_bounds__(BLOCK) void k_selftest_div(double d, double r, unsigned long long* __restrict__ bad)
{
    unsigned long long cnt = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
    for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < (1ull << 32); i += stride) {
        float f = __uint_as_float((unsigned)i);
        double x = (double)(f * f);  // the operand shape of PM.cc:436 (a squared float), plus...
        double a = div_theta(x, d, r, true), b = x / d;
        if (!(a == b || (a != a && b != b))) cnt++;
        double x2 = (double)fabsf(f);  // ...every non-negative float itself
        a = div_theta(x2, d, r, true);
        b = x2 / d;
        if (!(a == b || (a != a && b != b))) cnt++;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(bad, cnt);
}

// which = 6: rcp_exact vs the plain division for ALL 2^32 float bit patterns.  bad = mismatches, aux = operands
// that took the reciprocal path.
__global__ __launch_bounds__(BLOCK) void k_selftest_rcp(unsigned long long* __restrict__ bad,
                                                        unsigned long long* __restrict__ fast)
{
    unsigned long long cnt = 0, nf = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
    for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < (1ull << 32); i += stride) {
        const float b = __uint_as_float((unsigned)i);
        const float got = rcp_exact(b), want = 1.0f / b;
        if (!(__float_as_uint(got) == __float_as_uint(want) || (got != got && want != want))) cnt++;
        // sqrt_exact (the search's sqrtf(ustar_var), SDM_K1_OPT bit 16) against sqrtf, every bit pattern as well
        const float sg = sqrt_exact(b), sw = sqrtf(b);
        if (!(__float_as_uint(sg) == __float_as_uint(sw) || (sg != sg && sw != sw))) cnt++;
        const float ab = fabsf(b);
        if ((ab >= 0x1p-125f) & (ab < 0x1p125f)) nf++;
    }
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_down(cnt, o);
        nf += __shfl_down(nf, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (cnt) atomicAdd(bad, cnt);
        atomicAdd(fast, nf);
    }
}

// which = 1: chi_test_fast vs chi_test on pseudo-random operands concentrated around the 5.99
// threshold, plus special values (0, denormal, Inf, NaN sigmas).
__device__ __forceinline__ unsigned xs32(unsigned& s)
{
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
}

// which = 7: K4's straight-line per-neighbour body vs the reference statement on random geometry and 2x2 depth
// patches (log-uniform magnitudes far beyond the fast windows, tap statistics concentrated on the 3.84 threshold,
// zero / Inf / NaN operands).  bad = lanes whose fast result was accepted (slow flag clear) but differs from the
// exact one; aux = lanes that were accepted.
__device__ __forceinline__ float st_uniform(unsigned& s) { return (xs32(s) >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float st_logmag(unsigned& s, float lo_exp, float hi_exp)
{
    return exp2f(lo_exp + (hi_exp - lo_exp) * st_uniform(s));
}
__global__ __launch_bounds__(BLOCK) void k_selftest_k4(int iters, float2* __restrict__ patches,
                                                       PairConst* __restrict__ pcs, unsigned long long* __restrict__ bad,
                                                       unsigned long long* __restrict__ accepted, int report_projections)
{
    const int gid = blockIdx.x * BLOCK + threadIdx.x;
    unsigned s = 0x9E3779B9u * (gid + 1);
    float2* nb = patches + (long long)gid * 6;  // a private 3x2 map (W = 3): rows y0, y0+1 at x0 = 0 or 1
    PairConst* pc = pcs + gid;
    unsigned long long nbad = 0, nacc = 0, nchk = 0;
    for (int it = 0; it < iters; it++) {
        const unsigned mode = xs32(s) & 7u;
        // geometry: near-identity rotation rows, small translation, pixel-scale intrinsics; sometimes wild
        const float wild = (mode == 0) ? 1.0f : 0.0f;
        for (int i = 0; i < 9; i++) {
            float* row = (i < 3) ? pc->Rx : (i < 6) ? pc->Ry : pc->Rz;
            row[i % 3] = ((i % 4 == 0) ? 1.0f : 0.0f) + (st_uniform(s) - 0.5f) * (0.1f + wild);
        }
        pc->tx = (st_uniform(s) - 0.5f) * (0.2f + 10.0f * wild);
        pc->ty = (st_uniform(s) - 0.5f) * (0.2f + 10.0f * wild);
        pc->tz = (st_uniform(s) - 0.5f) * (0.2f + 10.0f * wild);
        if (mode == 1) pc->tz = 0.0f;
        pc->nfx = 1.0f + st_uniform(s);
        pc->nfy = 1.0f + st_uniform(s);
        pc->ncx = 0.75f + st_uniform(s);  // (below 1 the approximate projection is switched off for the pair)
        pc->ncy = 0.75f + st_uniform(s);
        if (mode == 5) {  // pixel-scale intrinsics: coordinates in the hundreds (mostly outside the 3x2 map)
            pc->nfx *= 400.0f;
            pc->nfy *= 400.0f;
            pc->ncx *= 300.0f;
            pc->ncy *= 200.0f;
        }
        const float xp0 = (st_uniform(s) - 0.5f) * 0.6f, xp1 = (st_uniform(s) - 0.5f) * 0.6f;
        k4_proj_bounds(*pc, 0.3f, 0.3f);
        float depthp = (mode == 2) ? st_logmag(s, -60.f, 60.f) : st_logmag(s, -4.f, 4.f);
        if (mode == 3) depthp = __uint_as_float((__float_as_uint(depthp) | 0x7FFFFFu));  // significand all ones
        const float dp = rcp_exact(depthp);
        // what rho_j will be for this geometry (any value works; used to put the taps near the threshold)
        const float rz = row_dot_xp(pc->Rz, xp0, xp1);
        const float depthj = depthp / (rz + depthp * pc->tz);
        for (int k = 0; k < 6; k++) {
            float sg = (mode == 4) ? st_logmag(s, -30.f, 30.f) : st_logmag(s, -9.f, -2.f);
            const float crit = 1.9595918f * sg;  // sqrt(3.84) sigma
            float off;
            const unsigned m2 = xs32(s) & 3u;
            if (m2 == 0) off = crit * (1.0f + (st_uniform(s) - 0.5f) * 4.0e-4f);      // inside / around the band
            else if (m2 == 1) off = crit * (1.0f + (st_uniform(s) - 0.5f) * 2.0e-6f); // rounding distance
            else off = crit * st_uniform(s) * 3.0f;
            float rho = depthj + ((xs32(s) & 1u) ? off : -off);
            const unsigned m3 = xs32(s) & 31u;
            if (m3 == 0) rho = 0.0f;
            if (m3 == 1) sg = 0.0f;
            if (m3 == 2) rho = __builtin_nanf("");
            if (m3 == 3) sg = __builtin_inff();
            if (m3 == 4) rho = 5.0e-7f;
            if (m3 == 5) sg = __uint_as_float((__float_as_uint(sg) | 0x7FFFFFu));
            nb[k] = make_float2(rho, sg);
        }
        __threadfence_block();
        K4Guard g0 = {absbits(depthp), absbits(depthp)};
        {  // the approximate projection, where it does not ask for the exact one, names the exact chain's cell and validity
            bool near;
            const K4Proj Pa = inter_project_approx(pc, 3, 2.0f, 1.0f, xp0, xp1, depthp, dp, g0, &near);
            const float e0 = row_dot_xp(pc->Rx, xp0, xp1) / depthp + pc->tx, e1 = row_dot_xp(pc->Ry, xp0, xp1) / depthp + pc->ty;
            const float e2 = rz / depthp + pc->tz;
            const float xe = (pc->nfx * e0 + pc->ncx * e2) / e2, ye = (pc->nfy * e1 + pc->ncy * e2) / e2;
            const bool ve = xe >= 0 && xe < 2.0f && ye >= 0 && ye < 1.0f;
            const bool window = (g0.lo >= K4_MAG_LO) & (g0.hi <= K4_MAG_HI);  // rho outside it is slow whatever the projection says
            if (!near && window) {
                nchk++;
                const bool same = ve == Pa.valid && floorf(xe) == floorf(Pa.xj) && floorf(ye) == floorf(Pa.yj) &&
                                  (!ve || Pa.off == (((unsigned)(int)floorf(ye) * 3u + (unsigned)(int)floorf(xe)) << 3));
                if (!same) nbad++;
            }
        }
        const K4Sums in = {(int)(xs32(s) & 3u), (st_uniform(s) - 0.5f) * 10.f, st_uniform(s) * 10.f};
        bool slow;
        // W = 3, H = 2: valid projections have 0 <= xj < 2, 0 <= yj < 1
        const K4Sums a = inter_neighbour_fast(nb, pc, 3, 2.0f, 1.0f, xp0, xp1, depthp, dp, g0, in, &slow);
        const K4Sums b = inter_neighbour_exact(nb, pc, 3, 2.0f, 1.0f, xp0, xp1, depthp, dp, in);
        if (!slow) {
            nacc++;
            const bool same = a.kf_count == b.kf_count &&
                              (__float_as_uint(a.sum_Jr) == __float_as_uint(b.sum_Jr) || (a.sum_Jr != a.sum_Jr && b.sum_Jr != b.sum_Jr)) &&
                              (__float_as_uint(a.sum_JJ) == __float_as_uint(b.sum_JJ) || (a.sum_JJ != a.sum_JJ && b.sum_JJ != b.sum_JJ));
            if (!same) nbad++;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        nbad += __shfl_down(nbad, o);
        nacc += __shfl_down(nacc, o);
        nchk += __shfl_down(nchk, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (nbad) atomicAdd(bad, nbad);
        atomicAdd(accepted, report_projections ? nchk : nacc);  // which = 9: projections decided by the approximate chain
    }
}

// which = 8: two scan identities over EVERY float bit pattern: (a) the lerp weight 1 - fract(yf) vs (floor(yf)+1) - yf for
// yf >= 0 (the scan and the refinement only form it for rows yf in [0, H-1)); (b) the integer-mask wrap of
// PM.cc:425-426 vs the compare statement -- values must agree bit for bit except that a -0 result may come back as +0
// (wrap_once_360's note), NaN matches NaN.  bad = mismatches, aux = operands tested.
__global__ __launch_bounds__(BLOCK) void k_selftest_scan_ids(unsigned long long* __restrict__ bad,
                                                             unsigned long long* __restrict__ tested)
{
    unsigned long long cnt = 0, n = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
    for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < (1ull << 32); i += stride) {
        const float f = __uint_as_float((unsigned)i);
        if (f >= 0.0f && f < 16777216.0f) {
            const float a = 1.0f - __builtin_amdgcn_fractf(f), b = (floorf(f) + 1.0f) - f;
            if (__float_as_uint(a) != __float_as_uint(b)) cnt++;
            n++;
        }
        {
            const float a = wrap_once_360(f), b = wrap_once_360_ref(f);
            const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b) || (a == 0.0f && b == 0.0f);
            if (!same) cnt++;
            n++;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_down(cnt, o);
        n += __shfl_down(n, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (cnt) atomicAdd(bad, cnt);
        atomicAdd(tested, n);
    }
}

__global__ __launch_bounds__(BLOCK) void k_selftest_chi(int iters, unsigned long long* __restrict__ bad,
                                                        unsigned long long* __restrict__ inband)
{
    unsigned s = 0x9E3779B9u * (blockIdx.x * BLOCK + threadIdx.x + 1);
    unsigned long long cnt = 0, band = 0;
    const float specials[8] = {0.f, 1e-41f, 1e-20f, 1e20f, __builtin_inff(), __builtin_nanf(""), 1e-16f, 3e15f};
    for (int i = 0; i < iters; i++) {
        float u1 = (xs32(s) >> 8) * (1.0f / 16777216.0f), u2 = (xs32(s) >> 8) * (1.0f / 16777216.0f);
        float u3 = (xs32(s) >> 8) * (1.0f / 16777216.0f), u4 = (xs32(s) >> 8) * (1.0f / 16777216.0f);
        float sa = 0.001f + u1, sb = 0.001f + u2;
        float a = 1.0f + u3;
        // choose b so that chi lands within ~1e-6 relative of 5.99 half of the time
        float target = (i & 1) ? 5.99f * (1.0f + (u4 - 0.5f) * 4e-6f) : 12.0f * u4;
        float k = 1.0f / (sa * sa) + 1.0f / (sb * sb);
        float b = a + sqrtf(target / k);
        unsigned pick = xs32(s);
        if ((pick & 1023u) == 0) sa = specials[(pick >> 10) & 7];
        if ((pick & 1023u) == 1) sb = specials[(pick >> 10) & 7];
        if ((pick & 1023u) == 2) b = a;
        if ((pick & 1023u) == 3) b = specials[(pick >> 10) & 7];
        bool e = chi_test(a, b, sa, sb);
        bool f = chi_test_fast(a, b, sa, sb, safe_rcp_sq(sa), safe_rcp_sq(sb));
        if (e != f) cnt++;
        float d = a - b, num = d * d;
        float chi = num / (sa * sa) + num / (sb * sb);
        if (chi > 5.9896f && chi < 5.9904f) band++;
    }
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_down(cnt, o);
        band += __shfl_down(band, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (cnt) atomicAdd(bad, cnt);
        if (band) atomicAdd(inband, band);
    }
}

// which = 2: match_cost (reciprocal fast path + mid-point fallback) vs the reference expression.
// Operands: random squares, plus pairs CONSTRUCTED so that the double sum lands on / next to a float
// rounding midpoint (forces the fallback and the near-miss cases on both sides).
__global__ __launch_bounds__(BLOCK) void k_selftest_cost(DevParams prm, int iters, unsigned long long* __restrict__ bad,
                                                         unsigned long long* __restrict__ risky_hits)
{
    unsigned s = 0x85EBCA6Bu * (blockIdx.x * BLOCK + threadIdx.x + 1);
    unsigned long long cnt = 0, hits = 0;
    for (int i = 0; i < iters; i++) {
        float ge = ((xs32(s) >> 8) * (1.0f / 16777216.0f)) * 60.0f;
        float ge2 = ge * ge;
        float pe2;
        if (i & 1) {
            float pe = ((xs32(s) >> 8) * (1.0f / 16777216.0f)) * 255.0f;
            pe2 = pe * pe;
        } else {
            // choose a small pe2 so that pe2 + ge2/theta sits within a few double-ulps of the first
            // float rounding midpoint above q = ge2/theta (pe2 is tiny, so its own grid is fine enough)
            double q = (double)ge2 / prm.theta_var;
            float lowf = (float)q;
            if ((double)lowf > q) lowf = __uint_as_float(__float_as_uint(lowf) - 1u);
            float upf = __uint_as_float(__float_as_uint(lowf) + 1u);
            double mid = ((double)lowf + (double)upf) * 0.5;
            if (mid < q) mid = ((double)upf + (double)__uint_as_float(__float_as_uint(upf) + 1u)) * 0.5;
            pe2 = (float)(mid - q);
            int k = (int)(xs32(s) % 5u) - 2;
            if (pe2 > 1e-30f) pe2 = __uint_as_float(__float_as_uint(pe2) + k);
        }
        float ref = (float)((double)pe2 + (double)ge2 / prm.theta_var);
        float got = match_cost(pe2, ge2, prm);
        if (!(ref == got || (ref != ref && got != got))) cnt++;
        got = match_cost1(pe2, ge2, prm);
        if (!(ref == got || (ref != ref && got != got))) cnt++;
        // magnitudes around and beyond the guard of both forms (tiny / huge squares)
        const float sc = exp2f((float)((int)(xs32(s) % 400u) - 200));
        const float pe2s = pe2 * sc, ge2s = ge2 * sc;
        ref = (float)((double)pe2s + (double)ge2s / prm.theta_var);
        got = match_cost1(pe2s, ge2s, prm);
        if (!(ref == got || (ref != ref && got != got))) cnt++;
        got = match_cost(pe2s, ge2s, prm);
        if (!(ref == got || (ref != ref && got != got))) cnt++;
        // the scan's approximate cost (SDM_K1_OPT bit 13) must stay within COST_BAND / 4 float steps of the reference value
        // (two costs further than COST_BAND steps apart then compare like their exact values): default theta only
        if (prm.fast_theta_div) {
            const float r1 = (float)((double)pe2 + (double)ge2 / prm.theta_var);
            const unsigned d1 = __float_as_uint(match_cost_approx(pe2, ge2)) - __float_as_uint(r1) + COST_BAND / 4u;
            const unsigned d2 = __float_as_uint(match_cost_approx(pe2s, ge2s)) - __float_as_uint(ref) + COST_BAND / 4u;
            if (r1 == r1 && d1 > COST_BAND / 2u) cnt++;
            if (ref == ref && d2 > COST_BAND / 2u) cnt++;
        }
        {  // ... and the same with the run-time constant of the theta_var in force (DevParams::inv_theta_f / approx_ok)
            const float r1 = (float)((double)pe2 + (double)ge2 / prm.theta_var);
            const unsigned d1 = __float_as_uint(__builtin_fmaf(ge2, prm.inv_theta_f, pe2)) - __float_as_uint(r1) + COST_BAND / 4u;
            const unsigned d2 = __float_as_uint(__builtin_fmaf(ge2s, prm.inv_theta_f, pe2s)) - __float_as_uint(ref) + COST_BAND / 4u;
            if (r1 == r1 && d1 > COST_BAND / 2u) cnt++;
            if (ref == ref && d2 > COST_BAND / 2u) cnt++;
        }
        double sfast = (double)pe2 + (double)ge2 * prm.inv_theta;
        unsigned lo = (unsigned)__double2loint(sfast);
        if (((lo & 0x1FFFFFFFu) - 0x0FFFFF00u) <= 0x200u) hits++;
    }
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_down(cnt, o);
        hits += __shfl_down(hits, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (cnt) atomicAdd(bad, cnt);
        if (hits) atomicAdd(risky_hits, hits);
    }
}

// which = 4: the shared-reciprocal form of GetFusion's two double quotients vs two plain divisions,
// over random (rho, sigma) across many magnitudes and the special values.
__global__ __launch_bounds__(BLOCK) void k_selftest_fusion_terms(int iters, unsigned long long* __restrict__ bad,
                                                                 unsigned long long* __restrict__ tested)
{
    unsigned s = 0x27D4EB2Fu * (blockIdx.x * BLOCK + threadIdx.x + 1);
    unsigned long long cnt = 0, n = 0;
    const float specials[8] = {0.f, 1e-41f, 1e-25f, 1e25f, __builtin_inff(), __builtin_nanf(""), -1.0f, 3e38f};
    for (int i = 0; i < iters; i++) {
        // random bit patterns restricted to exponents in a wide band, so every mantissa occurs
        unsigned br = xs32(s), bs = xs32(s);
        float rho = __uint_as_float((br & 0x807FFFFFu) | ((100u + (br >> 23) % 56u) << 23));
        float sg = __uint_as_float((bs & 0x007FFFFFu) | ((100u + (bs >> 23) % 56u) << 23));
        unsigned pick = xs32(s);
        if ((pick & 255u) == 0) rho = specials[(pick >> 8) & 7];
        if ((pick & 255u) == 1) sg = specials[(pick >> 8) & 7];
        double t_rho, t_one;
        fusion_terms(rho, sg, t_rho, t_one);
        double s2 = (double)sg * (double)sg;
        double e_rho = (double)rho / s2, e_one = 1.0 / s2;
        if (!(t_rho == e_rho || (t_rho != t_rho && e_rho != e_rho))) cnt++;
        if (!(t_one == e_one || (t_one != t_one && e_one != e_one))) cnt++;
        n++;
    }
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_down(cnt, o);
        n += __shfl_down(n, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (cnt) atomicAdd(bad, cnt);
        atomicAdd(tested, n);
    }
}

// which = 5: reciprocal-form float quotient (quot_fast, K4) vs the IEEE division, wherever quot_window_ok holds;
// operands are spread over and beyond the window, all significands, both signs.  tested = pairs inside the window.
__global__ __launch_bounds__(BLOCK) void k_selftest_quot(int iters, unsigned long long* __restrict__ bad,
                                                         unsigned long long* __restrict__ tested)
{
    unsigned s = 0x165667B1u * (blockIdx.x * BLOCK + threadIdx.x + 1);
    unsigned long long cnt = 0, n = 0;
    for (int i = 0; i < iters; i++) {
        unsigned ba = xs32(s), bb = xs32(s);
        float a = __uint_as_float((ba & 0x807FFFFFu) | ((80u + (ba >> 23) % 96u) << 23));   // 2^-47 .. 2^48
        float b = __uint_as_float((bb & 0x807FFFFFu) | ((80u + (bb >> 23) % 96u) << 23));
        unsigned pick = xs32(s);
        if ((pick & 127u) == 0) a = 0.0f;
        if ((pick & 127u) == 1) b = __uint_as_float(__float_as_uint(b) | 0x7FFFFFu);  // all-ones significand
        if ((pick & 127u) == 2) a = __uint_as_float(__float_as_uint(b) + ((pick >> 8) & 3u));  // a ~ b
        // the epipolar line's quotients (SDM_K1_OPT bit 15) rely on a wider operand range, zero numerators and all-ones
        // divisor significands: operands that are zero or in [2^-59, 2^38] (what line_quot_safe guarantees), b != 0
        {
            float a2 = __uint_as_float((ba & 0x807FFFFFu) | ((68u + (ba >> 23) % 97u) << 23));  // 2^-59 .. 2^37
            float b2 = __uint_as_float((bb & 0x807FFFFFu) | ((68u + (bb >> 23) % 97u) << 23));
            if ((pick & 127u) == 0) a2 = 0.0f;
            if ((pick & 127u) == 1 || (pick & 127u) == 3) b2 = __uint_as_float(__float_as_uint(b2) | 0x7FFFFFu);
            if ((pick & 127u) == 2) a2 = __uint_as_float(__float_as_uint(b2) + ((pick >> 8) & 3u));
            const float q2 = quot_fast(a2, b2, rcp_fast(b2)), e2 = a2 / b2;
            if (__float_as_uint(q2) != __float_as_uint(e2)) cnt++;
            n++;
        }
        if (!quot_window_ok(a, b)) continue;
        float q = quot_fast(a, b, rcp_fast(b));
        float e = a / b;
        if (!(q == e || (q != q && e != e))) cnt++;
        n++;
    }
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_down(cnt, o);
        n += __shfl_down(n, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (cnt) atomicAdd(bad, cnt);
        atomicAdd(tested, n);
    }
}

// which = 3: closed-form angle gates vs the reference statement, for d = theta2 - ref over
// (a) every float in [-400, 400] visited with a stride, (b) every float within 64 ulps of each
// boundary (+-45, +-80, +-100, +-260, +-280, +-315, 0, +-180, +-360), (c) random theta pairs.
// With prm: also the run-time-constant forms (gate2_fails_k / gate3_fails_k) against the reference statement under
// prm.lambdaL / prm.lambdaTheta -- how sdm_set_params decides whether thresholds other than the defaults may use them.
__global__ __launch_bounds__(BLOCK) void k_selftest_gates(DevParams prm, unsigned long long* __restrict__ bad,
                                                          unsigned long long* __restrict__ tested)
{
    const unsigned gid = blockIdx.x * BLOCK + threadIdx.x, nthreads = gridDim.x * BLOCK;
    unsigned long long cnt = 0, n = 0;
    auto check = [&](float d) {
        if (!(d < 360.0f)) return;  // the kernel's guard routes these to the reference form
        if (prm.default_gates) {
            if (gate2_fails_fast(d) != gate2_fails_ref(d, 80.0f)) cnt++;
            if (gate3_fails_fast(d) != gate3_fails_ref(d, 45.0f)) cnt++;
            if (gate2_fails_fast1(d) != gate2_fails_ref(d, 80.0f)) cnt++;  // the one-comparison forms the scan uses
            if (gate3_fails_fast1(d) != gate3_fails_ref(d, 45.0f)) cnt++;
        }
        if (gate2_fails_k(d, prm.g2c) != gate2_fails_ref(d, prm.lambdaL)) cnt++;
        if (gate3_fails_k(d, prm.g3lo, prm.g3span) != gate3_fails_ref(d, prm.lambdaTheta)) cnt++;
        n++;
    };
    // (a) strided sweep of the positive and negative float line up to |d| = 400 (0x43C80000)
    for (unsigned u = gid; u <= 0x43C80000u; u += nthreads) {
        check(__uint_as_float(u));
        check(__uint_as_float(u | 0x80000000u));
    }
    // (b) boundary neighbourhoods (the sweep above visits every float; these are the named places once more)
    const float bnd[12] = {0.f, 45.f, 80.f, 90.f, 100.f, 180.f, 260.f, 270.f, 280.f, 315.f, 360.f, 135.f};
    if (gid < 12 * 4096) {
        unsigned base = __float_as_uint(bnd[gid / 4096]);
        int k = (int)(gid % 4096) - 2048;
        float v = __uint_as_float(base + k);
        check(v);
        check(-v);
    }
    // (c) differences of in-range thetas
    unsigned s = 0xC2B2AE35u * (gid + 1);
    for (int i = 0; i < 64; i++) {
        float t2 = ((xs32(s) >> 8) * (1.0f / 16777216.0f)) * 360.0f;
        float rf = ((xs32(s) >> 8) * (1.0f / 16777216.0f)) * 360.0f;
        check(t2 - rf);
    }
    check(-__builtin_inff());
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_down(cnt, o);
        n += __shfl_down(n, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (cnt) atomicAdd(bad, cnt);
        atomicAdd(tested, n);
    }
}

}  // namespace sdm
