// Stand-alone reproducer for the launch-size limit DESIGN.md §7 reports: does a 1-D dispatch of >= 2^31 work-items
// (below HIP's documented 2^32) execute every workgroup exactly once, with the right blockIdx?
// Every thread adds 1 to hits[blockIdx.x]; thread 0 also stores blockIdx.x in id[blockIdx.x] and every thread compares
// the OpenCL-style global id with blockIdx.x * 256 + threadIdx.x.  A correct dispatch leaves hits[b] == 256 and
// id[b] == b for every b < blocks, untouched guard entries behind them, and no global-id mismatch.
// Second question (what the engine actually tripped over, tools/debug/unsliced_vs_sliced.py): is the NEXT kernel on the same
// stream held back until such a dispatch has finished?  k_check is launched right behind k_mark, without a host
// synchronisation, and counts the workgroups whose hits it finds incomplete.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/big_grid.hip -o orb-slam-free-space-carving_amd/lib/ubench_big_grid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" __device__ size_t __ockl_get_global_id(unsigned);

__global__ __launch_bounds__(256) void k_mark(unsigned* __restrict__ hits, unsigned* __restrict__ id,
                                              unsigned long long* __restrict__ gid_bad)
{
    const unsigned b = blockIdx.x;
    atomicAdd(&hits[b], 1u);
    if (threadIdx.x == 0) id[b] = b;
    const unsigned long long want = (unsigned long long)b * 256ull + threadIdx.x;
    if ((unsigned long long)__ockl_get_global_id(0) != want) atomicAdd(gid_bad, 1ull);
}

// stream order: runs behind k_mark on the same stream; every workgroup of k_mark must be complete by now
__global__ __launch_bounds__(256) void k_check(const unsigned* __restrict__ hits, unsigned nblocks,
                                               unsigned long long* __restrict__ early)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    if (i < nblocks && __builtin_nontemporal_load(&hits[i]) != 256u) atomicAdd(early, 1ull);
}

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));                                   \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

// Third question: the engine's K1 (256 threads, dynamic LDS, a barrier, some workgroups returning at once, ~10 us per
// workgroup, 8 waves per SIMD) executed only part of a 12 959 744-workgroup grid (tools/debug/unsliced_bisect.py).  Which of
// those properties does it take?  MODE bits: 1 dynamic LDS + barrier, 2 a third of the workgroups return at once,
// 4 ~`spin` dependent FMAs per thread (a long-running dispatch).
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_mark_like_k1(unsigned* __restrict__ hits,
                                                                                                int spin, float seed)
{
    extern __shared__ float lds[];
    const unsigned b = blockIdx.x;
    if ((MODE & 2) && (b % 3u) == 2u) {
        if (threadIdx.x == 0) hits[b] = 256u;  // counted as done
        return;
    }
    float a = seed + (float)threadIdx.x;
    if (MODE & 4)
        for (int i = 0; i < spin; i++) a = __builtin_fmaf(a, 1.0000001f, 0.5f);
    if (MODE & 1) {
        lds[threadIdx.x] = a;
        __syncthreads();
        a += lds[(threadIdx.x + 64) & 255];
    }
    if (a == 12345.678f) hits[b] = 0u;  // keeps `a` alive
    atomicAdd(&hits[b], 1u);
}

template <int MODE>
int run_like_k1(unsigned* hits, unsigned long long* early, unsigned blocks, int spin, hipStream_t st)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipMemsetAsync(hits, 0, sizeof(unsigned) * (size_t)blocks, st));
    CK(hipMemsetAsync(early, 0, sizeof(unsigned long long), st));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_mark_like_k1<MODE>, dim3(blocks), dim3(256), (MODE & 1) ? 8448 : 0, st, hits, spin, 1.0f);
    CK(hipGetLastError());
    CK(hipEventRecord(e1, st));
    hipLaunchKernelGGL(k_check, dim3((blocks + 255) / 256), dim3(256), 0, st, hits, blocks, early);
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long e = 0;
    CK(hipMemcpy(&e, early, sizeof(e), hipMemcpyDeviceToHost));
    printf("K1-like dispatch, mode %d, %u workgroups x 256, spin %d, %.1f ms: %llu workgroups not (fully) executed -> %s\n", MODE,
           blocks, spin, ms, e, e == 0 ? "OK" : "MIS-EXECUTED");
    fflush(stdout);
    return 0;
}

int main()
{
    const unsigned long long items[] = {1ull << 30, (1ull << 31) - 256, 1ull << 31, (1ull << 31) + 256,
                                        3800000000ull / 256 * 256, (1ull << 32) - 256};
    const unsigned guard = 4096;
    const unsigned max_blocks = (unsigned)(((1ull << 32) - 256) / 256);
    unsigned *hits, *id;
    unsigned long long* gid_bad;
    CK(hipMalloc(&hits, sizeof(unsigned) * ((size_t)max_blocks + guard)));
    CK(hipMalloc(&id, sizeof(unsigned) * ((size_t)max_blocks + guard)));
    CK(hipMalloc(&gid_bad, sizeof(unsigned long long)));
    int rt = 0, drv = 0;
    (void)hipRuntimeGetVersion(&rt);
    (void)hipDriverGetVersion(&drv);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s (%s), HIP runtime %d, driver %d\n", prop.name, prop.gcnArchName, rt, drv);
    std::vector<unsigned> h, hid;
    for (unsigned long long n : items) {
        const unsigned blocks = (unsigned)(n / 256);
        CK(hipMemset(hits, 0, sizeof(unsigned) * ((size_t)blocks + guard)));
        CK(hipMemset(id, 0xff, sizeof(unsigned) * ((size_t)blocks + guard)));
        CK(hipMemset(gid_bad, 0, sizeof(unsigned long long)));
        hipLaunchKernelGGL(k_mark, dim3(blocks), dim3(256), 0, 0, hits, id, gid_bad);
        hipError_t le = hipGetLastError();
        hipError_t se = hipDeviceSynchronize();
        if (le != hipSuccess || se != hipSuccess) {
            printf("work-items %llu (%u workgroups): launch %s, sync %s\n", n, blocks, hipGetErrorString(le),
                   hipGetErrorString(se));
            continue;
        }
        h.resize((size_t)blocks + guard);
        hid.resize((size_t)blocks + guard);
        CK(hipMemcpy(h.data(), hits, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hid.data(), id, sizeof(unsigned) * hid.size(), hipMemcpyDeviceToHost));
        unsigned long long gb = 0;
        CK(hipMemcpy(&gb, gid_bad, sizeof(gb), hipMemcpyDeviceToHost));
        unsigned long long bad_hits = 0, bad_id = 0, bad_guard = 0, total = 0;
        long long first = -1;
        for (size_t b = 0; b < blocks; b++) {
            total += h[b];
            if (h[b] != 256u) {
                bad_hits++;
                if (first < 0) first = (long long)b;
            }
            if (hid[b] != (unsigned)b) bad_id++;
        }
        for (size_t b = blocks; b < (size_t)blocks + guard; b++)
            if (h[b] != 0u || hid[b] != 0xffffffffu) bad_guard++;
        printf("work-items %llu (%u workgroups x 256): executed work-items %llu, workgroups with hits != 256: %llu (first %lld), "
               "wrong id: %llu, guard entries touched: %llu, global-id mismatches: %llu -> %s\n",
               n, blocks, total, bad_hits, first, bad_id, bad_guard, gb,
               (bad_hits | bad_id | bad_guard | gb) == 0 && total == n ? "OK" : "MIS-EXECUTED");
        fflush(stdout);
    }
    // ---- stream order behind a big dispatch
    unsigned long long* early;
    CK(hipMalloc(&early, sizeof(unsigned long long)));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (unsigned long long n : items) {
        const unsigned blocks = (unsigned)(n / 256);
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemsetAsync(hits, 0, sizeof(unsigned) * ((size_t)blocks + guard), st));
            CK(hipMemsetAsync(early, 0, sizeof(unsigned long long), st));
            hipLaunchKernelGGL(k_mark, dim3(blocks), dim3(256), 0, st, hits, id, gid_bad);
            hipLaunchKernelGGL(k_check, dim3((blocks + 255) / 256), dim3(256), 0, st, hits, blocks, early);
            CK(hipStreamSynchronize(st));
            unsigned long long e = 0;
            CK(hipMemcpy(&e, early, sizeof(e), hipMemcpyDeviceToHost));
            printf("stream order, work-items %llu, run %d: the next kernel on the stream saw %llu of %u workgroups unfinished -> %s\n",
                   n, rep, e, blocks, e == 0 ? "OK" : "STARTED EARLY");
            fflush(stdout);
        }
    }
    // ---- what the engine really launched (tools/debug/unsliced_holes.py): 2048 keyframes x 9880 workgroups = 20 234 240
    // workgroups x 256 = 5.18e9 work-items, ABOVE HIP's 2^32 limit.  Is that refused?
    {
        const unsigned blocks = 20234240u;
        unsigned *big, *big_id;  // sized for the whole grid, in case it does run in full
        CK(hipMalloc(&big, sizeof(unsigned) * (size_t)blocks));
        CK(hipMalloc(&big_id, sizeof(unsigned) * (size_t)blocks));
        CK(hipMemset(big, 0, sizeof(unsigned) * (size_t)blocks));
        CK(hipMemset(gid_bad, 0, sizeof(unsigned long long)));
        (void)hipGetLastError();
        hipLaunchKernelGGL(k_mark, dim3(blocks), dim3(256), 0, st, big, big_id, gid_bad);
        const hipError_t le = hipGetLastError();
        const hipError_t se = hipStreamSynchronize(st);
        std::vector<unsigned> hb(blocks);
        CK(hipMemcpy(hb.data(), big, sizeof(unsigned) * (size_t)blocks, hipMemcpyDeviceToHost));
        unsigned long long done = 0, last = 0;
        for (size_t b = 0; b < blocks; b++)
            if (hb[b] == 256u) {
                done++;
                last = b;
            }
        printf("grid of %u workgroups x 256 = %llu work-items (> 2^32): launch returned \"%s\", sync \"%s\"; workgroups executed: %llu "
               "(highest index %llu); 2^32-wrapped grid would be %llu workgroups\n",
               blocks, (unsigned long long)blocks * 256ull, hipGetErrorString(le), hipGetErrorString(se), done, last,
               ((unsigned long long)blocks * 256ull - (1ull << 32)) / 256ull);
        fflush(stdout);
        CK(hipFree(big));
        CK(hipFree(big_id));
    }
    // ---- the engine's launch shape
    for (unsigned blocks : {9719808u, 12959744u, 14843750u}) {
        if (run_like_k1<0>(hits, early, blocks, 0, st)) return 1;
        if (run_like_k1<1>(hits, early, blocks, 0, st)) return 1;
        if (run_like_k1<3>(hits, early, blocks, 0, st)) return 1;
        if (run_like_k1<4>(hits, early, blocks, 400, st)) return 1;
        if (run_like_k1<7>(hits, early, blocks, 400, st)) return 1;
        if (run_like_k1<7>(hits, early, blocks, 2000, st)) return 1;
    }
    return 0;
}
