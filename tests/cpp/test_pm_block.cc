// Drives ProbabilityMapping::SemiDenseReconBlock -- the batch / multi-GPU form of the class -- on
// one rank: the whole sequence is this rank's block, the exchange calls are no-ops (world size 1), and the result must
// be the snapshot-order pipeline of the CPU oracle (tests/test_gpu_cpp_class.py).  Also exercises the slot-cache
// contract: InvalidateDepth (host map edited -> re-uploaded) and Forget (slot dropped -> keyframe re-uploaded).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <unistd.h>

#include "sdm/ProbabilityMapping.h"
#include "sdm_c.h"

static void rd(FILE* f, void* p, size_t n)
{
    if (fread(p, 1, n, f) != n) {
        fprintf(stderr, "short read\n");
        exit(2);
    }
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[4];
    rd(f, hdr, sizeof(hdr));
    const int W = hdr[0], H = hdr[1], n_kf = hdr[2], covisN = hdr[3];
    std::vector<sdm::KeyFrame> kfs(n_kf);
    std::vector<std::vector<int>> covis(n_kf);
    for (int k = 0; k < n_kf; k++) {
        sdm::KeyFrame& kf = kfs[k];
        kf.mnId = k;
        kf.im_ = sdm::Mat<uint8_t>(H, W);
        rd(f, kf.im_.ptr(), (size_t)W * H);
        float K[4];
        rd(f, K, sizeof(K));
        kf.fx = K[0];
        kf.fy = K[1];
        kf.cx = K[2];
        kf.cy = K[3];
        rd(f, kf.Tcw, sizeof(float) * 12);
        int nc;
        rd(f, &nc, sizeof(int));
        covis[k].resize(nc);
        rd(f, covis[k].data(), sizeof(int) * nc);
        int nd;
        rd(f, &nd, sizeof(int));
        kf.point_depths.resize(nd);
        rd(f, kf.point_depths.data(), sizeof(float) * nd);
    }
    fclose(f);
    sdm::Map map;
    for (int k = 0; k < n_kf; k++) {
        for (int j : covis[k]) kfs[k].covisible.push_back(&kfs[j]);
        map.keyframes.push_back(&kfs[k]);
    }
    sdm::Options opt;
    opt.covisN = covisN;
    opt.max_keyframes = n_kf;
    ProbabilityMapping pm(&map, opt);
    pm.SemiDenseReconBlock(map.keyframes, 0, n_kf);  // sizes the context, runs K1-K5 for the block
    if (!pm.ok()) return 3;
    // world size 1: accepted, nothing to build -- unless SDM_COMM_SINGLE_RANK_RCCL=1 asks for a real one-rank RCCL
    // communicator (the hardware rehearsal): then the second pass runs its go / no-go all-reduce and an empty send/recv
    // group through RCCL, from the C++ class
    unsigned char comm_id[SDM_COMM_ID_BYTES];
    const unsigned char* idp = nullptr;
    const char* rehearse = getenv("SDM_COMM_SINGLE_RANK_RCCL");
    if (rehearse && atoi(rehearse) == 1) {
        if (sdm_comm_unique_id(comm_id) != SDM_OK) return 6;  // no RCCL here
        idp = comm_id;
    }
    if (!pm.InitSharding(idp, 1, 0)) return idp ? 6 : 4;
    // after pass 1 the device maps are still the reconstructed ones: compact sources
    if (!pm.CompactSourcesReady(map.keyframes)) return 7;
    pm.SemiDenseReconBlock(map.keyframes, 0, n_kf);  // second pass: everything already reconstructed -> no work, no change
    // ... and the second pass restored the (checked) host maps into the slots WITHOUT disqualifying them: a sharded second
    // pass sends such maps, and a compact send of a map that is not a pipeline map fails on the sender alone, after its
    // peers have posted their receives (round-3 advice)
    if (!pm.CompactSourcesReady(map.keyframes)) return 8;

    // slot-cache contract
    std::vector<float> xyz1 = kfs[1].SemiDensePointSets_.data, xyz2 = kfs[2].SemiDensePointSets_.data;
    pm.InvalidateDepth(&kfs[1]);
    pm.UpdateSemiDensePointSet(&kfs[1]);  // from the host (checked) map, uploaded again
    pm.Forget(&kfs[2]);
    pm.UpdateSemiDensePointSet(&kfs[2]);  // keyframe uploaded again into a fresh slot
    int same = (xyz1 == kfs[1].SemiDensePointSets_.data) && (xyz2 == kfs[2].SemiDensePointSets_.data);

    // ---- the mapping thread (PM.cc:65-135): a second mapper polls a map that fills while it runs -----------------
    std::vector<sdm::KeyFrame> kt(n_kf);
    sdm::Map tmap;
    for (int k = 0; k < n_kf; k++) {
        kt[k].mnId = k;
        kt[k].im_ = kfs[k].im_;
        kt[k].fx = kfs[k].fx; kt[k].fy = kfs[k].fy; kt[k].cx = kfs[k].cx; kt[k].cy = kfs[k].cy;
        memcpy(kt[k].Tcw, kfs[k].Tcw, sizeof(float) * 12);
        kt[k].point_depths = kfs[k].point_depths;
        for (int j : covis[k]) kt[k].covisible.push_back(&kt[j]);
        tmap.AddKeyFrame(&kt[k]);
    }
    sdm::Options topt = opt;
    topt.obj_path = argc > 3 ? argv[3] : "";
    topt.poll_us = 1000;
    ProbabilityMapping pt(&tmap, topt);
    std::thread th(&ProbabilityMapping::Run, &pt);
    // the thread owns the keyframes while it runs (the reference reads KeyFrame flags across threads without
    // synchronisation; this test does not): wait for two full passes (a mutex-guarded counter), then ask it to stop.
    // Every pass after the first finds nothing left to do, so the result does not depend on the timing.
    for (int waited = 0; waited < 30000 && pt.Passes() < 2; waited++) usleep(1000);  // two full passes, <= 30 s
    if (pt.Passes() < 2) return 5;
    pt.RequestFinish();
    th.join();
    int thread_ok = pt.isFinished() ? 1 : 0;

    FILE* o = fopen(argv[2], "wb");
    if (!o) return 2;
    for (int k = 0; k < n_kf; k++) {
        int flags[3] = {kt[k].semidense_flag_ && thread_ok, kt[k].interKF_depth_flag_, 0};
        fwrite(flags, sizeof(int), 3, o);
        fwrite(kt[k].depth_map_.ptr(), sizeof(float), (size_t)W * H, o);
        fwrite(kt[k].depth_sigma_.ptr(), sizeof(float), (size_t)W * H, o);
    }
    for (int k = 0; k < n_kf; k++) {
        int flags[3] = {kfs[k].semidense_flag_, kfs[k].interKF_depth_flag_, same};
        fwrite(flags, sizeof(int), 3, o);
        fwrite(kfs[k].depth_map_.ptr(), sizeof(float), (size_t)W * H, o);
        fwrite(kfs[k].depth_sigma_.ptr(), sizeof(float), (size_t)W * H, o);
        fwrite(kfs[k].SemiDensePointSets_.ptr(), sizeof(float), (size_t)3 * W * H, o);
    }
    fclose(o);
    return 0;
}
