// Prints ProbabilityMapping::PlanBlock (the C++ class's sharding plan: who reconstructs what first, which maps leave and
// arrive) for every rank of a sequence with index-local covisibility, one line per rank; tests/test_adapter.py compares it
// with shard.plan (the Python plan the bench and the multi-rank tests use).  Host logic only: no device call is made.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "sdm/ProbabilityMapping.h"

static void dump(const char* name, const std::vector<int>& v)
{
    std::printf(" %s", name);
    for (int x : v) std::printf(" %d", x);
    std::printf(" ;");
}

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    const int n_all = atoi(argv[1]), world = atoi(argv[2]), n = atoi(argv[3]);
    std::vector<sdm::KeyFrame> kfs(n_all);
    std::vector<sdm::KeyFrame*> all;
    for (int k = 0; k < n_all; k++) {
        kfs[k].mnId = k;
        // covisibility order of the synthetic sequences: |dk| ascending, +dk before -dk (synth.Scene.neighbours)
        for (int d = 1; d < n_all; d++) {
            if (k + d < n_all) kfs[k].covisible.push_back(&kfs[k + d]);
            if (k - d >= 0) kfs[k].covisible.push_back(&kfs[k - d]);
        }
        all.push_back(&kfs[k]);
    }
    const bool twopass = argc > 4 && std::string(argv[4]) == "twopass";
    if (!twopass) {
        if (argc > 4) kfs[atoi(argv[4])].bad = true;              // a bad keyframe is skipped as a reference and as a neighbour
        if (argc > 5) kfs[atoi(argv[5])].semidense_flag_ = true;  // an already reconstructed one is not a reference again
    }
    const int count = n_all / world;
    if (twopass) {
        // Pass 1 with some keyframes not yet mapped (LocalMapping still busy with them), then pass 2 after they became
        // usable.  Between the passes every rank applies what SemiDenseReconBlock applies: the stage flags of every
        // keyframe ANY rank reconstructed / checked (BlockPlan::recon_all / check_all) -- the state the next plan is
        // derived from is replicated, so the pass-2 lists printed below must match pairwise (with flags
        // set only on a rank's own keyframes the second pass deadlocked).
        for (int k = 0; k < n_all; k++) kfs[k].mapped = (k % 5 != 3);
        sdm::BlockPlan p1;
        if (!ProbabilityMapping::PlanBlock(all, 0, count, world, 0, n, &p1)) return 3;
        for (int r = 1; r < world; r++) {  // every rank derives the same replicated part
            sdm::BlockPlan pr;
            if (!ProbabilityMapping::PlanBlock(all, r * count, count, world, r, n, &pr)) return 3;
            if (pr.recon_all != p1.recon_all || pr.check_all != p1.check_all) return 6;
        }
        for (int k = 0; k < n_all; k++) {
            if (p1.recon_all[k]) kfs[k].semidense_flag_ = true;
            if (p1.check_all[k]) kfs[k].interKF_depth_flag_ = true;
            kfs[k].mapped = true;
        }
    }
    for (int r = 0; r < world; r++) {
        sdm::BlockPlan pl;
        if (!ProbabilityMapping::PlanBlock(all, r * count, count, world, r, n, &pl)) return 3;
        std::printf("rank %d", r);
        dump("refs", pl.refs);
        dump("check", pl.check);
        std::vector<int> needed, boundary, flat, cflat;
        for (const std::vector<int>& row : pl.check_nbrs) cflat.insert(cflat.end(), row.begin(), row.end());
        dump("check_nbrs", cflat);
        for (int i = 0; i < n_all; i++) {
            if (pl.needed[i]) needed.push_back(i);
            if (pl.boundary[i]) boundary.push_back(i);
        }
        for (const std::vector<int>& row : pl.nbrs) flat.insert(flat.end(), row.begin(), row.end());
        dump("nbrs", flat);
        dump("needed", needed);
        dump("boundary", boundary);
        dump("send_peer", pl.send_peer);
        dump("send_kf", pl.send_kf);
        dump("recv_peer", pl.recv_peer);
        dump("recv_kf", pl.recv_kf);
        std::printf("\n");
    }
    // malformed requests are refused
    sdm::BlockPlan pl;
    if (ProbabilityMapping::PlanBlock(all, 1, count, world, 0, n, &pl) && world > 1) return 4;  // block not at rank*count
    if (ProbabilityMapping::PlanBlock(all, 0, n_all + 1, 1, 0, n, &pl)) return 5;
    return 0;
}
