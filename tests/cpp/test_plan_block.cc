// Prints ProbabilityMapping::PlanBlock (the C++ class's sharding plan: who reconstructs what first, which maps leave and
// arrive) for every rank of a sequence with index-local covisibility, one line per rank; tests/test_adapter.py compares it
// with shard.plan (the Python plan the bench and the multi-rank tests use).  Host logic only: no device call is made.
#include <cstdio>
#include <cstdlib>
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
    if (argc > 4) kfs[atoi(argv[4])].bad = true;              // a bad keyframe is skipped as a reference and as a neighbour
    if (argc > 5) kfs[atoi(argv[5])].semidense_flag_ = true;  // an already reconstructed one is not a reference again
    const int count = n_all / world;
    for (int r = 0; r < world; r++) {
        sdm::BlockPlan pl;
        if (!ProbabilityMapping::PlanBlock(all, r * count, count, world, r, n, &pl)) return 3;
        std::printf("rank %d", r);
        dump("refs", pl.refs);
        std::vector<int> needed, boundary, flat;
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
