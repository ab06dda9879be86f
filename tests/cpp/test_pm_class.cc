// Drives the ProbabilityMapping C++ class (include/sdm/ProbabilityMapping.h) the way the reference's
// mapping thread would (PM.cc:65-135): SemiDenseRecon per keyframe in insertion order, which also
// triggers InterKeyFrameDepthChecking / UpdateSemiDensePointSet once a keyframe's neighbours are
// all reconstructed.  Input and output are flat binary blobs exchanged with tests/test_gpu_cpp_class.py,
// which repeats the same schedule on the CPU oracle.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "sdm/ProbabilityMapping.h"

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
    if (const char* cap = getenv("SDM_TEST_MAX_KF"))  // fewer device slots than keyframes: exercises the LRU eviction,
        if (atoi(cap) > 0) opt.max_keyframes = atoi(cap);  // re-upload and PushDepth paths
    ProbabilityMapping pm(&map, opt);

    for (int k = 0; k < n_kf; k++) pm.SemiDenseRecon(&kfs[k]);
    if (!pm.ok()) return 3;

    // the remaining public surface, on keyframes 1 and 2
    float mn = 0, mx = 0;
    pm.StereoSearchConstraints(&kfs[1], &mn, &mx);
    float F[9];
    pm.ComputeFundamental(&kfs[1], &kfs[2], F);
    float umin = 0, umax = 0;
    pm.GetSearchRange(umin, umax, W / 2, H / 2, mn, mx, &kfs[1], &kfs[2]);
    std::vector<float> px_out;
    for (int y = 2; y < H - 2; y += 5)
        for (int x = 2; x < W - 2; x += 7) {
            ProbabilityMapping::depthHo dh;
            float bu = 0, bv = 0;
            pm.EpipolarSearch(&kfs[1], &kfs[2], x, y, (float)kfs[1].im_.at(y, x), mn, mx, &dh, F, bu, bv,
                              kfs[1].GradTheta.at(y, x), 0.0f);
            px_out.push_back(dh.depth);
            px_out.push_back(dh.sigma);
            px_out.push_back(dh.supported ? 1.f : 0.f);
            px_out.push_back(bu);
        }
    {   // arguments that are not the resident keyframes' values are reported (once here), not silently ignored
        ProbabilityMapping::depthHo dh;
        float bu = 0, bv = 0;
        pm.EpipolarSearch(&kfs[1], &kfs[2], 9, 7, (float)kfs[1].im_.at(7, 9) + 1.0f, mn, mx, &dh, F, bu, bv,
                          kfs[1].GradTheta.at(7, 9), 0.0f);
    }
    std::vector<ProbabilityMapping::depthHo> hs(5);
    for (int i = 0; i < 5; i++) {
        hs[i].depth = 1.0f + 0.01f * i;
        hs[i].sigma = 0.05f;
    }
    ProbabilityMapping::depthHo fused;
    pm.InverseDepthHypothesisFusion(hs, fused);
    // stand-alone map operations on a copy of keyframe 3's maps
    sdm::Mat<float> dm = kfs[3].depth_map_.clone(), ds = kfs[3].depth_sigma_.clone();
    pm.IntraKeyFrameDepthChecking(dm, ds, kfs[3].GradImg);
    pm.IntraKeyFrameDepthGrowing(dm, ds, kfs[3].GradImg);
    // pose change after BA: kf 4 takes kf 5's pose, then UpdateAllSemiDensePointSet (needs >= 10 KFs)
    memcpy(kfs[4].Tcw, kfs[5].Tcw, sizeof(float) * 12);
    kfs[4].poseChanged = true;
    pm.UpdateAllSemiDensePointSet();
    long nv = pm.SavePointCloudObj(argc > 3 ? argv[3] : "/dev/null");
    if (argc > 4) {  // CARV transcript entries of keyframes 2 and 3
        std::ofstream tr(argv[4]);
        pm.AppendTranscriptEntry(&kfs[2], 7, 2, tr);
        pm.AppendTranscriptEntry(&kfs[3], 8, 3, tr, 0.25);  // a looser sigma filter so that points are emitted
    }

    FILE* o = fopen(argv[2], "wb");
    if (!o) return 2;
    for (int k = 0; k < n_kf; k++) {
        int flags[3] = {kfs[k].semidense_flag_, kfs[k].interKF_depth_flag_, kfs[k].poseChanged};
        fwrite(flags, sizeof(int), 3, o);
        fwrite(kfs[k].depth_map_.ptr(), sizeof(float), (size_t)W * H, o);
        fwrite(kfs[k].depth_sigma_.ptr(), sizeof(float), (size_t)W * H, o);
        fwrite(kfs[k].SemiDensePointSets_.ptr(), sizeof(float), (size_t)3 * W * H, o);
        fwrite(kfs[k].GradImg.ptr(), sizeof(float), (size_t)W * H, o);
    }
    float misc[16] = {mn, mx, umin, umax, fused.depth, fused.sigma, fused.supported ? 1.f : 0.f, (float)nv};
    fwrite(misc, sizeof(float), 8, o);
    fwrite(F, sizeof(float), 9, o);
    int npx = (int)px_out.size();
    fwrite(&npx, sizeof(int), 1, o);
    fwrite(px_out.data(), sizeof(float), px_out.size(), o);
    fwrite(dm.ptr(), sizeof(float), (size_t)W * H, o);
    fwrite(ds.ptr(), sizeof(float), (size_t)W * H, o);
    fclose(o);
    return 0;
}
