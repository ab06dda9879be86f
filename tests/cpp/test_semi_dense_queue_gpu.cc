// The Modeler seam END TO END on the GPU: adapters/semi_dense_queue.h instantiated with the real
// ProbabilityMapping (SemiDenseQueue = SemiDenseQueueT<ProbabilityMapping>) over test doubles of the fork's KeyFrame /
// MapPoint / cv::Mat (tests/cpp/mock_fork; the real headers and OpenCV are absent from the image).  Keyframes are
// enqueued in creation order like Modeler::AddKeyFrameEntry would (src/Modeler/Modeler.cc:1321-1353, 1465-1472), the
// Modeler thread's idle branch is played by a ProcessOne() loop (Modeler.cc:63-66, 100-128), and the Injector records
// what would go into the CARV transcript (SFMTranscriptInterface_ORBSLAM.cpp:319-374).  tests/test_gpu_queue.py replays
// the same schedule on the CPU oracle and compares every injected point bit for bit.
//
// usage: test_semi_dense_queue_gpu IN.bin OUT.bin [MAX_SIGMA]   (IN: the blob of tests/test_gpu_cpp_class.py::write_blob;
//        MAX_SIGMA: the injection filter, default the obj writer's 0.01, PM.cc:120)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sdm/ProbabilityMapping.h"
#include "semi_dense_queue.h"

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
    std::vector<ORB_SLAM2::KeyFrame> kfs(n_kf);
    std::vector<std::vector<unsigned char> > images(n_kf);
    std::vector<std::vector<int> > covis(n_kf);
    std::vector<std::vector<ORB_SLAM2::MapPoint> > points(n_kf);
    for (int k = 0; k < n_kf; k++) {
        ORB_SLAM2::KeyFrame& kf = kfs[k];
        kf.mnId = k;
        kf.mnFrameId = 1000 + k;
        images[k].resize((size_t)W * H);
        rd(f, images[k].data(), (size_t)W * H);
        float K[4], T[12];
        rd(f, K, sizeof(K));
        kf.fx = K[0];
        kf.fy = K[1];
        kf.cx = K[2];
        kf.cy = K[3];
        rd(f, T, sizeof(T));
        kf.Tcw = cv::Mat(4, 4, CV_32F);
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) kf.Tcw.at<float>(r, c) = r < 3 ? T[r * 4 + c] : (c == 3 ? 1.f : 0.f);
        int nc;
        rd(f, &nc, sizeof(int));
        covis[k].resize(nc);
        rd(f, covis[k].data(), sizeof(int) * nc);
        int nd;
        rd(f, &nd, sizeof(int));
        std::vector<float> depths(nd);
        rd(f, depths.data(), sizeof(float) * nd);
        // map points of this keyframe only (no shared points -> in-plane rotation 0, PM.cc:174-177), placed on the optical
        // axis at the given camera depths: Pw = Rwc * (0,0,d) + Ow
        points[k].resize(nd);
        for (int i = 0; i < nd; i++) {
            ORB_SLAM2::MapPoint& mp = points[k][i];
            mp.mnId = 100000ul * (k + 1) + i;
            mp.pos = cv::Mat(3, 1, CV_32F);
            for (int r = 0; r < 3; r++) {
                double ow = 0;
                for (int q = 0; q < 3; q++) ow -= (double)T[q * 4 + r] * T[q * 4 + 3];
                mp.pos.at<float>(r, 0) = (float)((double)T[2 * 4 + r] * depths[i] + ow);
            }
        }
    }
    fclose(f);
    for (int k = 0; k < n_kf; k++) {
        for (int j : covis[k]) kfs[k].cov.push_back(&kfs[j]);
        for (size_t i = 0; i < points[k].size(); i++) kfs[k].mps.push_back(&points[k][i]);
        kfs[k].mvKeysUn.resize(points[k].size());  // angle -1: no orientation
    }

    sdm::Map map;
    sdm::Options opt;
    opt.covisN = covisN;
    opt.max_keyframes = n_kf;
    opt.obj_path = "";
    ProbabilityMapping pm(&map, opt);
    FILE* o = fopen(argv[2], "wb");
    if (!o) return 2;
    int n_lookups = 0, pin_errors = 0;
    auto image = [&](ORB_SLAM2::KeyFrame* k, cv::Mat& gray) {  // Modeler.cc:143-155: the stored frame, as gray
        n_lookups++;
        gray = cv::Mat(H, W, CV_8UC1);
        memcpy(gray.data, images[k->mnId].data(), (size_t)W * H);
        return true;
    };
    auto inject = [&](ORB_SLAM2::KeyFrame* k, std::vector<cv::Point3f>& pts) {  // would be addKeyFrameInsertionWithLinesEntry
        if (k->not_erase < 1) pin_errors++;
        const int rec[2] = {(int)k->mnId, (int)pts.size()};
        fwrite(rec, sizeof(int), 2, o);
        for (size_t i = 0; i < pts.size(); i++) {
            const float p[3] = {pts[i].x, pts[i].y, pts[i].z};
            fwrite(p, sizeof(float), 3, o);
        }
    };
    const double max_sigma = argc > 3 ? atof(argv[3]) : 0.01;
    sdm_adapter::SemiDenseQueue q(&pm, &map, image, inject, /*max_queue=*/(size_t)n_kf, max_sigma);
    for (int k = 0; k < n_kf; k++) q.Enqueue(&kfs[k]);  // LocalMapping thread, creation order
    int processed = 0;
    while (q.ProcessOne()) processed++;                  // Modeler thread's idle branch
    // bundle adjustment moves a finished keyframe: the next drain re-projects it (PM.cc:321-334); its points are not
    // injected twice
    const int end_marker[2] = {-1, 0};
    fwrite(end_marker, sizeof(int), 2, o);
    // the depth priors the adapter derived from the map points (for the oracle's StereoSearchConstraints)
    for (int k = 0; k < n_kf; k++) {
        sdm::KeyFrame* s = q.Find(&kfs[k]);
        const int nd = s ? (int)s->point_depths.size() : 0;
        fwrite(&nd, sizeof(int), 1, o);
        if (nd) fwrite(s->point_depths.data(), sizeof(float), nd, o);
        const int flags[2] = {s ? (int)s->semidense_flag_ : 0, s ? (int)s->interKF_depth_flag_ : 0};
        fwrite(flags, sizeof(int), 2, o);
    }
    int unpinned = 0;
    for (int k = 0; k < n_kf; k++) unpinned += (kfs[k].not_erase == 0 && kfs[k].pins == kfs[k].unpins) ? 1 : 0;
    const int tail[5] = {processed, n_lookups, pin_errors, unpinned, (int)map.GetAllKeyFrames().size()};
    fwrite(tail, sizeof(int), 5, o);
    fclose(o);
    return pm.ok() ? 0 : 3;
}
