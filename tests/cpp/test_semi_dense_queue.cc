// Unit test of adapters/semi_dense_queue.h (the Modeler seam) against test doubles of the fork's
// KeyFrame / cv::Mat (tests/cpp/mock_fork) and a recording stand-in for the mapper: queue bound and order, bad keyframes,
// pin / unpin around the work, the image lookup, registration in the map view, mutual covisibility, injection exactly
// once per finished keyframe with the obj writer's filter, erase and pose-update hooks.  CPU only.
#include <cstdio>
#include <string>

#include "sdm/ProbabilityMapping.h"
#include "semi_dense_queue.h"

#define CHECK(c)                                                    \
    do {                                                            \
        if (!(c)) {                                                 \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            return 1;                                               \
        }                                                           \
    } while (0)

// stands in for ProbabilityMapping: "reconstructs" a keyframe at once and finishes (inter-keyframe check + point set)
// the keyframe processed `lag` calls earlier, like the real class does once a keyframe's neighbours are all done
struct FakeMapper {
    std::vector<sdm::KeyFrame*> recon_calls, forgotten;
    int update_all = 0, lag = 1;
    int pinned_during_call = -1;
    ORB_SLAM2::KeyFrame* watch = nullptr;
    void SemiDenseRecon(sdm::KeyFrame* kf)
    {
        if (watch) pinned_during_call = watch->not_erase;
        recon_calls.push_back(kf);
        kf->semidense_flag_ = true;
        const int H = kf->im_.rows, W = kf->im_.cols;
        kf->depth_map_ = sdm::Mat<float>(H, W, 0.f);
        kf->depth_sigma_ = sdm::Mat<float>(H, W, 0.f);
        kf->SemiDensePointSets_ = sdm::Mat<float>(H, 3 * W, 0.f);
        if ((int)recon_calls.size() > lag) {
            sdm::KeyFrame* done = recon_calls[recon_calls.size() - 1 - lag];
            // three pixels: kept, sigma too large, no depth
            done->depth_map_.at(2, 3) = 0.5f;   done->depth_sigma_.at(2, 3) = 0.005f;
            done->depth_map_.at(2, 4) = 0.6f;   done->depth_sigma_.at(2, 4) = 0.5f;
            done->depth_map_.at(3, 3) = 0.0f;   done->depth_sigma_.at(3, 3) = 0.001f;
            done->SemiDensePointSets_.at(2, 9) = 1.f + (float)done->mnId;
            done->SemiDensePointSets_.at(2, 10) = 2.f;
            done->SemiDensePointSets_.at(2, 11) = 3.f;
            done->interKF_depth_flag_ = true;
        }
    }
    void Forget(sdm::KeyFrame* kf) { forgotten.push_back(kf); }
    void UpdateAllSemiDensePointSet() { update_all++; }
};

int main()
{
    const int W = 16, H = 8, N = 5;
    std::vector<ORB_SLAM2::KeyFrame> kfs(N);
    for (int i = 0; i < N; i++) {
        kfs[i].mnId = 10 + i;
        kfs[i].mnFrameId = 100 + i;
        kfs[i].fx = kfs[i].fy = 100.f;
        kfs[i].cx = 8.f;
        kfs[i].cy = 4.f;
        kfs[i].Tcw = cv::Mat(4, 4, CV_32F);
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) kfs[i].Tcw.at<float>(r, c) = (r == c) ? 1.f : 0.f;
        for (int j = 0; j < N; j++)
            if (j != i) kfs[i].cov.push_back(&kfs[j]);  // everybody sees everybody
    }
    kfs[2].bad = true;
    std::map<long unsigned int, int> lookups;
    int missing_frame = 103;  // the Modeler's bounded frame store (Modeler.cc:1506-1508) may have dropped a frame
    auto image = [&](ORB_SLAM2::KeyFrame* k, cv::Mat& gray) {
        lookups[k->mnFrameId]++;
        if ((int)k->mnFrameId == missing_frame) return false;
        gray = cv::Mat(H, W, CV_8UC1);
        for (int i = 0; i < W * H; i++) gray.data[i] = (unsigned char)(i + k->mnId);
        return true;
    };
    std::vector<std::pair<ORB_SLAM2::KeyFrame*, std::vector<cv::Point3f>>> injected;
    auto inject = [&](ORB_SLAM2::KeyFrame* k, std::vector<cv::Point3f>& pts) { injected.push_back(std::make_pair(k, pts)); };

    FakeMapper mapper;
    sdm::Map map;
    sdm_adapter::SemiDenseQueueT<FakeMapper> q(&mapper, &map, image, inject, /*max_queue=*/4);
    CHECK(!q.ProcessOne());  // empty queue: nothing to do (Modeler.cc:106-108)
    for (int i = 0; i < N; i++) q.Enqueue(&kfs[i]);
    CHECK(q.Pending() == 4);  // bounded: the oldest (kfs[0]) was dropped (Modeler.cc:1468-1470)

    mapper.watch = &kfs[1];
    CHECK(q.ProcessOne());  // kfs[1]
    CHECK(mapper.recon_calls.size() == 1 && mapper.pinned_during_call == 1);  // pinned while the mapper works
    CHECK(kfs[1].pins == 1 && kfs[1].unpins == 1 && kfs[1].not_erase == 0);   // and released afterwards
    CHECK(map.keyframes.size() == 1 && q.Find(&kfs[1]) == map.keyframes[0]);
    CHECK(q.Find(&kfs[1])->mnId == 11 && q.Find(&kfs[1])->im_.at(1, 2) == (unsigned char)(W + 2 + 11));
    CHECK(lookups[101] == 1 && injected.empty());
    mapper.watch = nullptr;

    CHECK(q.ProcessOne());  // kfs[2] is bad: skipped before pinning (Modeler.cc:112-113)
    CHECK(kfs[2].pins == 0 && mapper.recon_calls.size() == 1 && lookups.count(102) == 0);

    CHECK(q.ProcessOne());  // kfs[3]: its frame is gone -> pinned, looked up, released, not mapped
    CHECK(kfs[3].pins == 1 && kfs[3].unpins == 1 && mapper.recon_calls.size() == 1 && q.Find(&kfs[3]) == nullptr);

    CHECK(q.ProcessOne());  // kfs[4]: finishes kfs[1] (lag 1) -> injected once
    CHECK(mapper.recon_calls.size() == 2 && map.keyframes.size() == 2);
    CHECK(injected.size() == 1 && injected[0].first == &kfs[1]);
    CHECK(injected[0].second.size() == 1);  // sigma <= 0.01 and rho > 1e-6 only (PM.cc:120-121)
    CHECK(injected[0].second[0].x == 12.f && injected[0].second[0].y == 2.f && injected[0].second[0].z == 3.f);
    // covisibility became mutual once both exist in the view
    CHECK(q.Find(&kfs[1])->covisible.size() == 1 && q.Find(&kfs[1])->covisible[0] == q.Find(&kfs[4]));
    CHECK(q.Find(&kfs[4])->covisible.size() == 1 && q.Find(&kfs[4])->covisible[0] == q.Find(&kfs[1]));
    CHECK(!q.ProcessOne() && q.Pending() == 0);

    // a keyframe processed twice is mapped once; finishing kfs[4] now injects it; kfs[1] is not injected again
    q.Enqueue(&kfs[1]);
    CHECK(q.ProcessOne());
    CHECK(map.keyframes.size() == 2 && mapper.recon_calls.size() == 3 && lookups[101] == 2);
    CHECK(injected.size() == 2 && injected[1].first == &kfs[4]);

    // bundle adjustment moved kfs[4]
    kfs[4].Tcw.at<float>(0, 3) = 0.25f;
    std::set<ORB_SLAM2::KeyFrame*> adj;
    adj.insert(&kfs[4]);
    adj.insert(&kfs[0]);  // never mapped: ignored
    q.OnPosesAdjusted(adj);  // any thread: only queues the new poses ...
    CHECK(mapper.update_all == 0 && !q.Find(&kfs[4])->poseChanged);
    kfs[4].Tcw.at<float>(0, 3) = 0.75f;  // (the pose was copied when the hook ran; a later change is a later event)
    CHECK(q.DrainEvents() == 2);  // ... the Modeler thread applies them
    CHECK(mapper.update_all == 1 && q.Find(&kfs[4])->poseChanged && q.Find(&kfs[4])->Tcw[3] == 0.25f && !q.Find(&kfs[1])->poseChanged);

    // the SLAM side erases kfs[1] while it is queued again
    q.Enqueue(&kfs[1]);
    q.Enqueue(&kfs[4]);
    sdm::KeyFrame* s1 = q.Find(&kfs[1]);
    q.OnKeyFrameErased(&kfs[1]);  // any thread: leaves the work queue at once, the rest is queued for the Modeler thread
    CHECK(q.Pending() == 1 && q.Find(&kfs[1]) == s1 && mapper.forgotten.empty());
    CHECK(q.DrainEvents() == 1);
    CHECK(q.Pending() == 1 && q.Find(&kfs[1]) == nullptr && mapper.forgotten.size() == 1 && mapper.forgotten[0] == s1);
    CHECK(map.keyframes.size() == 1 && map.keyframes[0] == q.Find(&kfs[4]) && q.Find(&kfs[4])->covisible.empty());
    q.OnKeyFrameErased(&kfs[0]);  // unknown keyframe: no-op
    CHECK(q.DrainEvents() == 1 && map.keyframes.size() == 1);

    // a finished keyframe that turned bad, or whose erasure is still queued, is not handed to the mesher
    // (Modeler.cc:112-116); every injection happens between SetNotErase and SetErase
    {
        std::vector<ORB_SLAM2::KeyFrame> k2(4);
        for (int i = 0; i < 4; i++) {
            k2[i] = kfs[4];
            k2[i].mnId = 50 + i;
            k2[i].mnFrameId = 200 + i;
            k2[i].bad = false;
            k2[i].pins = k2[i].unpins = k2[i].not_erase = 0;
            k2[i].cov.clear();
        }
        FakeMapper m2;
        m2.lag = 1;
        sdm::Map map2;
        std::vector<ORB_SLAM2::KeyFrame*> inj2;
        std::vector<int> pinned_at_injection;
        auto inject2 = [&](ORB_SLAM2::KeyFrame* k, std::vector<cv::Point3f>&) {
            inj2.push_back(k);
            pinned_at_injection.push_back(k->not_erase);
        };
        sdm_adapter::SemiDenseQueueT<FakeMapper> q2(&m2, &map2, image, inject2, 8);
        for (int i = 0; i < 4; i++) q2.Enqueue(&k2[i]);
        CHECK(q2.ProcessOne());        // k2[0] reconstructed
        k2[0].bad = true;              // ... and culled by the SLAM side before it is finished
        CHECK(q2.ProcessOne());        // k2[1]: finishes k2[0], which is bad -> not injected
        CHECK(inj2.empty());
        q2.OnKeyFrameErased(&k2[1]);   // erasure queued (not drained yet) ...
        m2.lag = 0;                    // (from now on a keyframe is finished by its own call)
        CHECK(q2.ProcessOne());        // k2[2]: drains the erasure first, finishes k2[2] itself -> injected, pinned
        CHECK(inj2.size() == 1 && inj2[0] == &k2[2] && pinned_at_injection[0] == 2);  // ProcessOne's pin + the injection's
        CHECK(k2[2].not_erase == 0 && q2.Find(&k2[1]) == nullptr);
    }
    std::printf("OK\n");
    return 0;
}
