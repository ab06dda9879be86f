// adapters/semi_dense_queue.h -- the seam between the fork's Modeler thread and the MI355X engine (SURVEY.md §3.3,
// §8f-3): queue + pin/unpin + trigger + injection, following the fork's own deferred per-keyframe densifier:
//
//   Modeler::DetectLineSegmentsLater(pKF)   src/Modeler/Modeler.cc:1465-1472   bounded FIFO under its own mutex, oldest dropped
//     (call site, commented out)            src/Modeler/Modeler.cc:1350        in Modeler::AddKeyFrameEntry, LocalMapping thread
//                                           src/LocalMapping.cc:93-94          ... which LocalMapping::Run calls per new keyframe
//   Modeler::Run idle branch                src/Modeler/Modeler.cc:63-66       drains one keyframe per pass of the loop
//   Modeler::AddPointsOnLineSegments()      src/Modeler/Modeler.cc:100-128     pop; skip bad; SetNotErase; compute points;
//                                                                               addKeyFrameInsertionWithLinesEntry(pKF, new KeyFrame(pKF), pts)
//                                                                               under mMutexTranscript; SetErase
//   frame lookup                            src/Modeler/Modeler.cc:143-155     mmFrameQueue[pKF->mnFrameId], RGB -> gray
//
// SemiDenseQueueT does the same with "compute points" = ProbabilityMapping::SemiDenseRecon on the GPU.  Everything
// that belongs to the Modeler stays the Modeler's, reached through two callbacks:
//   ImageProvider  (pKF, cv::Mat& gray) -> bool   the stored frame of the keyframe as CV_8UC1 (Modeler.cc:143-155)
//   Injector       (pKF, points)                  lock mMutexTranscript and call
//                                                 mTranscriptInterface.addKeyFrameInsertionWithLinesEntry(pKF, new KeyFrame(pKF), points)
// The mapper type is a template parameter so that the queue logic is unit-tested on CPU with a recording stand-in
// (tests/cpp/test_semi_dense_queue.cc); production code uses SemiDenseQueue = SemiDenseQueueT<ProbabilityMapping>.
// Needs OpenCV and the fork's headers, like orbslam_carv_adapter.h; compiled here against tests/cpp/mock_fork.
#pragma once
#include <array>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

#include "orbslam_carv_adapter.h"

namespace sdm_adapter {

// THREADING CONTRACT.  One thread -- the Modeler thread -- owns the mapper, the map view and every
// container below except the queues: it is the only caller of ProcessOne() / DrainEvents() / Find().  Enqueue(),
// OnKeyFrameErased() and OnPosesAdjusted() may be called from any thread (LocalMapping, LoopClosing: the places that
// already notify the Modeler, src/LocalMapping.cc:93-94, src/Optimizer.cc:761-790, src/LoopClosing.cc:651-749); they
// only append to queues under mutex_ and never touch the mapper (the GPU context is single-caller).  The Modeler
// thread applies the queued erasures and pose updates at the start of its next ProcessOne() / DrainEvents().
// ProbabilityMapping::Run() is the OTHER way to drive the mapper (the reference's own thread, PM.cc:65-135); a
// process uses one of the two drivers, not both.
template <class Mapper>
class SemiDenseQueueT {
public:
    typedef std::function<bool(ORB_SLAM2::KeyFrame*, cv::Mat&)> ImageProvider;
    typedef std::function<void(ORB_SLAM2::KeyFrame*, std::vector<cv::Point3f>&)> Injector;

    // max_queue: the bound of mdToLinesQueue (mnMaxToLinesQueueSize, include/Modeler/Modeler.h); max_sigma: the obj
    // writer's filter sigma <= 0.01 (PM.cc:120)
    SemiDenseQueueT(Mapper* mapper, sdm::Map* map, ImageProvider image, Injector inject, size_t max_queue = 100,
                    double max_sigma = 0.01)
        : mapper_(mapper), map_(map), image_(image), inject_(inject), max_queue_(max_queue), max_sigma_(max_sigma)
    {
    }

    // LocalMapping thread: Modeler::AddKeyFrameEntry -> here (the DetectLineSegmentsLater pattern, Modeler.cc:1465-1472)
    void Enqueue(ORB_SLAM2::KeyFrame* pKF)
    {
        std::unique_lock<std::mutex> lock(mutex_);
        if (queue_.size() >= max_queue_) queue_.pop_front();
        queue_.push_back(pKF);
    }

    size_t Pending()
    {
        std::unique_lock<std::mutex> lock(mutex_);
        return queue_.size();
    }

    // Modeler thread, idle branch of Modeler::Run (Modeler.cc:63-66): one keyframe per call, like
    // AddPointsOnLineSegments (Modeler.cc:100-128).  Returns false when there was nothing to do.
    bool ProcessOne()
    {
        DrainEvents();
        ORB_SLAM2::KeyFrame* pKF;
        {
            std::unique_lock<std::mutex> lock(mutex_);
            if (queue_.empty()) return false;
            pKF = queue_.front();
            queue_.pop_front();
        }
        if (pKF->isBad()) return true;  // Modeler.cc:112-113
        pKF->SetNotErase();             // Modeler.cc:116: the keyframe must not be erased while this thread works on it
        cv::Mat gray;
        if (image_(pKF, gray)) {
            std::unique_ptr<sdm::KeyFrame>& slot = owned_[pKF];
            if (!slot) {
                slot.reset(new sdm::KeyFrame());
                FillSemiDenseKeyFrame(pKF, gray, *slot, registry_);
                map_->AddKeyFrame(slot.get());  // under the map's mutex (src/Map.cc:38-44)
                // covisibility is mutual: keyframes filled earlier did not know this one yet (FillSemiDenseKeyFrame
                // only links registered keyframes), so refresh their lists now that it exists
                RefreshCovisibility();
            }
            mapper_->SemiDenseRecon(slot.get());  // PM.cc:137-256, then PM.cc:262-315 for every keyframe that became ready
            InjectFinished();
        }
        pKF->SetErase();  // Modeler.cc:126
        return true;
    }

    // Map::EraseKeyFrame / KeyFrame::SetBadFlag (src/Map.cc:55-66, src/KeyFrame.cc:449-497), any thread: the SLAM side
    // is about to drop the keyframe.  It leaves the work queue at once; the map view and the device-slot cache (address
    // reuse!) follow on the Modeler thread, and until then nothing is injected for it.
    void OnKeyFrameErased(ORB_SLAM2::KeyFrame* pKF)
    {
        std::unique_lock<std::mutex> lock(mutex_);
        for (typename std::deque<ORB_SLAM2::KeyFrame*>::iterator it = queue_.begin(); it != queue_.end();)
            it = (*it == pKF) ? queue_.erase(it) : it + 1;
        erased_.insert(pKF);
    }

    // Bundle adjustment moved keyframes (the hooks that already notify the Modeler: src/Optimizer.cc:761-790,
    // src/LoopClosing.cc:651-749), any thread: the new poses are copied HERE (the caller holds them consistent) and
    // applied -- with the re-projection of the finished keyframes, PM.cc:321-334 -- on the Modeler thread.
    void OnPosesAdjusted(const std::set<ORB_SLAM2::KeyFrame*>& adjusted)
    {
        std::vector<std::pair<ORB_SLAM2::KeyFrame*, std::array<float, 12> > > upd;
        for (std::set<ORB_SLAM2::KeyFrame*>::const_iterator it = adjusted.begin(); it != adjusted.end(); ++it) {
            cv::Mat Tcw = (*it)->GetPose();
            std::array<float, 12> t;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 4; c++) t[r * 4 + c] = Tcw.at<float>(r, c);
            upd.push_back(std::make_pair(*it, t));
        }
        std::unique_lock<std::mutex> lock(mutex_);
        for (size_t i = 0; i < upd.size(); i++) poses_.push_back(upd[i]);
    }

    // Modeler thread: apply the queued erasures and pose updates (ProcessOne does it first thing; call it from the
    // idle branch, too, when no keyframe is waiting).  Returns the number of events applied.
    size_t DrainEvents()
    {
        std::set<ORB_SLAM2::KeyFrame*> erased;
        std::vector<std::pair<ORB_SLAM2::KeyFrame*, std::array<float, 12> > > poses;
        {
            std::unique_lock<std::mutex> lock(mutex_);
            erased.swap(erased_);
            poses.swap(poses_);
        }
        for (std::set<ORB_SLAM2::KeyFrame*>::iterator e = erased.begin(); e != erased.end(); ++e) Erase(*e);
        bool moved = false;
        for (size_t i = 0; i < poses.size(); i++) {
            typename Owned::iterator o = owned_.find(poses[i].first);
            if (o == owned_.end()) continue;  // never mapped, or erased meanwhile
            for (int k = 0; k < 12; k++) o->second->Tcw[k] = poses[i].second[k];
            o->second->poseChanged = true;  // PM.cc:329
            moved = true;
        }
        if (moved) mapper_->UpdateAllSemiDensePointSet();
        return erased.size() + poses.size();
    }

    sdm::KeyFrame* Find(ORB_SLAM2::KeyFrame* pKF)
    {
        typename Owned::iterator it = owned_.find(pKF);
        return it == owned_.end() ? nullptr : it->second.get();
    }

private:
    typedef std::map<ORB_SLAM2::KeyFrame*, std::unique_ptr<sdm::KeyFrame> > Owned;

    void Erase(ORB_SLAM2::KeyFrame* pKF)  // Modeler thread
    {
        typename Owned::iterator it = owned_.find(pKF);
        if (it == owned_.end()) return;
        sdm::KeyFrame* skf = it->second.get();
        mapper_->Forget(skf);
        map_->EraseKeyFrame(skf);  // under the map's mutex (src/Map.cc:55-66)
        for (typename Owned::iterator o = owned_.begin(); o != owned_.end(); ++o) {
            std::vector<sdm::KeyFrame*>& c = o->second->covisible;
            for (size_t j = 0; j < c.size();) j = (c[j] == skf) ? (c.erase(c.begin() + j), j) : j + 1;
        }
        registry_.erase(pKF);
        injected_.erase(skf);
        owned_.erase(it);
    }

    void RefreshCovisibility()
    {
        for (typename Owned::iterator it = owned_.begin(); it != owned_.end(); ++it) {
            it->second->covisible.clear();
            std::vector<ORB_SLAM2::KeyFrame*> cov = it->first->GetVectorCovisibleKeyFrames();  // src/KeyFrame.cc:168-172
            for (size_t i = 0; i < cov.size(); i++)
                if (registry_.count(cov[i])) it->second->covisible.push_back(registry_[cov[i]]);
        }
    }

    bool ErasePending(ORB_SLAM2::KeyFrame* pKF)
    {
        std::unique_lock<std::mutex> lock(mutex_);
        return erased_.count(pKF) != 0;
    }

    // every keyframe whose inter-keyframe check has completed (interKF_depth_flag_, PM.cc:306) is handed to the
    // mesher exactly once: points with sigma <= max_sigma and rho > 1e-6 (PM.cc:120-121), raster order.  Like the
    // keyframe ProcessOne works on, each one is skipped when bad (Modeler.cc:112-113) and pinned around the injection
    // (Modeler.cc:116,126); one the SLAM side reported as erased since the last drain is left alone.
    void InjectFinished()
    {
        for (typename Owned::iterator it = owned_.begin(); it != owned_.end(); ++it) {
            sdm::KeyFrame* k = it->second.get();
            if (!k->interKF_depth_flag_ || injected_.count(k)) continue;
            if (ErasePending(it->first) || it->first->isBad()) continue;
            it->first->SetNotErase();
            std::vector<cv::Point3f> pts;
            for (int y = 0; y < k->im_.rows; y++)
                for (int x = 0; x < k->im_.cols; x++) {
                    if (k->depth_sigma_.at(y, x) > max_sigma_) continue;
                    if (!(k->depth_map_.at(y, x) > 0.000001)) continue;
                    pts.push_back(cv::Point3f(k->SemiDensePointSets_.at(y, 3 * x), k->SemiDensePointSets_.at(y, 3 * x + 1),
                                              k->SemiDensePointSets_.at(y, 3 * x + 2)));
                }
            inject_(it->first, pts);
            injected_.insert(k);
            it->first->SetErase();
        }
    }

    Mapper* mapper_;
    sdm::Map* map_;
    ImageProvider image_;
    Injector inject_;
    size_t max_queue_;
    double max_sigma_;
    std::mutex mutex_;  // mMutexToLines: guards queue_, erased_, poses_ -- everything another thread may touch
    std::deque<ORB_SLAM2::KeyFrame*> queue_;
    std::set<ORB_SLAM2::KeyFrame*> erased_;
    std::vector<std::pair<ORB_SLAM2::KeyFrame*, std::array<float, 12> > > poses_;
    // Modeler thread only:
    Owned owned_;
    Registry registry_;
    std::set<sdm::KeyFrame*> injected_;
};

}  // namespace sdm_adapter

#ifdef SDM_PROBABILITY_MAPPING_H
namespace sdm_adapter {
typedef SemiDenseQueueT<ProbabilityMapping> SemiDenseQueue;
}
#endif
