// adapters/orbslam_carv_adapter.h -- glue a maintainer of atlas-jj/ORB-SLAM-free-space-carving adds to call the
// MI355X engine from the fork.  It needs OpenCV and the fork's own headers (KeyFrame.h, MapPoint.h), neither of which
// is available in the build image; here it is compiled and unit-tested only against test doubles of exactly the
// members it touches (tests/cpp/mock_fork, tests/test_adapter.py).  See INTEGRATION.md §2-§3.
//
// It fills sdm::KeyFrame (include/sdm/ProbabilityMapping.h) -- exactly the members the reference's
// ProbabilityMapping.cc reads from ORB_SLAM2::KeyFrame -- from the real keyframe plus the
// undistorted gray frame that Tracking already hands to Modeler::AddFrameImage (src/Tracking.cc:266-271).
#pragma once
#include <cstring>
#include <map>

#include <opencv2/core/core.hpp>

#include "KeyFrame.h"   // ORB_SLAM2::KeyFrame  (fork: include/KeyFrame.h:43-239)
#include "MapPoint.h"   // ORB_SLAM2::MapPoint
#include "sdm/ProbabilityMapping.h"

namespace sdm_adapter {

typedef std::map<ORB_SLAM2::KeyFrame*, sdm::KeyFrame*> Registry;

inline void FillSemiDenseKeyFrame(ORB_SLAM2::KeyFrame* pKF, const cv::Mat& gray, sdm::KeyFrame& out, Registry& registry)
{
    CV_Assert(gray.type() == CV_8UC1 && gray.isContinuous());
    out.mnId = pKF->mnId;
    out.im_ = sdm::Mat<uint8_t>(gray.rows, gray.cols);
    std::memcpy(out.im_.ptr(), gray.data, (size_t)gray.rows * gray.cols);
    out.fx = pKF->fx;  // include/KeyFrame.h:162
    out.fy = pKF->fy;
    out.cx = pKF->cx;
    out.cy = pKF->cy;
    cv::Mat Tcw = pKF->GetPose();  // src/KeyFrame.cc:86-90, CV_32F 4x4
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) out.Tcw[r * 4 + c] = Tcw.at<float>(r, c);
    out.bad = pKF->isBad();  // src/KeyFrame.cc:547
    out.covisible.clear();   // descending covisibility weight, src/KeyFrame.cc:168-172
    std::vector<ORB_SLAM2::KeyFrame*> cov = pKF->GetVectorCovisibleKeyFrames();
    for (size_t i = 0; i < cov.size(); i++)
        if (registry.count(cov[i])) out.covisible.push_back(registry[cov[i]]);
    // ORB depths for StereoSearchConstraints (the loop of src/KeyFrame.cc:644-662) and keypoint angles for
    // GetRotInPlane (PM.cc:467-484)
    cv::Mat Rcw2 = Tcw.row(2).colRange(0, 3).t();
    float zcw = Tcw.at<float>(2, 3);
    const std::vector<ORB_SLAM2::MapPoint*> mps = pKF->GetMapPointMatches();  // src/KeyFrame.cc:277
    out.point_depths.clear();
    out.map_point_ids.assign(mps.size(), -1);
    out.keypoint_angles.resize(mps.size());
    for (size_t i = 0; i < mps.size(); i++) {
        out.keypoint_angles[i] = pKF->mvKeysUn[i].angle;  // include/KeyFrame.h:169
        if (!mps[i] || mps[i]->isBad()) continue;
        out.map_point_ids[i] = (int)mps[i]->mnId;
        out.point_depths.push_back((float)(Rcw2.dot(mps[i]->GetWorldPos()) + zcw));
    }
    registry[pKF] = &out;
    // GradImg / GradTheta / I_stddev stay empty: ProbabilityMapping computes them on the GPU (sdm_upload_image).
}

}  // namespace sdm_adapter
