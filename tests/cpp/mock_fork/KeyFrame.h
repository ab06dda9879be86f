// TEST DOUBLE of the members of ORB_SLAM2::KeyFrame the adapter reads (fork: include/KeyFrame.h:43-239).
#pragma once
#include <vector>
#include <opencv2/core/core.hpp>
#include "MapPoint.h"
namespace ORB_SLAM2 {
class KeyFrame {
public:
    long unsigned int mnId = 0;
    long unsigned int mnFrameId = 0;
    int not_erase = 0, pins = 0, unpins = 0;  // SetNotErase / SetErase bookkeeping (src/KeyFrame.cc:419-447)
    void SetNotErase() { not_erase++; pins++; }
    void SetErase() { not_erase--; unpins++; }
    float fx = 0, fy = 0, cx = 0, cy = 0;
    std::vector<cv::KeyPoint> mvKeysUn;
    cv::Mat Tcw;  // 4x4 CV_32F
    bool bad = false;
    std::vector<KeyFrame*> cov;
    std::vector<MapPoint*> mps;
    cv::Mat GetPose() const { return Tcw.clone(); }
    bool isBad() const { return bad; }
    std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() const { return cov; }
    std::vector<MapPoint*> GetMapPointMatches() const { return mps; }
};
}  // namespace ORB_SLAM2
