// TEST DOUBLE of the accessors of ORB_SLAM2::MapPoint the adapter calls (fork: include/MapPoint.h).
#pragma once
#include <opencv2/core/core.hpp>
namespace ORB_SLAM2 {
class MapPoint {
public:
    long unsigned int mnId = 0;
    bool bad = false;
    cv::Mat pos;  // 3x1 CV_32F
    bool isBad() const { return bad; }
    cv::Mat GetWorldPos() const { return pos.clone(); }
};
}  // namespace ORB_SLAM2
