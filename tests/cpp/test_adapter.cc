// Compiles adapters/orbslam_carv_adapter.h against test doubles of the fork's KeyFrame/MapPoint and of the cv::Mat
// operations it uses (tests/cpp/mock_fork), and checks what it writes into sdm::KeyFrame.  CPU only; no GPU calls.
#include <cmath>
#include <cstdio>

#include "orbslam_carv_adapter.h"

#define CHECK(c)                                                 \
    do {                                                         \
        if (!(c)) {                                              \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            return 1;                                            \
        }                                                        \
    } while (0)

int main()
{
    const int W = 16, H = 8;
    cv::Mat gray(H, W, CV_8UC1);
    for (int i = 0; i < W * H; i++) gray.data[i] = (unsigned char)(i * 7);
    ORB_SLAM2::KeyFrame a, b;
    a.mnId = 11;
    b.mnId = 12;
    for (ORB_SLAM2::KeyFrame* k : {&a, &b}) {
        k->fx = 517.3f; k->fy = 516.5f; k->cx = 318.6f; k->cy = 255.3f;
        k->Tcw = cv::Mat(4, 4, CV_32F);
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) k->Tcw.at<float>(r, c) = (r == c) ? 1.f : 0.f;
    }
    // b: rotated 90 deg about y and shifted, so row 2 of Rcw is (-1, 0, 0) and t_z = 0.5
    b.Tcw.at<float>(0, 0) = 0; b.Tcw.at<float>(0, 2) = 1; b.Tcw.at<float>(2, 0) = -1; b.Tcw.at<float>(2, 2) = 0;
    b.Tcw.at<float>(0, 3) = 0.25f; b.Tcw.at<float>(2, 3) = 0.5f;
    ORB_SLAM2::MapPoint p0, p1, p2;
    p0.mnId = 100; p1.mnId = 101; p2.mnId = 102;
    p1.bad = true;
    for (ORB_SLAM2::MapPoint* p : {&p0, &p1, &p2}) p->pos = cv::Mat(3, 1, CV_32F);
    p0.pos.at<float>(0, 0) = -2.f; p0.pos.at<float>(1, 0) = 0.3f; p0.pos.at<float>(2, 0) = 4.f;
    p2.pos.at<float>(0, 0) = -1.f; p2.pos.at<float>(1, 0) = 0.f; p2.pos.at<float>(2, 0) = 9.f;
    b.mps = {&p0, nullptr, &p1, &p2};
    b.mvKeysUn.resize(4);
    b.mvKeysUn[0].angle = 10.f; b.mvKeysUn[1].angle = 20.f; b.mvKeysUn[2].angle = 30.f; b.mvKeysUn[3].angle = 40.f;
    a.mps = {};
    b.cov = {&a};
    a.cov = {&b};  // b is not registered yet when a is filled

    sdm_adapter::Registry reg;
    sdm::KeyFrame sa, sb;
    sdm_adapter::FillSemiDenseKeyFrame(&a, gray, sa, reg);
    CHECK(sa.mnId == 11 && sa.im_.rows == H && sa.im_.cols == W && sa.im_.at(3, 5) == (unsigned char)((3 * W + 5) * 7));
    CHECK(sa.covisible.empty());  // unknown covisible keyframes are skipped, not invented
    CHECK(sa.GradImg.empty() && sa.GradTheta.empty());
    sdm_adapter::FillSemiDenseKeyFrame(&b, gray, sb, reg);
    CHECK(sb.covisible.size() == 1 && sb.covisible[0] == &sa);
    CHECK(sb.fx == 517.3f && sb.cy == 255.3f && !sb.isBad());
    CHECK(sb.Tcw[0] == 0.f && sb.Tcw[2] == 1.f && sb.Tcw[3] == 0.25f && sb.Tcw[8] == -1.f && sb.Tcw[11] == 0.5f);
    // depth of a map point in b's camera = Rcw.row(2) . Xw + t_z  (src/KeyFrame.cc:644-662)
    CHECK(sb.point_depths.size() == 2);
    CHECK(std::fabs(sb.point_depths[0] - (2.f + 0.5f)) < 1e-6f && std::fabs(sb.point_depths[1] - (1.f + 0.5f)) < 1e-6f);
    CHECK(sb.map_point_ids.size() == 4 && sb.map_point_ids[0] == 100 && sb.map_point_ids[1] == -1 && sb.map_point_ids[2] == -1 &&
          sb.map_point_ids[3] == 102);
    CHECK(sb.keypoint_angles.size() == 4 && sb.keypoint_angles[2] == 30.f);
    CHECK(reg.size() == 2 && reg[&b] == &sb);
    std::printf("OK\n");
    return 0;
}
