// TEST DOUBLE, not OpenCV: the few cv::Mat operations adapters/orbslam_carv_adapter.h uses, with OpenCV's
// semantics (row-major CV_32F / CV_8U matrices, value-returning row/colRange/t, dot of equal-sized matrices).
// It exists so the adapter can be compiled and unit-tested in an image without OpenCV (tests/test_adapter.py).
#pragma once
#include <cassert>
#include <cstdint>
#include <memory>
#include <vector>

#define CV_8UC1 0
#define CV_32F 5
#define CV_Assert(expr) assert(expr)

namespace cv {
class Mat {
public:
    int rows = 0, cols = 0;
    unsigned char* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) : rows(r), cols(c), type_(type), buf_(std::make_shared<std::vector<unsigned char>>((size_t)r * c * (type == CV_32F ? 4 : 1)))
    {
        data = buf_->data();
    }
    int type() const { return type_; }
    bool isContinuous() const { return true; }
    template <typename T> T& at(int r, int c) { return reinterpret_cast<T*>(data)[(size_t)r * cols + c]; }
    template <typename T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(data)[(size_t)r * cols + c]; }
    Mat clone() const
    {
        Mat m(rows, cols, type_);
        *m.buf_ = *buf_;
        return m;
    }
    Mat row(int r) const
    {
        Mat m(1, cols, type_);
        for (int c = 0; c < cols; c++) m.at<float>(0, c) = at<float>(r, c);
        return m;
    }
    Mat colRange(int a, int b) const
    {
        Mat m(rows, b - a, type_);
        for (int r = 0; r < rows; r++)
            for (int c = a; c < b; c++) m.at<float>(r, c - a) = at<float>(r, c);
        return m;
    }
    Mat t() const
    {
        Mat m(cols, rows, type_);
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++) m.at<float>(c, r) = at<float>(r, c);
        return m;
    }
    double dot(const Mat& o) const
    {
        assert(rows * cols == o.rows * o.cols);
        double s = 0;
        for (int i = 0; i < rows * cols; i++) s += (double)reinterpret_cast<const float*>(data)[i] * reinterpret_cast<const float*>(o.data)[i];
        return s;
    }

private:
    int type_ = 0;
    std::shared_ptr<std::vector<unsigned char>> buf_;
};
struct Point3f {
    float x, y, z;
    Point3f() : x(0), y(0), z(0) {}
    Point3f(float a, float b, float c) : x(a), y(b), z(c) {}
};
struct KeyPoint {
    float angle = -1.f;
};
}  // namespace cv
