// cvmini.h -- the few cv:: types the SIFT entry points take, for builds without OpenCV.
//
// The reference's interface (sift/sift.h) passes images as cv::Mat and uses it only as a 2-D buffer:
// rows, cols, data, step, dataend, create(), copyTo(), clone(), empty(), size(), type(), channels()
// (and cv::Point, cv::Scalar, CV_RGB for its drawing helpers).
// This header provides exactly that much, with reference counting like cv::Mat's shallow copies.
// Define SIFT_WITH_OPENCV before including sift_gpu.h to use the real OpenCV types instead.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <memory>

namespace cv {

typedef unsigned char uchar;

enum { CV_8U_DEPTH = 0, CV_64F_DEPTH = 6 };
#define CV_MAKETYPE_MINI(depth, cn) ((depth) + (((cn)-1) << 3))
#ifndef CV_8UC3
#define CV_8U 0
#define CV_64F 6
#define CV_8UC1 CV_MAKETYPE_MINI(0, 1)
#define CV_8UC3 CV_MAKETYPE_MINI(0, 3)
#define CV_8UC4 CV_MAKETYPE_MINI(0, 4)
#define CV_64FC1 CV_MAKETYPE_MINI(6, 1)
#endif

struct Size {
    int width = 0, height = 0;
    Size() {}
    Size(int w, int h) : width(w), height(h) {}
};

struct Point {
    int x = 0, y = 0;
    Point() {}
    // like cv::Point's converting constructor: floating coordinates are rounded (cv::saturate_cast<int>)
    template <typename A, typename B>
    Point(A px, B py) : x(round_to_int(px)), y(round_to_int(py)) {}

private:
    static int round_to_int(int v) { return v; }
    static int round_to_int(double v) { return (int)std::lrint(v); }
};

struct Scalar {
    double val[4] = {0, 0, 0, 0};
    Scalar() {}
    Scalar(double v0, double v1 = 0, double v2 = 0, double v3 = 0)
    {
        val[0] = v0; val[1] = v1; val[2] = v2; val[3] = v3;
    }
};
#ifndef CV_RGB
#define CV_RGB(r, g, b) cv::Scalar((b), (g), (r), 0)
#endif

class Mat {
public:
    int rows = 0, cols = 0;
    uchar *data = nullptr;
    uchar *dataend = nullptr;
    size_t step = 0;

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size s, int type) { create(s.height, s.width, type); }
    // wraps caller-owned pixels (no copy, no ownership)
    Mat(int r, int c, int type, void *ext, size_t step_bytes = 0) : rows(r), cols(c), type_(type)
    {
        data = static_cast<uchar *>(ext);
        step = step_bytes ? step_bytes : (size_t)c * elemSize();
        dataend = data + step * (size_t)r;
    }

    void create(int r, int c, int type)
    {
        if (data && r == rows && c == cols && type == type_)
            return;
        rows = r;
        cols = c;
        type_ = type;
        step = (size_t)c * elemSize();
        size_t bytes = step * (size_t)r;
        buf_.reset(static_cast<uchar *>(std::malloc(bytes ? bytes : 1)), std::free);
        data = buf_.get();
        dataend = data + bytes;
    }
    void create(Size s, int type) { create(s.height, s.width, type); }

    void copyTo(Mat &dst) const
    {
        dst.create(rows, cols, type_);
        for (int y = 0; y < rows; y++)
            std::memcpy(dst.data + dst.step * (size_t)y, data + step * (size_t)y, (size_t)cols * elemSize());
    }
    Mat clone() const
    {
        Mat m;
        copyTo(m);
        return m;
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    Size size() const { return Size(cols, rows); }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> 3) + 1; }
    size_t elemSize() const { return (size_t)channels() * (depth() == CV_64F ? 8 : 1); }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }

private:
    int type_ = 0;
    std::shared_ptr<uchar> buf_;
};

} // namespace cv
