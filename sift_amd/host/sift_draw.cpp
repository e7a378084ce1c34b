// sift_draw.cpp -- the reference's drawing and pyramid-dump helpers (sift/sift.cpp:1772-1884) for builds without
// OpenCV.  Plain host code: nothing here touches the GPU.
#include "sift_gpu.h"

#include <cmath>
#include <cstdio>
#include <string>

namespace {

const double kPi = 3.1415926535897932384626433832795; // CV_PI

inline int cv_round(double v) { return (int)std::lrint(v); } // cvRound: to nearest, ties to even

inline void put_pixel(Mat &img, int x, int y, const cv::Scalar &color)
{
    if (x < 0 || y < 0 || x >= img.cols || y >= img.rows)
        return;
    unsigned char *p = img.data + img.step * (size_t)y + (size_t)x * img.channels();
    for (int c = 0; c < img.channels() && c < 4; c++) {
        const double v = color.val[c];
        p[c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : cv_round(v)));
    }
}

// 8-connected line, both end points included
void draw_line(Mat &img, cv::Point a, cv::Point b, const cv::Scalar &color)
{
    if (img.empty() || img.depth() != CV_8U)
        return;
    int dx = std::abs(b.x - a.x), dy = -std::abs(b.y - a.y);
    const int sx = a.x < b.x ? 1 : -1, sy = a.y < b.y ? 1 : -1;
    // a line that lies wholly outside the image draws nothing; bound the walk of far end points
    const long long span = (long long)dx - dy;
    if (span > 4LL * (img.cols + img.rows) + 65536)
        return;
    int err = dx + dy, x = a.x, y = a.y;
    for (;;) {
        put_pixel(img, x, y, color);
        if (x == b.x && y == b.y)
            break;
        const int e2 = 2 * err;
        if (e2 >= dy) {
            err += dy;
            x += sx;
        }
        if (e2 <= dx) {
            err += dx;
            y += sy;
        }
    }
}

// midpoint circle of thickness 1
void draw_circle(Mat &img, cv::Point c, int radius, const cv::Scalar &color)
{
    if (img.empty() || img.depth() != CV_8U || radius < 0)
        return;
    int x = radius, y = 0, err = 1 - radius;
    while (x >= y) {
        put_pixel(img, c.x + x, c.y + y, color);
        put_pixel(img, c.x + y, c.y + x, color);
        put_pixel(img, c.x - y, c.y + x, color);
        put_pixel(img, c.x - x, c.y + y, color);
        put_pixel(img, c.x - x, c.y - y, color);
        put_pixel(img, c.x - y, c.y - x, color);
        put_pixel(img, c.x + y, c.y - x, color);
        put_pixel(img, c.x + x, c.y - y, color);
        y++;
        if (err < 0) {
            err += 2 * y + 1;
        } else {
            x--;
            err += 2 * (y - x) + 1;
        }
    }
}

} // namespace

// sift.cpp:1772-1806
void DrawSiftFeature(Mat &src, Keypoint &feat, cv::Scalar color)
{
    const double scale = 5.0, hscale = 0.75;
    const int start_x = cv_round(feat.dx), start_y = cv_round(feat.dy);
    const double scl = feat.scale, ori = feat.ori;
    const int len = cv_round(scl * scale), hlen = cv_round(scl * hscale), blen = len - hlen;
    const int end_x = cv_round(len * std::cos(ori)) + start_x;
    const int end_y = cv_round(len * -std::sin(ori)) + start_y;
    const int h1_x = cv_round(blen * std::cos(ori + kPi / 18.0)) + start_x;
    const int h1_y = cv_round(blen * -std::sin(ori + kPi / 18.0)) + start_y;
    const int h2_x = cv_round(blen * std::cos(ori - kPi / 18.0)) + start_x;
    const int h2_y = cv_round(blen * -std::sin(ori - kPi / 18.0)) + start_y;
    const cv::Point start(start_x, start_y), end(end_x, end_y), h1(h1_x, h1_y), h2(h2_x, h2_y);
    draw_line(src, start, end, color);
    draw_line(src, end, h1, color);
    draw_line(src, end, h2, color);
}

// sift.cpp:1808-1815
void DrawSiftFeatures(Mat &src, std::vector<Keypoint> &features)
{
    const cv::Scalar color = CV_RGB(0, 255, 0);
    for (size_t i = 0; i < features.size(); i++)
        DrawSiftFeature(src, features[i], color);
}

// sift.cpp:1817-1827
void DrawKeyPoints(Mat &src, std::vector<Keypoint> &keypoints)
{
    const cv::Scalar color = CV_RGB(0, 0, 255);
    for (size_t i = 0; i < keypoints.size(); i++)
        draw_circle(src, cv::Point(keypoints[i].dx, keypoints[i].dy), 3, color);
}

// sift.cpp:1829-1840
void DrawKeyPointsRight(Mat &src, std::vector<Keypoint> &keypoints, cv::Size size)
{
    const cv::Scalar color = CV_RGB(0, 0, 255);
    for (size_t i = 0; i < keypoints.size(); i++)
        draw_circle(src, cv::Point(keypoints[i].dx + size.width, keypoints[i].dy), 3, color);
}

// sift.cpp:1843-1847: normalize(src, dst, 0, 255, NORM_MINMAX), then convertTo(CV_8U) (rounded, saturated).
// A constant image maps to 0, as cv::normalize does when max == min.
void cv64f_to_cv8U(const Mat &src, Mat &dst)
{
    if (src.empty() || src.type() != CV_64FC1)
        return;
    double lo = 0, hi = 0;
    bool first = true;
    for (int y = 0; y < src.rows; y++) {
        const double *r = reinterpret_cast<const double *>(src.data + src.step * (size_t)y);
        for (int x = 0; x < src.cols; x++) {
            if (first || r[x] < lo)
                lo = r[x];
            if (first || r[x] > hi)
                hi = r[x];
            first = false;
        }
    }
    const double k = hi > lo ? 255.0 / (hi - lo) : 0.0;
    Mat out(src.rows, src.cols, CV_8UC1);
    for (int y = 0; y < src.rows; y++) {
        const double *r = reinterpret_cast<const double *>(src.data + src.step * (size_t)y);
        unsigned char *o = out.data + out.step * (size_t)y;
        for (int x = 0; x < src.cols; x++) {
            const double v = (r[x] - lo) * k;
            o[x] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : cv_round(v)));
        }
    }
    dst = out;
}

// sift.cpp:1858-1863 (binary PGM in place of cv::imwrite)
void writecv64f(const char *filename, const Mat &mat)
{
    Mat dst;
    cv64f_to_cv8U(mat, dst);
    if (dst.empty())
        return;
    FILE *f = std::fopen(filename, "wb");
    if (!f)
        return;
    std::fprintf(f, "P5\n%d %d\n255\n", dst.cols, dst.rows);
    for (int y = 0; y < dst.rows; y++)
        std::fwrite(dst.data + dst.step * (size_t)y, 1, (size_t)dst.cols, f);
    std::fclose(f);
}

// sift.cpp:1849-1873: level i goes to <dir><i>.<ext> (the reference's format string is "%s\%d.jpg")
void write_pyr(const std::vector<Mat> &pyr, const char *dir)
{
    for (size_t i = 0; i < pyr.size(); i++) {
        const std::string name = std::string(dir) + std::to_string(i) + ".pgm";
        writecv64f(name.c_str(), pyr[i]);
    }
}
