// sift_legacy.cpp -- the reference's off-path helpers, so that a program written against sift/sift.h links
// unchanged: the two 8-bit legacy blurs (never called by the pipeline), the dump-name helpers and the
// Inverse3D print-out.  Plain host code; none of it touches the GPU or the extraction path.
#include "sift_gpu.h"

#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

namespace {

// Direct 2-D correlation of an 8-bit image with a k x k kernel, the way both legacy blurs do it
// (sift/sift.cpp:203-233, 289-310): the window's top-left corner runs over [0, rows-k) x [0, cols-k) -- so the
// last valid window position is skipped -- the result is truncated to int and stored at the window centre; pixels
// no window centre reaches keep whatever dst held.  `kern` is indexed [column offset][row offset] as there.
void correlate_u8(const Mat &src, Mat &dst, const std::vector<double> &kern, int k, int channels_used)
{
    const int ch = src.channels(), half = (k - 1) / 2;
    for (int j = 0; j < src.cols - k; j++)
        for (int i = 0; i < src.rows - k; i++)
            for (int c = 0; c < channels_used; c++) {
                double acc = 0;
                for (int m = 0; m < k; m++)
                    for (int n = 0; n < k; n++)
                        acc += src.data[src.step * (size_t)(i + n) + (size_t)ch * (j + m) + c] * kern[(size_t)m * k + n];
                dst.data[dst.step * (size_t)(i + half) + (size_t)dst.channels() * (j + half) + c] = (unsigned char)(int)acc;
            }
}

} // namespace

// sift/sift.cpp:183-236: fixed 7x7 template (sigma = 0.84089642, normalised), one or three channels.
void GaussianTemplateSmooth(const Mat &src, Mat &dst)
{
    if (src.empty() || src.depth() != CV_8U)
        return;
    // the template is the outer product of this column with itself, rounded to 8 decimals in the reference's table
    static const double t[7][7] = {
        {0.00000067, 0.00002292, 0.00019117, 0.00038771, 0.00019117, 0.00002292, 0.00000067},
        {0.00002292, 0.00078633, 0.00655965, 0.01330373, 0.00655965, 0.00078633, 0.00002292},
        {0.00019117, 0.00655965, 0.05472157, 0.11098164, 0.05472157, 0.00655965, 0.00019117},
        {0.00038771, 0.01330373, 0.11098164, 0.22508352, 0.11098164, 0.01330373, 0.00038771},
        {0.00019117, 0.00655965, 0.05472157, 0.11098164, 0.05472157, 0.00655965, 0.00019117},
        {0.00002292, 0.00078633, 0.00655965, 0.01330373, 0.00655965, 0.00078633, 0.00002292},
        {0.00000067, 0.00002292, 0.00019117, 0.00038771, 0.00019117, 0.00002292, 0.00000067}};
    std::vector<double> kern(49);
    for (int m = 0; m < 7; m++)
        for (int n = 0; n < 7; n++)
            kern[(size_t)m * 7 + n] = t[m][n];
    dst.create(src.size(), src.type());
    correlate_u8(src, dst, kern, 7, src.channels() == 1 ? 1 : 3);
}

// sift/sift.cpp:245-314: ksize = 2*cvRound(3 sigma) + 1, kernel exp(-(x^2+y^2)/(2 sigma^2)) / (2 pi sigma^2)
// (pi taken as 3.141592653) divided by its sum; one-channel 8-bit images only.
void GaussianSmooth2D(const Mat &src, Mat &dst, double sigma)
{
    if (src.empty() || src.channels() != 1 || src.depth() != CV_8U)
        return;
    sigma = sigma > 0 ? sigma : 0;
    const int k = (int)std::lrint(sigma * 3) * 2 + 1;
    if (k == 1) {
        src.copyTo(dst);
        return;
    }
    dst.create(src.size(), src.type());
    std::vector<double> kern((size_t)k * k);
    const double scale = -0.5 / (sigma * sigma), cons = -scale / 3.141592653;
    double sum = 0;
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++) {
            const int x = i - (k - 1) / 2, y = j - (k - 1) / 2;
            sum += kern[(size_t)i * k + j] = cons * std::exp(scale * (x * x + y * y));
        }
    for (double &v : kern)
        v /= sum;
    correlate_u8(src, dst, kern, k, 1);
}

// sift/sift.cpp:1849-1855.  The reference's format string is "%s\%d.jpg": the backslash before %d is not an
// escape, so the name is <dir><i>.jpg; its 50-byte buffer is leaked by every caller.  Here the buffer is per
// thread and reused, and long names are truncated instead of overrunning it.
const char *GetFileName(const char *dir, int i)
{
    thread_local char name[4096];
    std::snprintf(name, sizeof name, "%s%d.jpg", dir ? dir : "", i);
    return name;
}

// sift/sift.cpp:1875-1884: prints the names write_pyr would use; shows nothing (the imshow call is commented out there)
void display_pyr(const std::vector<Mat> &pyr, std::string dir)
{
    for (size_t i = 0; i < pyr.size(); i++) {
        std::cout << "name: " << dir << " " << i << std::endl;
        std::cout << dir + std::to_string(i) + ".jpg" << std::endl;
    }
}

// sift/sift.cpp:1934-1960: prints H * H^-1 for H = {3,2,3; 4,5,6; 7,8,9} with the adjugate inverse the
// refinement uses (sift/sift.cpp:720-766); "Not inverse." when |det| < 1e-10.
void testInverse3D()
{
    const double H[9] = {3, 2, 3, 4, 5, 6, 7, 8, 9};
    const double det = H[0] * H[4] * H[8] + H[1] * H[5] * H[6] + H[2] * H[3] * H[7] - H[0] * H[5] * H[7] -
                       H[1] * H[3] * H[8] - H[2] * H[4] * H[6];
    if (std::fabs(det) < 1e-10) {
        std::cout << "Not inverse." << std::endl;
        return;
    }
    const double inv[9] = {(H[4] * H[8] - H[5] * H[7]) / det, -(H[1] * H[8] - H[2] * H[7]) / det,
                           (H[1] * H[5] - H[2] * H[4]) / det, (H[5] * H[6] - H[3] * H[8]) / det,
                           -(H[2] * H[6] - H[0] * H[8]) / det, (H[2] * H[3] - H[0] * H[5]) / det,
                           (H[3] * H[7] - H[4] * H[6]) / det, -(H[0] * H[7] - H[1] * H[6]) / det,
                           (H[0] * H[4] - H[1] * H[3]) / det};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            double r = 0;
            for (int k = 0; k < 3; k++)
                r += H[3 * i + k] * inv[3 * k + j];
            std::cout << r << " ";
        }
        std::cout << std::endl;
    }
}
