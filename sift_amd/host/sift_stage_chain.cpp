// sift_stage_chain.cpp -- calls the stage-level entry points one after another, the way the reference's
// Sift() does internally (sift/sift.cpp:1646-1707), and checks that the result equals the one-call driver.
//   usage: sift_stage_chain <width> <height> <seed>      prints "stage chain ok: N features" and exits 0
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "sift_gpu.h"

static uint64_t sm64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static double lat(int ix, int iy, int k, uint64_t seed)
{
    uint64_t c = (uint64_t)(uint32_t)ix | ((uint64_t)(uint32_t)iy << 32);
    uint64_t h = sm64(seed ^ sm64(c) ^ sm64(0x1000 + (uint64_t)k));
    return ((double)(h >> 11) * (1.0 / 9007199254740992.0)) * 2 - 1;
}
static double vnoise(double x, double y, int k, uint64_t seed)
{
    double fx0 = std::floor(x), fy0 = std::floor(y);
    int ix = (int)fx0, iy = (int)fy0;
    double fx = x - fx0, fy = y - fy0;
    fx = fx * fx * (3 - 2 * fx);
    fy = fy * fy * (3 - 2 * fy);
    double a = lat(ix, iy, k, seed), b = lat(ix + 1, iy, k, seed), c = lat(ix, iy + 1, k, seed), d = lat(ix + 1, iy + 1, k, seed);
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy;
}

#define CHECK(cond, what)                                                       \
    do {                                                                        \
        if (!(cond)) {                                                          \
            std::fprintf(stderr, "FAILED: %s (%s)\n", what, SiftLastError());    \
            return 1;                                                           \
        }                                                                       \
    } while (0)

int main(int argc, char **argv)
{
    if (argc != 4) {
        std::fprintf(stderr, "usage: %s width height seed\n", argv[0]);
        return 2;
    }
    const int w = std::atoi(argv[1]), h = std::atoi(argv[2]);
    const uint64_t seed = std::strtoull(argv[3], nullptr, 10);
    static const double L[5] = {4, 8, 16, 32, 64}, A[5] = {20, 28, 36, 40, 40};
    Mat src(h, w, CV_8UC3);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double v = 128;
            for (int k = 0; k < 5; k++)
                v += A[k] * vnoise(x / L[k], y / L[k], k, seed);
            long q = std::lrint(v);
            q = q < 0 ? 0 : (q > 255 ? 255 : q);
            unsigned char *p = src.data + src.step * (size_t)y + 3 * (size_t)x;
            p[0] = p[1] = p[2] = (unsigned char)q;
        }

    // one call
    std::vector<Keypoint> whole;
    Sift(src, whole, SIGMA, INTERVALS);
    CHECK(SiftLastStatus() == 0, "Sift");

    // the same through the stage entry points
    Mat gray, up, base, init_gray;
    ConvertToGray(src, gray);
    CHECK(SiftLastStatus() == 0, "ConvertToGray");
    UpSample(gray, up);
    CHECK(SiftLastStatus() == 0, "UpSample");
    double sigma_init = std::sqrt(SIGMA * SIGMA - (INIT_SIGMA * 2) * (INIT_SIGMA * 2));
    GaussianSmooth(up, base, sigma_init);
    CHECK(SiftLastStatus() == 0, "GaussianSmooth");
    CreateInitSmoothGray(src, init_gray, SIGMA);
    CHECK(SiftLastStatus() == 0, "CreateInitSmoothGray");
    CHECK(std::memcmp(base.data, init_gray.data, (size_t)(base.dataend - base.data)) == 0,
          "gray -> upsample -> blur equals CreateInitSmoothGray");
    int octaves = std::log((double)std::min(init_gray.rows, init_gray.cols)) / std::log(2.0) - 2;
    std::vector<Mat> gauss_pyr, dog_pyr;
    GaussianPyramid(init_gray, gauss_pyr, octaves, INTERVALS, SIGMA);
    CHECK(SiftLastStatus() == 0 && (int)gauss_pyr.size() == octaves * (INTERVALS + 3), "GaussianPyramid");
    Mat down;
    DownSample(gauss_pyr[INTERVALS], down);
    CHECK(down.rows == gauss_pyr[INTERVALS + 3].rows &&
              std::memcmp(down.data, gauss_pyr[INTERVALS + 3].data, (size_t)(down.dataend - down.data)) == 0,
          "DownSample of level S seeds the next octave");
    DogPyramid(gauss_pyr, dog_pyr, octaves, INTERVALS);
    CHECK(SiftLastStatus() == 0 && (int)dog_pyr.size() == octaves * (INTERVALS + 2), "DogPyramid");
    std::vector<Keypoint> extrema, features;
    DetectionLocalExtrema(dog_pyr, extrema, octaves, INTERVALS);
    CHECK(SiftLastStatus() == 0, "DetectionLocalExtrema");
    CalculateScale(extrema, SIGMA, INTERVALS);
    HalfFeatures(extrema);
    OrientationAssignment(extrema, features, gauss_pyr);
    CHECK(SiftLastStatus() == 0, "OrientationAssignment");
    DescriptorRepresentation(features, gauss_pyr, DESCR_HIST_BINS, DESCR_WINDOW_WIDTH);
    CHECK(SiftLastStatus() == 0, "DescriptorRepresentation");
    std::stable_sort(features.begin(), features.end(), [](const Keypoint &a, const Keypoint &b) { return a.scale < b.scale; });

    CHECK(features.size() == whole.size() && !whole.empty(), "feature count of the stage chain");
    for (size_t i = 0; i < whole.size(); i++) {
        const Keypoint &a = whole[i], &b = features[i];
        // CalculateScale runs on the host here and on the device inside Sift(), both through the same correctly
        // rounded 2^y: equal scales; the orientation sums may differ in their last bits
        bool same = a.x == b.x && a.y == b.y && a.octave == b.octave && a.interval == b.interval && a.dx == b.dx &&
                    a.dy == b.dy && a.scale == b.scale && std::fabs(a.ori - b.ori) <= 1e-9 &&
                    a.descr_length == b.descr_length &&
                    std::memcmp(a.descriptor, b.descriptor, sizeof a.descriptor) == 0;
        CHECK(same, "feature records of the stage chain");
    }
    std::printf("stage chain ok: %zu features, %d octaves\n", whole.size(), octaves);
    return 0;
}
