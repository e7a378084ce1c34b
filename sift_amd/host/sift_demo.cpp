// sift_demo.cpp -- the reference's demo flow (sift/源.cpp:13-34) without the GUI: two images, Sift() on the
// first and Sift_nointerval() on the second, descriptors written in the reference's text format.
// Images are 5-octave value-noise frames generated here (the reference reads two JPEG files that are
// not part of its repository).   usage: sift_demo <width> <height> <seed1> <seed2> <out1.txt> <out2.txt>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>

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
static Mat make_image(int w, int h, uint64_t seed)
{
    static const double L[5] = {4, 8, 16, 32, 64}, A[5] = {20, 28, 36, 40, 40};
    Mat img(h, w, CV_8UC3);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double v = 128;
            for (int k = 0; k < 5; k++)
                v += A[k] * vnoise(x / L[k], y / L[k], k, seed);
            long q = std::lrint(v);
            q = q < 0 ? 0 : (q > 255 ? 255 : q);
            unsigned char *p = img.data + img.step * (size_t)y + 3 * (size_t)x;
            p[0] = p[1] = p[2] = (unsigned char)q;
        }
    return img;
}

int main(int argc, char **argv)
{
    if (argc != 7 && argc != 8) {
        std::fprintf(stderr, "usage: %s width height seed1 seed2 out1.txt out2.txt [matches.txt]\n", argv[0]);
        return 2;
    }
    int w = std::atoi(argv[1]), h = std::atoi(argv[2]);
    Mat image01 = make_image(w, h, std::strtoull(argv[3], nullptr, 10));
    Mat image02 = make_image(w, h, std::strtoull(argv[4], nullptr, 10));

    std::vector<Keypoint> features1, features2;
    Sift(image01, features1, SIGMA);
    if (SiftLastStatus() != 0) {
        std::fprintf(stderr, "Sift failed: %s\n", SiftLastError());
        return 1;
    }
    Sift_nointerval(image02, features2, SIGMA);
    if (SiftLastStatus() != 0) {
        std::fprintf(stderr, "Sift_nointerval failed: %s\n", SiftLastError());
        return 1;
    }
    write_features(features1, argv[5]);
    write_features(features2, argv[6]);

    // round trip through the text format
    std::vector<Keypoint> back;
    read_features(back, argv[5]);
    if (back.size() != features1.size()) {
        std::fprintf(stderr, "read_features returned %zu of %zu records\n", back.size(), features1.size());
        return 1;
    }
    std::printf("features1 %zu features2 %zu\n", features1.size(), features2.size());

    // the matching loop of the reference's main (sift/源.cpp:88-163)
    std::vector<MatchPoint> v_matchpoint;
    MatchFeatures(features1, features2, v_matchpoint);
    if (SiftLastStatus() != 0) {
        std::fprintf(stderr, "MatchFeatures failed: %s\n", SiftLastError());
        return 1;
    }
    std::printf("match number: %zu\n", v_matchpoint.size());
    if (argc == 8) {
        std::FILE *f = std::fopen(argv[7], "w");
        if (!f)
            return 1;
        for (const MatchPoint &m : v_matchpoint)
            std::fprintf(f, "%d %d\n", m.index_1, m.index_2);
        std::fclose(f);
    }
    return 0;
}
