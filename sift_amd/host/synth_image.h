// synth_image.h -- synthetic test images for the host programs (the reference's demo reads two JPEG files that are
// not part of its repository): the 5-octave value-noise frame every parity number of this repository is quoted on,
// and a checkerboard of 3-pixel cells, far denser in features (four orientations at every crossing) than any natural
// image.
#pragma once

#include <cmath>
#include <cstdint>

#include "sift_gpu.h"

static inline uint64_t sm64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static inline double lat(int ix, int iy, int k, uint64_t seed)
{
    uint64_t c = (uint64_t)(uint32_t)ix | ((uint64_t)(uint32_t)iy << 32);
    uint64_t h = sm64(seed ^ sm64(c) ^ sm64(0x1000 + (uint64_t)k));
    return ((double)(h >> 11) * (1.0 / 9007199254740992.0)) * 2 - 1;
}
static inline double vnoise(double x, double y, int k, uint64_t seed)
{
    double fx0 = std::floor(x), fy0 = std::floor(y);
    int ix = (int)fx0, iy = (int)fy0;
    double fx = x - fx0, fy = y - fy0;
    fx = fx * fx * (3 - 2 * fx);
    fy = fy * fy * (3 - 2 * fy);
    double a = lat(ix, iy, k, seed), b = lat(ix + 1, iy, k, seed), c = lat(ix, iy + 1, k, seed), d = lat(ix + 1, iy + 1, k, seed);
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy;
}
static inline Mat make_image(int w, int h, uint64_t seed)
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

// saturated checkerboard with `cell`-pixel squares; b = g = r
static inline Mat make_checker_image(int w, int h, int cell)
{
    Mat img(h, w, CV_8UC3);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const unsigned char q = ((x / cell + y / cell) % 2 == 0) ? 0 : 255;
            unsigned char *p = img.data + img.step * (size_t)y + 3 * (size_t)x;
            p[0] = p[1] = p[2] = q;
        }
    return img;
}
