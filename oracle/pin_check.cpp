// pin_check.cpp -- TEST INFRASTRUCTURE ONLY (never part of the product library).
//
// The reference orders its output with std::sort on `scale` alone (sift/sift.cpp:1640-1643,
// 1703), which is unstable, so the order of exact-scale ties depends on the standard library.
// This tool runs the oracle on vnoise5(width,height,seed), orders the features with libstdc++'s
// std::sort exactly as the reference does, and writes them in the reference's text format
// (sift/sift.cpp:1910-1931) so that the file can be compared with a recorded checksum of the
// reference's own output on the same input.
//
//   usage: pin_check <width> <height> <seed> <out.txt>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sift_oracle.h"

static bool feature_cmp(so_keypoint &a, so_keypoint &b) { return a.scale < b.scale; }

int main(int argc, char **argv)
{
    if (argc != 5) {
        std::fprintf(stderr, "usage: %s width height seed out.txt\n", argv[0]);
        return 2;
    }
    int w = std::atoi(argv[1]), h = std::atoi(argv[2]);
    unsigned long long seed = std::strtoull(argv[3], nullptr, 10);
    std::vector<uint8_t> img((size_t)w * h * 3);
    so_vnoise5(img.data(), h, w, seed);
    size_t n = 0, nk = 0;
    so_counters ctr;
    so_keypoint *f = so_sift_debug(img.data(), h, w, 3, (size_t)w * 3, 1.6, 3, 0, &n, nullptr, &nk, nullptr, &ctr);
    std::vector<so_keypoint> v(f, f + n);
    so_free(f);
    std::sort(v.begin(), v.end(), feature_cmp);
    std::printf("cc00 %ld cc0 %ld cc1 %ld cc2 %ld cc3 %ld features %zu\n", ctr.cc00, ctr.cc0, ctr.cc1, ctr.cc2,
                ctr.cc3, n);
    return so_write_features(v.data(), n, argv[4]);
}
