// sift_demo.cpp -- the reference's demo flow (sift/源.cpp:13-34) without the GUI: two images, Sift() on the
// first and Sift_nointerval() on the second, descriptors written in the reference's text format.
// Images are 5-octave value-noise frames generated here (the reference reads two JPEG files that are
// not part of its repository).   usage: sift_demo <width> <height> <seed1> <seed2> <out1.txt> <out2.txt>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>

#include "sift_gpu.h"
#include "synth_image.h"

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
