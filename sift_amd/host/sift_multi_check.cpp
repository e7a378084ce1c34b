// sift_multi_check.cpp -- SiftBatchMultiGpu against Sift() frame by frame, and its rate:
//   check <w> <h> <frames> <gpus> <batch> <rehearsal 0|1>
//       frames of one size (value noise, every fifth a 3-pixel checkerboard, far denser than the default capacity)
//       through SiftBatchMultiGpu; every frame's packed features must equal what Sift() returns for that frame
//       (dx, dy, scale, ori rounded to float, the descriptor as bytes), in the same order
//   bench <w> <h> <frames> <gpus> <batch>
//       the rate of SiftBatchMultiGpu over host frames (a few distinct frames, repeated), one JSON line
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "sift_gpu.h"
#include "synth_image.h"

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "";
    if (mode == "check" && argc == 8) {
        const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), F = std::atoi(argv[4]);
        SiftMultiGpuOptions opt;
        opt.n_gpus = std::atoi(argv[5]);
        opt.batch = std::atoi(argv[6]);
        opt.rehearsal = std::atoi(argv[7]) != 0;
        std::vector<Mat> frames;
        for (int f = 0; f < F; f++)
            frames.push_back(f % 5 == 4 ? make_checker_image(w, h, 3) : make_image(w, h, 1000 + (uint64_t)f));
        SiftBatchFeatures got;
        std::string err;
        int rc = SiftBatchMultiGpu(frames, got, opt, &err);
        if (rc == 0) { // once more through the resources the first call left: same answer
            SiftBatchFeatures again;
            rc = SiftBatchMultiGpu(frames, again, opt, &err);
            if (rc == 0 && (again.first != got.first || again.features.size() != got.features.size() ||
                            std::memcmp(again.features.data(), got.features.data(),
                                        got.features.size() * sizeof(SiftPackedFeature)) != 0)) {
                std::fprintf(stderr, "FAILED: the second call returned other features than the first\n");
                return 1;
            }
        }
        if (rc != 0 || got.first.size() != frames.size() + 1) {
            std::fprintf(stderr, "FAILED: SiftBatchMultiGpu: status %d (%s)\n", rc, err.c_str());
            return 1;
        }
        size_t total = 0;
        for (int f = 0; f < F; f++) {
            std::vector<Keypoint> want;
            Sift(frames[(size_t)f], want);
            if (SiftLastStatus() != 0 || want.size() != got.count((size_t)f)) {
                std::fprintf(stderr, "FAILED: frame %d: %zu features, Sift() gives %zu (status %d)\n", f,
                             got.count((size_t)f), want.size(), SiftLastStatus());
                return 1;
            }
            for (size_t i = 0; i < want.size(); i++) {
                const SiftPackedFeature &g = got.frame((size_t)f)[i];
                const Keypoint &k = want[i];
                bool same = g.x == (float)k.dx && g.y == (float)k.dy && g.scale == (float)k.scale && g.ori == (float)k.ori;
                for (int q = 0; q < 128 && same; q++)
                    same = g.descriptor[q] == (unsigned char)k.descriptor[q];
                if (!same) {
                    std::fprintf(stderr, "FAILED: frame %d feature %zu differs from Sift()'s\n", f, i);
                    return 1;
                }
            }
            total += want.size();
        }
        SiftMultiGpuRelease();
        std::printf("multi ok: %d frames, %zu features\n", F, total);
        return 0;
    }
    if (mode == "bench" && argc == 7) {
        const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), F = std::atoi(argv[4]);
        SiftMultiGpuOptions opt;
        opt.n_gpus = std::atoi(argv[5]);
        opt.batch = std::atoi(argv[6]);
        std::vector<Mat> distinct, frames;
        for (int f = 0; f < 4; f++)
            distinct.push_back(make_image(w, h, 1000 + (uint64_t)f));
        for (int f = 0; f < F; f++)
            frames.push_back(distinct[(size_t)f % distinct.size()]); // shared pixels: the frames are only read
        SiftBatchFeatures got;
        std::string err;
        double best = 0;
        size_t feats = 0;
        for (int rep = 0; rep < 4; rep++) { // the first run creates the contexts (kept between calls); the best is reported
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = SiftBatchMultiGpu(frames, got, opt, &err);
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rc != 0) {
                std::fprintf(stderr, "FAILED: status %d (%s)\n", rc, err.c_str());
                return 1;
            }
            feats = got.features.size();
            if (F / sec > best)
                best = F / sec;
        }
        std::printf("{\"images_per_s\": %.1f, \"frames\": %d, \"n_gpus\": %d, \"batch\": %d, \"features\": %zu, "
                    "\"what\": \"SiftBatchMultiGpu: host frames in, every frame's packed features gathered on every GPU "
                    "and returned to the host as one array; contexts kept between calls\"}\n",
                    best, F, opt.n_gpus, opt.batch, feats);
        return 0;
    }
    std::fprintf(stderr, "usage: %s check w h frames gpus batch rehearsal | bench w h frames gpus batch\n", argv[0]);
    return 2;
}
