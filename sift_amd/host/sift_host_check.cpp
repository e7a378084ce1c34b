// sift_host_check.cpp -- checks of the host layer that need no reference output:
//   dense <w> <h> <cell> <out.txt>   Sift() on a checkerboard of <cell>-pixel squares, whose features overflow the
//                                    context's default sizing: the layer must grow the context and return them all
//   views <w> <h> <seed>             Sift() on a 4-channel frame and on a padded view (step > cols * channels, last
//                                    row ending at the end of its buffer) equals Sift() on the packed 3-channel frame
//   legacy                           the reference's off-path helpers (8-bit legacy blurs, dump names)
//   shared <w> <h> <threads> <calls> <batch>
//                                    threads calling Sift()/Sift_nointerval() through one shared context
//                                    (SiftSetBatching) get the records a context of their own returns: frames of two
//                                    sizes, both detectors and one frame denser than the default sizing in the mix
//   precision <w> <h> <seed>         SiftSetPrecision: SIFT_FUSED and SIFT_SINGLE return about the features of SIFT_EXACT
//                                    (same positions for nearly all), and SIFT_EXACT afterwards the very same records
//                                    as before, per thread and through the shared contexts
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "sift_gpu.h"
#include "synth_image.h"

static bool same_features(const std::vector<Keypoint> &a, const std::vector<Keypoint> &b)
{
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(Keypoint)) == 0);
}

static int fail(const char *what)
{
    std::fprintf(stderr, "FAILED: %s (%s)\n", what, SiftLastError());
    return 1;
}

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "";
    if (mode == "dense" && argc == 6) {
        Mat img = make_checker_image(std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]));
        std::vector<Keypoint> f;
        Sift(img, f, SIGMA);
        if (SiftLastStatus() != 0)
            return fail("Sift on a dense frame");
        write_features(f, argv[5]);
        std::vector<Keypoint> again; // the grown context serves the next call directly
        Sift(img, again, SIGMA);
        if (SiftLastStatus() != 0 || !same_features(f, again))
            return fail("second Sift on the dense frame");
        std::printf("dense ok: %zu features\n", f.size());
        return 0;
    }
    if (mode == "views" && argc == 5) {
        const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
        Mat img = make_image(w, h, std::strtoull(argv[4], nullptr, 10));
        std::vector<Keypoint> want, got4, gotv;
        Sift(img, want, SIGMA);
        if (SiftLastStatus() != 0 || want.empty())
            return fail("Sift");
        Mat four(h, w, CV_8UC4); // b, g, r and a fourth channel the path must ignore
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const unsigned char *s = img.data + img.step * (size_t)y + 3 * (size_t)x;
                unsigned char *d = four.data + four.step * (size_t)y + 4 * (size_t)x;
                d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = (unsigned char)(x * 7 + y);
            }
        Sift(four, got4, SIGMA);
        if (SiftLastStatus() != 0 || !same_features(want, got4))
            return fail("4-channel frame");
        // a view into a wider parent whose last row ends exactly at the end of the allocation
        const size_t step = (size_t)3 * w + 40;
        std::vector<unsigned char> parent((size_t)(h - 1) * step + (size_t)3 * w);
        for (int y = 0; y < h; y++)
            std::memcpy(parent.data() + step * (size_t)y, img.data + img.step * (size_t)y, (size_t)3 * w);
        Mat view(h, w, CV_8UC3, parent.data(), step);
        Sift(view, gotv, SIGMA);
        if (SiftLastStatus() != 0 || !same_features(want, gotv))
            return fail("padded view");
        std::printf("views ok: %zu features\n", want.size());
        return 0;
    }
    if (mode == "precision" && argc == 5) {
        const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
        Mat img = make_image(w, h, std::strtoull(argv[4], nullptr, 10));
        auto positions_shared = [](const std::vector<Keypoint> &a, const std::vector<Keypoint> &b) {
            size_t n = 0, j = 0; // both lists are ordered by scale; count records of a found in b near their rank
            for (size_t i = 0; i < a.size(); i++) {
                const size_t lo = j > 64 ? j - 64 : 0, hi = j + 64 < b.size() ? j + 64 : b.size();
                for (size_t k = lo; k < hi; k++)
                    if (a[i].x == b[k].x && a[i].y == b[k].y && a[i].octave == b[k].octave && a[i].interval == b[k].interval) {
                        n++;
                        j = k;
                        break;
                    }
                j++;
            }
            return n;
        };
        for (int batch = 0; batch <= 4; batch += 4) {
            SiftSetBatching(batch);
            std::vector<Keypoint> exact, fused, single, again;
            SiftSetPrecision(SIFT_EXACT);
            Sift(img, exact, SIGMA);
            if (SiftLastStatus() != 0 || exact.size() < 100)
                return fail("Sift, exact");
            SiftSetPrecision(SIFT_FUSED);
            Sift(img, fused, SIGMA);
            if (SiftLastStatus() != 0)
                return fail("Sift, fused");
            SiftSetPrecision(SIFT_SINGLE);
            Sift(img, single, SIGMA);
            if (SiftLastStatus() != 0)
                return fail("Sift, single");
            SiftSetPrecision(SIFT_EXACT);
            Sift(img, again, SIGMA);
            if (SiftLastStatus() != 0 || !same_features(exact, again))
                return fail("exact after the other modes");
            const size_t nf = positions_shared(exact, fused), ns = positions_shared(exact, single);
            std::printf("precision (batching %d): exact %zu, fused %zu (%zu shared), single %zu (%zu shared)\n", batch,
                        exact.size(), fused.size(), nf, single.size(), ns);
            if (nf * 1000 < exact.size() * 990 || ns * 100 < exact.size() * 90)
                return fail("the modes disagree beyond what their arithmetic explains");
        }
        SiftSetBatching(0);
        std::printf("precision ok\n");
        return 0;
    }
    if (mode == "legacy") {
        Mat g(40, 50, CV_8UC1), out, out2, c3(20, 20, CV_8UC3), out3;
        for (int y = 0; y < 40; y++)
            for (int x = 0; x < 50; x++)
                g.data[g.step * (size_t)y + x] = 100;
        std::memset(c3.data, 60, (size_t)(c3.dataend - c3.data));
        GaussianTemplateSmooth(g, out);
        GaussianSmooth2D(g, out2, 1.0);
        GaussianTemplateSmooth(c3, out3);
        // a constant image stays constant up to the truncation of the (not exactly normalised) sums, where a window
        // centre reaches: rows 3..rows-5, columns 3..cols-5 for the 7x7 kernels
        const int a = out.data[out.step * 10 + 10], b = out2.data[out2.step * 10 + 10], c = out3.data[out3.step * 8 + 3 * 8 + 1];
        if (out.rows != 40 || out.cols != 50 || a < 99 || a > 100 || b < 99 || b > 100 || c < 59 || c > 60)
            return fail("legacy blurs");
        Mat same;
        GaussianSmooth2D(g, same, 0.1); // kernel size 1: a copy
        if (same.rows != 40 || std::memcmp(same.data, g.data, 40 * 50) != 0)
            return fail("GaussianSmooth2D with a 1x1 kernel");
        if (std::string(GetFileName("pyr/level", 7)) != "pyr/level7.jpg")
            return fail("GetFileName");
        std::vector<Mat> pyr(2);
        display_pyr(pyr, "dog");
        testInverse3D();
        std::printf("legacy ok\n");
        return 0;
    }
    if (mode == "shared" && argc == 7) {
        const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), T = std::atoi(argv[4]), N = std::atoi(argv[5]);
        const int batch = std::atoi(argv[6]);
        struct Job {
            Mat img;
            bool nointerval;
            std::vector<Keypoint> want;
        };
        std::vector<Job> jobs((size_t)T);
        for (int t = 0; t < T; t++) {
            Job &j = jobs[(size_t)t];
            if (t % 5 == 3)
                j.img = make_image(w / 2 + 1, h / 2 - 1, 77 + (uint64_t)t); // a second, odd size
            else if (t == 6)
                j.img = make_checker_image(w, h, 3); // overflows the default sizing: the shared context must grow
            else
                j.img = make_image(w, h, 1000 + (uint64_t)t);
            j.nointerval = t % 4 == 1;
            if (j.nointerval)
                Sift_nointerval(j.img, j.want, SIGMA);
            else
                Sift(j.img, j.want, SIGMA);
            if (SiftLastStatus() != 0 || j.want.empty())
                return fail("Sift with a context of the thread's own");
        }
        SiftReleaseContext();
        SiftSetBatching(batch);
        std::atomic<int> bad(0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                const Job &j = jobs[(size_t)t];
                for (int i = 0; i < N; i++) {
                    std::vector<Keypoint> got;
                    if (j.nointerval)
                        Sift_nointerval(j.img, got, SIGMA);
                    else
                        Sift(j.img, got, SIGMA);
                    if (SiftLastStatus() != 0 || !same_features(j.want, got)) {
                        std::fprintf(stderr, "thread %d call %d: status %d (%s), %zu features, %zu expected\n", t, i,
                                     SiftLastStatus(), SiftLastError(), got.size(), j.want.size());
                        bad++;
                        return;
                    }
                }
            });
        for (std::thread &x : th)
            x.join();
        SiftSetBatching(0);
        if (bad)
            return fail("shared context");
        std::printf("shared ok: %d threads x %d calls\n", T, N);
        return 0;
    }
    std::fprintf(stderr, "usage: %s dense w h cell out.txt | views w h seed | precision w h seed | legacy | shared w h threads calls batch\n", argv[0]);
    return 2;
}
