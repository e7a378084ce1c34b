// host_san_main.cpp -- the GPU-free host code under AddressSanitizer / UndefinedBehaviorSanitizer (built by
// `make -C sift_amd/host SAN=1` against tests/native/fake_siftgpu.cpp; run by tests/test_cpu_sanitizers.py):
// text feature files, drawing and pyramid dumps, the legacy 8-bit blurs, wire-record expansion, and Sift() through
// the per-thread context and the shared batching contexts (context growth, a failing image, odd image sizes).
//   host_san_check <scratch dir>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "siftgpu.h"
#include "sift_gpu.h"
#include "synth_image.h"

extern "C" int fake_sg_live(void);
extern "C" int fake_sg_creates(void);

#define CHECK(cond)                                                                     \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            std::fprintf(stderr, "FAILED line %d: %s (%s)\n", __LINE__, #cond, SiftLastError()); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

static bool same(const std::vector<Keypoint> &a, const std::vector<Keypoint> &b)
{
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(Keypoint)) == 0);
}

int main(int argc, char **argv)
{
    if (argc < 2)
        return 2;
    const std::string dir = std::string(argv[1]) + "/";

    // ---- wire expansion ----
    {
        std::vector<sg_wire_feature> w(37);
        for (size_t i = 0; i < w.size(); i++) {
            std::memset(&w[i], 0, sizeof w[i]);
            w[i].x = (int)i;
            w[i].scale = 1.5 * (double)i;
            for (int q = 0; q < 128; q++)
                w[i].desc[q] = (uint8_t)(i + (size_t)q);
        }
        std::vector<Keypoint> k(w.size());
        sg_expand_wire(w.data(), w.size(), reinterpret_cast<sg_keypoint *>(k.data()));
        CHECK(k[36].x == 36 && k[36].descr_length == 128 && k[36].descriptor[127] == (double)(uint8_t)(36 + 127));
        sg_expand_wire(w.data(), 0, reinterpret_cast<sg_keypoint *>(k.data())); // nothing to do
    }

    // ---- text feature files: what is written is read back ----
    {
        std::vector<Keypoint> f(5), g;
        for (size_t i = 0; i < f.size(); i++) {
            f[i] = Keypoint();
            f[i].dx = 10.25 + (double)i;
            f[i].dy = 3.5 * (double)i;
            f[i].scale = 1.0 + (double)i;
            f[i].ori = 0.125 * (double)i - 0.25;
            f[i].descr_length = FEATURE_ELEMENT_LENGTH;
            for (int q = 0; q < FEATURE_ELEMENT_LENGTH; q++)
                f[i].descriptor[q] = (double)((int)(i * 7 + (size_t)q) % 256);
        }
        write_features(f, (dir + "feat.txt").c_str());
        read_features(g, (dir + "feat.txt").c_str());
        CHECK(g.size() == f.size());
        for (size_t i = 0; i < f.size(); i++) {
            CHECK(std::fabs(g[i].dx - f[i].dx) < 1e-4 && std::fabs(g[i].scale - f[i].scale) < 1e-4);
            for (int q = 0; q < FEATURE_ELEMENT_LENGTH; q++)
                CHECK(g[i].descriptor[q] == f[i].descriptor[q]);
        }
        std::vector<Keypoint> none;
        read_features(none, (dir + "no_such_file.txt").c_str()); // a missing file leaves the vector as it is
        CHECK(none.empty());
        write_features(none, (dir + "empty.txt").c_str());
        read_features(none, (dir + "empty.txt").c_str());
        CHECK(none.empty());
    }

    // ---- drawing (clipping at every border) and pyramid dumps ----
    {
        Mat canvas(33, 47, CV_8UC3);
        std::memset(canvas.data, 0, (size_t)(canvas.dataend - canvas.data));
        std::vector<Keypoint> k(6);
        const double xs[6] = {-4.0, 0.0, 46.0, 50.0, 20.0, 23.5}, ys[6] = {-3.0, 32.0, 0.0, 40.0, 16.0, 31.9};
        for (int i = 0; i < 6; i++) {
            k[(size_t)i] = Keypoint();
            k[(size_t)i].dx = xs[i];
            k[(size_t)i].dy = ys[i];
            k[(size_t)i].scale = 2.0 + i * 3.0; // arrows long enough to leave the image
            k[(size_t)i].ori = -3.0 + i;
        }
        DrawSiftFeatures(canvas, k);
        DrawKeyPoints(canvas, k);
        DrawKeyPointsRight(canvas, k, cv::Size(30, 0));
        DrawSiftFeature(canvas, k[4], cv::Scalar(1, 2, 3));
        Mat a(5, 7, CV_64FC1), flat(4, 4, CV_64FC1), out8;
        for (int y = 0; y < a.rows; y++)
            for (int x = 0; x < a.cols; x++)
                reinterpret_cast<double *>(a.data + a.step * (size_t)y)[x] = std::sin(0.7 * x) + 0.1 * y;
        for (int i = 0; i < 16; i++)
            reinterpret_cast<double *>(flat.data)[i] = 0.5;
        cv64f_to_cv8U(a, out8);
        CHECK(out8.rows == 5 && out8.cols == 7);
        cv64f_to_cv8U(flat, out8); // max == min
        writecv64f((dir + "a.pgm").c_str(), a);
        std::vector<Mat> pyr;
        pyr.push_back(a);
        pyr.push_back(flat);
        write_pyr(pyr, (dir + "lvl").c_str());
        display_pyr(pyr, "dog");
        CHECK(std::string(GetFileName("x/", 12)) == "x/12.jpg");
    }

    // ---- the legacy 8-bit blurs (sizes around their 7x7 kernels) ----
    {
        const int sizes[][2] = {{1, 1}, {3, 9}, {7, 7}, {8, 6}, {40, 50}};
        for (const auto &sz : sizes) {
            Mat g(sz[0], sz[1], CV_8UC1), c3(sz[0], sz[1], CV_8UC3), o1, o2, o3;
            std::memset(g.data, 90, (size_t)(g.dataend - g.data));
            std::memset(c3.data, 60, (size_t)(c3.dataend - c3.data));
            GaussianTemplateSmooth(g, o1);
            GaussianSmooth2D(g, o2, 1.0);
            GaussianSmooth2D(g, o2, 0.1);
            GaussianTemplateSmooth(c3, o3);
        }
        testInverse3D();
    }

    // ---- Sift() through the fake library: a context per thread, then the shared batching contexts ----
    {
        Mat a = make_image(97, 61, 5), b = make_image(33, 20, 6), dense = make_image(64, 48, 7), bad = make_image(64, 48, 8);
        dense.data[0] = 0xDD; // needs a larger context: the layer grows it and runs again
        bad.data[0] = 0xEE;   // fails whatever the context
        // a padded view whose last row ends with its buffer
        const size_t step = (size_t)3 * 97 + 11;
        std::vector<unsigned char> parent((size_t)60 * step + (size_t)3 * 97);
        for (int y = 0; y < 61; y++)
            std::memcpy(parent.data() + step * (size_t)y, a.data + a.step * (size_t)y, (size_t)3 * 97);
        Mat view(61, 97, CV_8UC3, parent.data(), step);
        std::vector<Keypoint> fa, fb, fd, fv, fni, fbad, fa2, fd2;
        Sift(a, fa);
        CHECK(SiftLastStatus() == 0 && fa.size() >= 16);
        Sift(view, fv);
        CHECK(SiftLastStatus() == 0 && same(fa, fv));
        Sift(b, fb);
        CHECK(SiftLastStatus() == 0 && !fb.empty());
        Sift_nointerval(a, fni);
        CHECK(SiftLastStatus() == 0 && fni.size() >= 16 && !same(fa, fni));
        Sift(dense, fd);
        CHECK(SiftLastStatus() == 0 && fd.size() > 4000);
        Sift(bad, fbad);
        CHECK(SiftLastStatus() != 0 && fbad.empty());
        Sift(a, fa, SIGMA, 4); // intervals != 3: refused, nothing appended
        CHECK(SiftLastStatus() != 0 && fa.size() == fv.size());
        Mat gray(10, 10, CV_8UC1);
        std::memset(gray.data, 1, 100);
        Sift(gray, fbad); // one channel: refused
        CHECK(SiftLastStatus() != 0 && fbad.empty());
        SiftReleaseContext();
        SiftSetBatching(4);
        Sift(a, fa2);
        CHECK(SiftLastStatus() == 0 && same(fa, fa2));
        Sift(dense, fd2);
        CHECK(SiftLastStatus() == 0 && same(fd, fd2));
        Sift(bad, fbad);
        CHECK(SiftLastStatus() != 0 && fbad.empty());
        fa2.clear();
        Sift(view, fa2); // appended behind what the vector holds
        Sift(view, fa2);
        CHECK(fa2.size() == 2 * fa.size() && std::memcmp(&fa2[fa.size()], fa.data(), fa.size() * sizeof(Keypoint)) == 0);
        // another arithmetic: the shared contexts go, the next call builds new ones (the stand-in's records do not
        // depend on the precision)
        const int creates = fake_sg_creates();
        SiftSetPrecision(SIFT_FUSED);
        fa2.clear();
        Sift(a, fa2);
        CHECK(SiftLastStatus() == 0 && same(fa, fa2) && fake_sg_creates() > creates);
        SiftSetBatching(0);
        CHECK(fake_sg_live() == 0); // every context was handed back
        const int creates2 = fake_sg_creates();
        fa2.clear();
        Sift(a, fa2); // a context of this thread, built with the fused arithmetic
        SiftSetPrecision(SIFT_EXACT);
        Sift(a, fa2); // ... and rebuilt for the exact one
        CHECK(SiftLastStatus() == 0 && fa2.size() == 2 * fa.size() && fake_sg_creates() == creates2 + 2);
        SiftReleaseContext();
        CHECK(fake_sg_live() == 0);
    }
    std::printf("host sanitizer check ok\n");
    return 0;
}
