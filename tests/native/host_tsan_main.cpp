// host_tsan_main.cpp -- the shared batching contexts of the host layer under ThreadSanitizer (built by
// `make -C sift_amd/host SAN=1` against tests/native/fake_siftgpu.cpp; run by tests/test_cpu_sanitizers.py).
//   host_tsan_check <threads> <calls> <batch>
// Threads call Sift() and Sift_nointerval() through SiftSetBatching(batch): frames of two sizes, both detectors, one
// frame that needs a grown context, and one thread whose frame makes every batch it is in fail.  Every call of the
// other threads must succeed with the records a context of the thread's own returns; the offender's calls must fail;
// interleaved detectors must not rebuild a context per batch.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "sift_gpu.h"
#include "synth_image.h"

extern "C" int fake_sg_creates(void);
extern "C" int fake_sg_live(void);
extern "C" int fake_sg_max_batch(void);

static bool same(const std::vector<Keypoint> &a, const std::vector<Keypoint> &b)
{
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(Keypoint)) == 0);
}

int main(int argc, char **argv)
{
    const int T = argc > 1 ? std::atoi(argv[1]) : 8, N = argc > 2 ? std::atoi(argv[2]) : 20;
    const int batch = argc > 3 ? std::atoi(argv[3]) : 4;
    struct Job {
        Mat img;
        bool nointerval = false, offender = false;
        std::vector<Keypoint> want;
    };
    std::vector<Job> jobs((size_t)T);
    for (int t = 0; t < T; t++) {
        Job &j = jobs[(size_t)t];
        j.img = t % 5 == 3 ? make_image(41, 27, 70 + (uint64_t)t) : make_image(80, 60, 100 + (uint64_t)t);
        j.nointerval = t % 2 == 1; // the two detectors alternate: two contexts per lane, kept
        if (t == 2)
            j.img.data[0] = 0xDD;
        if (t == 4) {
            j.img.data[0] = 0xEE;
            j.offender = true;
        }
        if (j.nointerval)
            Sift_nointerval(j.img, j.want);
        else
            Sift(j.img, j.want);
        if ((SiftLastStatus() != 0) != j.offender) {
            std::fprintf(stderr, "FAILED: set-up call of thread %d: status %d\n", t, SiftLastStatus());
            return 1;
        }
    }
    SiftReleaseContext();
    const int creates_before = fake_sg_creates();
    SiftSetBatching(batch);
    std::atomic<int> bad(0);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            const Job &j = jobs[(size_t)t];
            for (int i = 0; i < N; i++) {
                std::vector<Keypoint> got;
                if (j.nointerval)
                    Sift_nointerval(j.img, got);
                else
                    Sift(j.img, got);
                const bool failed = SiftLastStatus() != 0;
                if (failed != j.offender || (!failed && !same(j.want, got)) || (failed && !got.empty())) {
                    std::fprintf(stderr, "thread %d call %d: status %d (%s), %zu records, %zu expected\n", t, i,
                                 SiftLastStatus(), SiftLastError(), got.size(), j.want.size());
                    bad++;
                    return;
                }
            }
        });
    for (std::thread &x : th)
        x.join();
    const int creates = fake_sg_creates() - creates_before;
    SiftSetBatching(0);
    if (bad) {
        std::fprintf(stderr, "FAILED: %d threads saw a wrong result\n", bad.load());
        return 1;
    }
    // two lanes x (two detectors + one grown context for the dense frame) and some slack: far fewer than one per batch
    if (creates > 16) {
        std::fprintf(stderr, "FAILED: %d contexts were created for %d calls (a context per batch?)\n", creates, T * N);
        return 1;
    }
    if (fake_sg_live() != 0) {
        std::fprintf(stderr, "FAILED: %d contexts were not handed back\n", fake_sg_live());
        return 1;
    }
    std::printf("host thread-sanitizer check ok: %d threads x %d calls, %d contexts created, largest batch %d\n", T, N,
                creates, fake_sg_max_batch());
    return 0;
}
