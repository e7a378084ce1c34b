// fake_siftgpu.cpp -- a stand-in for libsiftgpu used only by the sanitizer builds of the host layer (tests/): it
// implements the part of the C ABI (include/siftgpu.h) that Sift()/Sift_nointerval() of the host layer call, with
// records that are a deterministic function of the image bytes and the parameters, so that the host code (context
// management, the shared batching contexts, wire expansion) can run under AddressSanitizer / ThreadSanitizer on a
// machine without a GPU.  It computes nothing of SIFT and is never linked into the product.
//
// Behaviours the tests steer through the first byte of an image:
//   0xEE  "offender": any batch that contains it fails with SG_E_HIP
//   0xDD  "dense": needs a context with max_features >= kDenseNeed (SG_E_CAPACITY + sg_required_features otherwise)
// A context is not thread-safe (as in the real library): the fake touches plain context fields in every call, so a
// host layer that let two threads into one context at a time is reported by ThreadSanitizer.
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "siftgpu.h"

namespace {
constexpr size_t kDenseNeed = 5000;
std::atomic<int> g_creates(0), g_live(0), g_max_live(0), g_batches(0), g_max_batch_seen(0);

uint64_t mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
} // namespace

struct sg_ctx {
    sg_params prm;
    std::string err;
    size_t need = 0;
    int last_octaves = 0;
    unsigned long long calls = 0; // plain field, written by every call: races show under ThreadSanitizer
    std::vector<std::vector<sg_wire_feature>> pending;
    std::vector<sg_counters> counters;
    int pieces = 0;
};

extern "C" {

void sg_default_params(sg_params *p)
{
    memset(p, 0, sizeof *p);
    p->sigma = 1.6;
    p->intervals = 3;
    p->max_width = 1920;
    p->max_height = 1080;
    p->max_batch = 1;
}

const char *sg_strerror(int status)
{
    switch (status) {
    case SG_OK: return "ok";
    case SG_E_INVALID: return "invalid argument";
    case SG_E_HIP: return "HIP error";
    case SG_E_CAPACITY: return "capacity exceeded";
    case SG_E_UNSUPPORTED: return "unsupported";
    case SG_E_NOMEM: return "out of memory";
    }
    return "unknown";
}

const char *sg_last_error(const sg_ctx *c) { return c ? c->err.c_str() : ""; }

int sg_create(const sg_params *p, sg_ctx **out)
{
    if (!p || !out || p->max_batch < 1)
        return SG_E_INVALID;
    sg_ctx *c = new sg_ctx;
    c->prm = *p;
    g_creates++;
    const int live = ++g_live;
    int m = g_max_live.load();
    while (live > m && !g_max_live.compare_exchange_weak(m, live)) {
    }
    *out = c;
    return SG_OK;
}

int sg_destroy(sg_ctx *c)
{
    if (c) {
        g_live--;
        delete c;
    }
    return SG_OK;
}

int sg_required_features(const sg_ctx *c, size_t *n)
{
    if (!c || !n)
        return SG_E_INVALID;
    *n = c->need;
    return SG_OK;
}

int sg_last_octaves(const sg_ctx *c) { return c ? c->last_octaves : 0; }

int sg_read_counters(sg_ctx *c, int image, sg_counters *out)
{
    if (!c || !out || image < 0 || (size_t)image >= c->counters.size())
        return SG_E_INVALID;
    *out = c->counters[(size_t)image];
    return SG_OK;
}

int sg_host_alloc(size_t bytes, void **ptr)
{
    if (!ptr || !bytes)
        return SG_E_INVALID;
    *ptr = malloc(bytes);
    return *ptr ? SG_OK : SG_E_NOMEM;
}

int sg_host_free(void *ptr)
{
    free(ptr);
    return SG_OK;
}

int sg_extract_batch_begin(sg_ctx *c, const uint8_t *const *imgs, int n_img, int w, int h, int ch, size_t stride, int,
                           uint32_t *counts)
{
    if (!c || !imgs || !counts || n_img < 1 || n_img > c->prm.max_batch || w < 1 || h < 1 || ch < 3)
        return SG_E_INVALID;
    if (w > c->prm.max_width || h > c->prm.max_height) {
        c->err = "image larger than the context";
        return SG_E_CAPACITY;
    }
    c->calls++;
    g_batches++;
    int m = g_max_batch_seen.load();
    while (n_img > m && !g_max_batch_seen.compare_exchange_weak(m, n_img)) {
    }
    c->pending.clear();
    c->counters.clear();
    for (int i = 0; i < n_img; i++) {
        const uint8_t first = imgs[i][0];
        if (first == 0xEE) {
            c->err = "the offending image";
            return SG_E_HIP;
        }
        if (first == 0xDD && (size_t)c->prm.max_features < kDenseNeed) {
            c->need = kDenseNeed;
            c->err = "capacity exceeded";
            return SG_E_CAPACITY;
        }
    }
    c->last_octaves = 7;
    for (int i = 0; i < n_img; i++) {
        // every byte of the image is read (a slot released too early or a short copy shows under the sanitizers)
        uint64_t hsh = mix((uint64_t)w * 65537u + (uint64_t)h) ^ mix((uint64_t)c->prm.nointerval + 17) ^
                       mix((uint64_t)(c->prm.sigma * 1e6));
        for (int y = 0; y < h; y++) {
            const uint8_t *row = imgs[i] + (size_t)y * stride;
            uint64_t r = 0;
            for (int x = 0; x < w * ch; x++)
                r = r * 131 + row[x];
            hsh = mix(hsh ^ r);
        }
        const size_t n = imgs[i][0] == 0xDD ? kDenseNeed - 7 : 16 + (size_t)(hsh % 48);
        std::vector<sg_wire_feature> recs(n);
        for (size_t k = 0; k < n; k++) {
            sg_wire_feature &f = recs[k];
            memset(&f, 0, sizeof f);
            const uint64_t a = mix(hsh + k);
            f.octave = (int32_t)(a % 7);
            f.interval = 1 + (int32_t)((a >> 8) % 3);
            f.x = (int32_t)((a >> 16) % (uint64_t)w);
            f.y = (int32_t)((a >> 32) % (uint64_t)h);
            f.scale = 1.0 + (double)k * 0.25;
            f.dx = f.x * 0.5;
            f.dy = f.y * 0.5;
            f.ori = ((double)(a % 6283)) / 1000.0 - 3.1415;
            f.val = 0.01 * (double)(a % 100);
            for (int q = 0; q < 128; q++)
                f.desc[q] = (uint8_t)(mix(a + (uint64_t)q) % 160);
        }
        counts[i] = (uint32_t)n;
        sg_counters ctr = {(unsigned long long)w * h, n * 3, n * 2, n + 1, n};
        c->counters.push_back(ctr);
        c->pending.push_back(std::move(recs));
    }
    return SG_OK;
}

int sg_extract_batch_fetch_wire(sg_ctx *c, sg_wire_feature *out, size_t cap_total)
{
    if (!c || c->pending.empty())
        return SG_E_INVALID;
    c->calls++;
    size_t total = 0;
    for (const auto &v : c->pending)
        total += v.size();
    if (total > cap_total || (total && !out))
        return SG_E_CAPACITY;
    size_t at = 0;
    for (const auto &v : c->pending) {
        if (!v.empty())
            memcpy(out + at, v.data(), v.size() * sizeof(sg_wire_feature));
        at += v.size();
    }
    return SG_OK;
}

int sg_extract_batch_fetch_wire_begin(sg_ctx *c, sg_wire_feature *out, size_t cap_total, size_t piece_records, int *n_pieces)
{
    if (!c || !n_pieces || piece_records == 0)
        return SG_E_INVALID;
    *n_pieces = 0;
    const int rc = sg_extract_batch_fetch_wire(c, out, cap_total); // everything "lands" at once here
    if (rc != SG_OK)
        return rc;
    int k = 0;
    for (const auto &v : c->pending)
        k += (int)((v.size() + piece_records - 1) / piece_records);
    c->pieces = k;
    *n_pieces = k;
    return SG_OK;
}

int sg_extract_batch_fetch_wire_wait(sg_ctx *c, int piece) { return c && piece >= 0 && piece < c->pieces ? SG_OK : SG_E_INVALID; }

// ---- what the tests read back ----
int fake_sg_creates(void) { return g_creates.load(); }
int fake_sg_live(void) { return g_live.load(); }
int fake_sg_max_live(void) { return g_max_live.load(); }
int fake_sg_batches(void) { return g_batches.load(); }
int fake_sg_max_batch(void) { return g_max_batch_seen.load(); }

// ---- the stage entry points: present so that the host layer links; never reached by the sanitizer programs ----
int sg_convert_to_gray(sg_ctx *, const uint8_t *, int, int, int, size_t, double *) { return SG_E_UNSUPPORTED; }
int sg_upsample(sg_ctx *, const double *, int, int, double *) { return SG_E_UNSUPPORTED; }
int sg_downsample(sg_ctx *, const double *, int, int, double *) { return SG_E_UNSUPPORTED; }
int sg_gaussian_smooth(sg_ctx *, const double *, double *, int, int, double) { return SG_E_UNSUPPORTED; }
int sg_create_init_smooth_gray(sg_ctx *, const uint8_t *, int, int, int, size_t, double, double *) { return SG_E_UNSUPPORTED; }
int sg_gaussian_pyramid(sg_ctx *, const double *, int, int, int, int, double, double *const *) { return SG_E_UNSUPPORTED; }
int sg_sub(sg_ctx *, const double *, const double *, double *, int, int) { return SG_E_UNSUPPORTED; }
int sg_detect_local_extrema(sg_ctx *, const double *const *, int, int, int, int, int, sg_keypoint *, size_t, size_t *,
                            sg_counters *)
{
    return SG_E_UNSUPPORTED;
}
int sg_calculate_scale(sg_keypoint *, size_t, double, int) { return SG_E_UNSUPPORTED; }
int sg_half_features(sg_keypoint *, size_t) { return SG_E_UNSUPPORTED; }
int sg_orientation_assignment(sg_ctx *, const sg_keypoint *, size_t, const double *const *, int, int, int, sg_keypoint *,
                              size_t, size_t *)
{
    return SG_E_UNSUPPORTED;
}
int sg_descriptor_representation(sg_ctx *, sg_keypoint *, size_t, const double *const *, int, int, int)
{
    return SG_E_UNSUPPORTED;
}
int sg_match(sg_ctx *, const sg_keypoint *, size_t, const sg_keypoint *, size_t, double, sg_match_point *, size_t, size_t *)
{
    return SG_E_UNSUPPORTED;
}

} // extern "C"
