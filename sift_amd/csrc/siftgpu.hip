// siftgpu.hip -- context, launch sequence and the C ABI of libsiftgpu (see include/siftgpu.h).
//
// Launch sequence of one batch (reference sift/sift.cpp:1646-1707, Sift()):
//   gray+upsample -> init blur -> per octave { decimate, 5 x (row pass, column pass), fused DoG+extrema scan }
//   -> ordered compaction -> refine -> ordered compaction -> orientation -> ordered compaction
//   -> descriptors -> rank by scale -> pack.
// Everything is enqueued on the context's stream with fixed grids and device-side counts, so the host
// never waits inside a batch.
#include <math.h>
#include <cmath>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "sg_internal.h"
#include "sg_dd.h"

using namespace sg;

namespace sg {
const Env &env()
{
    static const Env e = [] {
        Env v;
        if (const char *b = getenv("SIFTGPU_BLUR"))
            v.blur_mode = strcmp(b, "2pass") == 0 ? 1 : strcmp(b, "ring") == 0 ? 2 : strcmp(b, "win") == 0 ? 3 : 0;
        v.blur_direct = getenv("SIFTGPU_BLUR_DIRECT") != nullptr;
        v.detect_direct = getenv("SIFTGPU_DETECT_DIRECT") != nullptr;
        if (const char *pr = getenv("SIFTGPU_BLUR_PAIR"))
            v.blur_pair = strcmp(pr, "0") == 0 ? 0 : (strcmp(pr, "2") == 0 ? 2 : 1);
        if (const char *t = getenv("SIFTGPU_PYR_TAIL"))
            v.pyr_tail = strcmp(t, "0") != 0;
        if (const char *t = getenv("SIFTGPU_BLUR_REMAP"))
            v.blur_remap = strcmp(t, "0") != 0;
        if (const char *w = getenv("SIFTGPU_ORI_WAVES"))
            v.ori_waves = atoi(w);
        if (const char *w = getenv("SIFTGPU_DESCR_WAVES"))
            v.descr_waves = atoi(w);
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0)
            v.n_cu = cu;
        return v;
    }();
    return e;
}
} // namespace sg

static_assert(sizeof(sg_keypoint) == 1120, "sg_keypoint must match the reference's Keypoint layout");
static_assert(sizeof(sg_packed_feature) == 144, "packed feature is 144 bytes");
static_assert(sizeof(KpRec) == 80, "KpRec is 80 bytes");
static_assert(sizeof(FeatRec) == 64, "FeatRec is 64 bytes");

namespace {

// ---- scratch of the allocation-free matcher (sg_match_packed_device_noalloc) ---------------------------------
// One array per GPU, held from the moment the library's code object is loaded on that GPU until the process
// ends.  Layout: signed-byte rows of the two feature sets, their squared norms, the per-slice partial results,
// one flag word.  Host code never uses the array's name as an address: its device address is resolved once per
// device by static_scratch_for() and kept.
constexpr unsigned int kStaticMatchMax = 32768; // features per set
constexpr size_t kSsRows = (size_t)kStaticMatchMax * 128;
constexpr size_t kSsNorms = (size_t)kStaticMatchMax * sizeof(int);
constexpr size_t kSsPart = (size_t)kStaticMatchMax * kMatchMaxSlices * 12;
constexpr size_t kSsOffRows0 = 0, kSsOffRows1 = kSsRows, kSsOffNorm0 = 2 * kSsRows, kSsOffNorm1 = kSsOffNorm0 + kSsNorms,
                 kSsOffPart = kSsOffNorm1 + kSsNorms, kSsOffBad = kSsOffPart + kSsPart, kSsBytes = kSsOffBad + 256;
static_assert(kSsOffNorm0 % 256 == 0 && kSsOffNorm1 % 256 == 0 && kSsOffPart % 256 == 0 && kSsOffBad % 256 == 0,
              "every part starts on a 256-byte boundary");
static __device__ __align__(256) unsigned char scratch[kSsBytes];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
unsigned char *g_scratch_ptr[kMaxDevices]; // device address of `scratch` on each GPU, resolved on first use

// Device address of this GPU's copy of `scratch`; `device` must be the current device.  nullptr on failure.
unsigned char *static_scratch_for(int device, hipError_t *err)
{
    *err = hipSuccess;
    if (device < 0 || device >= kMaxDevices) {
        *err = hipErrorInvalidDevice;
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        *err = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (*err != hipSuccess)
            return nullptr;
        g_scratch_ptr[device] = static_cast<unsigned char *>(p);
    }
    return g_scratch_ptr[device];
}

constexpr int kEventPairs = 4096;
constexpr int kMaxCommRanks = 1024; // ranks sg_allgather_features sizes its header buffer for

struct StageEvents {
    hipEvent_t start[kEventPairs], stop[kEventPairs];
    int stage[kEventPairs];
    int used = 0;
    bool created = false;
};

} // namespace

// octaves of at most this many pixels per level (all frames of the batch together) blur levels 4 and 5 on the side stream
#ifndef SG_SIDE_PIXELS
#define SG_SIDE_PIXELS (40u << 20) // measured at 1080p: four frames gain 4 %, eight (octave 0: 66 M pixels) gain nothing
#endif
constexpr size_t kSideStreamPixels = SG_SIDE_PIXELS;

struct sg_ctx {
    sg_params prm;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;   // levels 4 and 5 of under-filled octaves (run_pipeline)
    hipEvent_t side_ev[2] = {nullptr, nullptr}; // fork, join
    std::string err;
    int B = 1;
    size_t esize = 8;
    unsigned int feat_cap = 0, cand_cap = 0;
    // descriptor matching scratch (grown on demand): signed-byte rows and squared norms of the two feature sets
    signed char *d_mrows[2] = {nullptr, nullptr};
    int *d_mnorm[2] = {nullptr, nullptr};
    unsigned int m_cap[2] = {0, 0};
    unsigned int *d_mbad = nullptr;
    void *d_mpart = nullptr; // per-slice partial results of the matcher
    size_t mpart_bytes = 0;
    unsigned char *static_scratch = nullptr; // this device's copy of the allocation-free matcher's scratch array
    uint32_t *d_xhdr = nullptr; // sg_allgather_features: this rank's four header words, then every rank's

    // geometry of the current call
    int w = 0, h = 0, octaves = 0, last_batch = 0;
    int rows[kMaxOctaves], cols[kMaxOctaves], pitch[kMaxOctaves];
    Plane level[kMaxOctaves][kLevels];
    OctaveTab tab;
    unsigned int n_words = 0;

    // taps: [0] initial blur, [1..5] pyramid levels
    double taps[kLevels][kMaxAnyTaps]; // Gaussian taps of the base blur (0) and the five pyramid blurs
    int ntaps[kLevels];
    void *d_taps[kLevels] = {}; // device copies in the pyramid's type, for blurs wider than the unrolled kernels

    // device buffers
    uint8_t *d_in = nullptr;
    size_t in_bytes = 0;
    void *d_up = nullptr, *d_tmp = nullptr, *d_levels = nullptr;
    size_t plane0_elems = 0, levels_elems = 0;
    unsigned long long *d_words = nullptr, *d_vwords = nullptr, *d_pwords = nullptr;
    size_t words_img = 0, vwords_img = 0, pwords_img = 0;
    void *d_zero = nullptr; // one allocation for what every batch starts from zero: d_ctr | d_counts | d_chunks
    size_t zero_bytes = 0;
    unsigned int *d_chunks = nullptr; // three arrays per image: detect, valid, peaks
    size_t chunks1 = 0, chunks2 = 0, chunks3 = 0, chunks_total = 0;
    unsigned int *d_cand = nullptr, *d_kpsrc = nullptr, *d_fsrc = nullptr, *d_rank = nullptr;
    unsigned int *d_bwork = nullptr, *d_rtot = nullptr; // rank scratch
    RankEntry *d_blist = nullptr;
    KpRec *d_kp = nullptr;
    FeatRec *d_frec = nullptr;
    double *d_ori = nullptr, *d_pq = nullptr, *d_fscale = nullptr;
    uint8_t *d_desc = nullptr;
    unsigned int *d_counts = nullptr; // [3][B]: candidates, keypoints, features
    unsigned long long *d_ctr = nullptr; // [B][8]
    unsigned long long *d_trips = nullptr; // [B][kTripWordsPerImage]: trip counts of the histogram kernels, in slots
    unsigned int *d_overflow = nullptr;  // [4]: candidates, keypoints, features, output; sticky until read
    unsigned int last_ov[4] = {0, 0, 0, 0}; // the counts of the last reported overflow (sg_required_features)
    void *d_out = nullptr;
    // pinned staging for large downloads into pageable caller memory: two halves, filled and emptied alternately
    unsigned char *h_in = nullptr; // pinned staging of the single-image entry points' input
    size_t h_in_bytes = 0;
    unsigned char *h_stage = nullptr;
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    size_t pending_features = 0; // feature count of the last sg_extract_begin, not fetched yet
    bool have_pending = false;
    std::vector<uint32_t> pending_counts; // per image, of the last sg_extract_batch_begin (empty: none pending)
    std::vector<hipEvent_t> chunk_ev;     // sg_extract_batch_fetch_wire_begin: one per piece of the download
    int chunks_pending = 0;               // pieces of the download in flight (0: none)
    size_t out_bytes = 0;
    unsigned int *d_outcounts = nullptr;

    bool prof = false;
    unsigned int pair_octaves = 0; // octaves of the last call whose levels 1 and 2 came from one launch
    int tail_octave = 0;           // first octave of the last call produced by the pyramid-tail launch (octaves: none)
    bool launch_failed = false; // a blur could not be launched (its taps found no device memory)
    bool fused_blur = true; // one-launch blur; SIFTGPU_BLUR=2pass selects the separate row/column kernels
    bool blur_fma = false;  // SG_F64_FMA: the double filters take fused multiply-adds (prm.precision then reads SG_F64)
    int blur_prec() const { return prm.precision | (blur_fma ? kPrecFma : 0); } // what the blur launchers are handed
    StageEvents ev;
    double stage_ms[SG_ST_COUNT];
    unsigned long long stage_launches[SG_ST_COUNT];
};

// ---------------------------------------------------------------------------------------- helpers

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            if (ctx)                                                                             \
                (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                  \
            return e_ == hipErrorOutOfMemory ? SG_E_NOMEM : SG_E_HIP;                            \
        }                                                                                        \
    } while (0)

static int fail(sg_ctx *ctx, int code, const char *msg)
{
    if (ctx)
        ctx->err = msg;
    return code;
}

// Bytes of a host image that are guaranteed to exist: whole strides for all rows but the last, pixels only for the last.
static size_t image_bytes(int width, int height, int channels, size_t stride)
{
    return (size_t)(height - 1) * stride + (size_t)width * channels;
}

static int round_half_even(double v) { return (int)lrint(v); }

// OpenCV GaussianBlur with Size(0,0) on a non-8-bit plane: ksize = cvRound(sigma*4*2+1)|1;
// getGaussianKernel: exp(-0.5 x^2 / sigma^2) scaled by 1/sum.
static int gaussian_taps(double sigma, double *k, int max_n)
{
    int n = round_half_even(sigma * 4 * 2 + 1) | 1;
    if (n > max_n)
        return -n;
    double scale2x = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = exp(scale2x * x * x);
        k[i] = t;
        sum += t;
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++)
        k[i] *= sum;
    return n;
}

// reference sift/sift.cpp:1653
static int num_octaves(int base_rows, int base_cols)
{
    int m = std::min(base_rows, base_cols);
    int o = (int)(log((double)m) / log(2.0) - 2);
    return o;
}

static int pitch_of(int cols) { return (cols + 3) & ~3; }

struct Geometry {
    int octaves;
    int rows[kMaxOctaves], cols[kMaxOctaves], pitch[kMaxOctaves];
    unsigned int wbase[kMaxOctaves + 1];
    int nrows[kMaxOctaves], wpr[kMaxOctaves];
};

static int make_geometry(int base_rows, int base_cols, int octaves, Geometry *g)
{
    if (octaves < 0)
        octaves = 0;
    if (octaves > kMaxOctaves)
        return SG_E_UNSUPPORTED;
    g->octaves = octaves;
    unsigned int wb = 0;
    for (int o = 0; o < octaves; o++) {
        if (o == 0) {
            g->rows[0] = base_rows;
            g->cols[0] = base_cols;
        } else if (g->rows[o - 1] <= 1 || g->cols[o - 1] <= 1) {
            g->rows[o] = g->rows[o - 1];
            g->cols[o] = g->cols[o - 1];
        } else {
            g->rows[o] = g->rows[o - 1] / 2;
            g->cols[o] = g->cols[o - 1] / 2;
        }
        g->pitch[o] = pitch_of(g->cols[o]);
        int nr = g->rows[o] - 2 * kBorder, nc = g->cols[o] - 2 * kBorder;
        g->nrows[o] = nr > 0 && nc > 0 ? nr : 0;
        g->wpr[o] = nr > 0 && nc > 0 ? (nc + 63) / 64 : 0;
        g->wbase[o] = wb;
        wb += 3u * (unsigned int)g->nrows[o] * (unsigned int)g->wpr[o];
    }
    g->wbase[octaves] = wb;
    for (int o = octaves + 1; o <= kMaxOctaves; o++)
        g->wbase[o] = wb;
    return SG_OK;
}

static size_t levels_elems_for(const Geometry &g, int B)
{
    size_t total = 0;
    for (int o = 0; o < g.octaves; o++)
        total += (size_t)kLevels * B * ((size_t)g.rows[o] * g.pitch[o] + 64);
    return total;
}

// ---- profiling events -----------------------------------------------------------------------------

static void flush_events(sg_ctx *c)
{
    if (!c->ev.used)
        return;
    (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < c->ev.used; i++) {
        float ms = 0;
        if (c->ev.stage[i] >= 0 && hipEventElapsedTime(&ms, c->ev.start[i], c->ev.stop[i]) == hipSuccess) {
            c->stage_ms[c->ev.stage[i]] += ms;
        }
    }
    c->ev.used = 0;
}

struct StageScope {
    sg_ctx *c;
    int idx = -1, n = 0;
    StageScope(sg_ctx *ctx, int stage, int launches = 1) : c(ctx), n(launches)
    {
        if (!c->prof)
            return;
        if (c->ev.used == kEventPairs)
            flush_events(c);
        idx = c->ev.used++;
        c->ev.stage[idx] = stage;
        c->stage_launches[stage] += launches;
        (void)hipEventRecord(c->ev.start[idx], c->stream);
    }
    // nothing was launched under this scope: its event pair and launch count are handed back
    void cancel()
    {
        if (idx < 0)
            return;
        c->stage_launches[c->ev.stage[idx]] -= n;
        if (idx == c->ev.used - 1)
            c->ev.used--; // (the start event was recorded; an unused recorded event is harmless)
        else
            c->ev.stage[idx] = -1;
        idx = -1;
    }
    ~StageScope()
    {
        if (idx >= 0)
            (void)hipEventRecord(c->ev.stop[idx], c->stream);
    }
};

// ---- geometry / tables ------------------------------------------------------------------------------

static int set_geometry(sg_ctx *c, int w, int h, int B)
{
    if (w < 2 || h < 2)
        return fail(c, SG_E_INVALID, "image must be at least 2x2");
    if (w > c->prm.max_width || h > c->prm.max_height)
        return fail(c, SG_E_CAPACITY, "image larger than the context's max_width/max_height");
    if (B < 1 || B > c->B)
        return fail(c, SG_E_CAPACITY, "batch larger than the context's max_batch");
    Geometry g;
    int oct = num_octaves(2 * h, 2 * w);
    int rc = make_geometry(2 * h, 2 * w, oct, &g);
    if (rc != SG_OK)
        return fail(c, rc, "too many octaves");
    if (levels_elems_for(g, B) > c->levels_elems || g.wbase[g.octaves] > c->words_img)
        return fail(c, SG_E_CAPACITY, "pyramid does not fit the context's buffers");
    c->w = w;
    c->h = h;
    c->octaves = g.octaves;
    c->last_batch = B;
    memset(&c->tab, 0, sizeof(c->tab));
    size_t off = 0;
    for (int o = 0; o < g.octaves; o++) {
        c->rows[o] = g.rows[o];
        c->cols[o] = g.cols[o];
        c->pitch[o] = g.pitch[o];
        size_t img = (size_t)g.rows[o] * g.pitch[o] + 64;
        img = (img + 63) & ~(size_t)63;
        for (int l = 0; l < kLevels; l++) {
            Plane &p = c->level[o][l];
            p.p = (char *)c->d_levels + off * c->esize;
            p.rows = g.rows[o];
            p.cols = g.cols[o];
            p.pitch = g.pitch[o];
            p.img_stride = img;
            off += img * B;
            c->tab.lv[o][l] = p.p;
        }
        c->tab.rows[o] = g.rows[o];
        c->tab.cols[o] = g.cols[o];
        c->tab.pitch[o] = g.pitch[o];
        c->tab.img_stride[o] = img;
        c->tab.nrows[o] = g.nrows[o];
        c->tab.wpr[o] = g.wpr[o];
    }
    for (int o = 0; o <= kMaxOctaves; o++)
        c->tab.wbase[o] = g.wbase[o];
    c->tab.octaves = g.octaves;
    c->tab.from_dog = 0;
    c->n_words = g.wbase[g.octaves];
    return SG_OK;
}

// ---- keypoint stages (shared by the whole path and the stage entry points) ------------------------------

// The working buffers (keypoints, feature lists, descriptors, ranks, counts) are about to be overwritten: what an
// earlier sg_extract_begin / sg_extract_batch_begin left for its fetch call is gone, and a fetch must say so rather
// than pack records from the new contents with the old counts.
static void invalidate_pending(sg_ctx *c)
{
    c->have_pending = false;
    c->pending_counts.clear();
    c->chunks_pending = 0;
}

constexpr double kDetectThreshold = 0.5 * 0.03 / 3; // 0.5 * DXTHRESHOLD / intervals (sift.cpp:924)

// What a call starts from zero: chunk sums, counters and counts are one region, one fill.
static void zero_work_area(sg_ctx *c) { (void)hipMemsetAsync(c->d_zero, 0, c->zero_bytes, c->stream); }

// scanned_before: octaves below it have been scanned already (and the work area zeroed before that), see run_pipeline
static void detect_phase(sg_ctx *c, const OctaveTab &tab, unsigned int n_words, int B, int nointerval,
                         int scanned_before = 0)
{
    hipStream_t s = c->stream;
    const int prec = c->prm.precision;
    if (scanned_before == 0)
        zero_work_area(c);
    unsigned int *ch1 = c->d_chunks, *ch2 = c->d_chunks + c->chunks1, *ch3 = ch2 + c->chunks2;
    (void)ch3;
    const double thresh = kDetectThreshold;
    {
        StageScope sc(c, SG_ST_DETECT, 0);
        const int launches = launch_detect(tab, B, c->d_words, c->words_img, ch1, c->chunks_total, c->d_ctr, thresh,
                                           nointerval, prec, s, scanned_before);
        if (c->prof)
            c->stage_launches[SG_ST_DETECT] += launches;
    }
    unsigned int *n_cand = c->d_counts, *n_kp = c->d_counts + c->B;
    {
        StageScope sc(c, SG_ST_COMPACT, 2);
        int nch = (int)((n_words + kChunkWords - 1) / kChunkWords);
        launch_scan(ch1, c->chunks_total, nch, n_cand, c->cand_cap, c->d_overflow + 0, B, s);
        launch_expand(c->d_words, c->words_img, n_words, nullptr, 64, ch1, c->chunks_total, c->d_cand, c->cand_cap,
                      c->cand_cap, B, s);
    }
    {
        StageScope sc(c, SG_ST_REFINE);
        launch_refine(tab, c->d_cand, c->cand_cap, n_cand, c->d_kp, c->cand_cap, c->d_vwords, c->vwords_img, ch2,
                      c->chunks_total, c->d_ctr, c->prm.sigma, c->cand_cap, B, prec, s);
    }
    {
        StageScope sc(c, SG_ST_COMPACT, 2);
        launch_scan(ch2, c->chunks_total, (int)c->chunks2, n_kp, c->feat_cap, c->d_overflow + 1, B, s);
        launch_expand(c->d_vwords, c->vwords_img, (unsigned int)c->vwords_img, n_cand, 64, ch2, c->chunks_total,
                      c->d_kpsrc, c->feat_cap, c->feat_cap, B, s, true); // nearly half of the candidates survive
    }
}

static void orient_phase(sg_ctx *c, const OctaveTab &tab, int B, int max_radius)
{
    hipStream_t s = c->stream;
    unsigned int *ch3 = c->d_chunks + c->chunks1 + c->chunks2;
    unsigned int *n_kp = c->d_counts + c->B, *n_feat = c->d_counts + 2 * c->B;
    {
        StageScope sc(c, SG_ST_ORIENT);
        launch_orient(tab, c->d_kp, c->cand_cap, c->d_kpsrc, c->feat_cap, n_kp, c->d_ori, (size_t)c->feat_cap * kOriBins,
                      c->d_pwords, c->pwords_img, ch3, c->chunks_total, (int)c->chunks3, c->feat_cap, max_radius, B,
                      c->prm.precision, c->d_trips, s);
    }
    {
        StageScope sc(c, SG_ST_COMPACT, 2);
        launch_scan(ch3, c->chunks_total, (int)c->chunks3, n_feat, c->feat_cap, c->d_overflow + 2, B, s);
        // one word per keypoint
        launch_expand(c->d_pwords, c->pwords_img, (unsigned int)c->pwords_img, n_kp, 1, ch3, c->chunks_total, c->d_fsrc,
                      c->feat_cap, c->feat_cap, B, s);
    }
}

// Descriptor window radius for an octave-relative scale (reference sift/sift.cpp:1488).
static int descr_radius(double oscale)
{
    const double r = (3.0 * oscale * sqrt(2.0) * (4 + 1)) / 2.0;
    return r < 1e9 ? (int)r : 1000000000;
}

// Orientation window radius (reference sift/sift.cpp:1326).
static int ori_radius(double oscale)
{
    const double r = 3.0 * 1.5 * oscale;
    return r < 1e9 ? (int)rint(r) : 1000000000;
}

// Largest radii the pipeline can produce: interval + offset_interval stays below intervals + 1.
static int descr_max_radius(const sg_params &p)
{
    return descr_radius(p.sigma * pow(2.0, (p.intervals + 1.0) / p.intervals));
}

static int ori_max_radius(const sg_params &p)
{
    return ori_radius(p.sigma * pow(2.0, (p.intervals + 1.0) / p.intervals));
}

static void descr_phase(sg_ctx *c, const OctaveTab &tab, int B, int max_radius)
{
    hipStream_t s = c->stream;
    unsigned int *n_feat = c->d_counts + 2 * c->B;
    {
        StageScope sc(c, SG_ST_DESCR, 2);
        launch_feat_gather(c->d_kp, c->cand_cap, c->d_kpsrc, c->feat_cap, c->d_fsrc, c->feat_cap, n_feat, c->d_ori,
                           (size_t)c->feat_cap * kOriBins, c->d_frec, c->feat_cap, c->d_fscale, c->feat_cap,
                           c->feat_cap, B, s);
        launch_descr(tab, c->d_frec, c->feat_cap, n_feat, c->d_desc, (size_t)c->feat_cap * 128, c->d_pq,
                     (size_t)c->feat_cap * 128, c->feat_cap, max_radius, B, c->prm.precision, c->d_trips, s);
    }
}

static void rank_phase(sg_ctx *c, int B)
{
    unsigned int *n_feat = c->d_counts + 2 * c->B;
    StageScope sc(c, SG_ST_SORT);
    launch_rank(c->d_fscale, c->feat_cap, n_feat, c->d_rank, c->feat_cap, c->d_bwork, c->d_blist, c->d_rtot,
                c->d_rtot + c->B, c->feat_cap, B, c->stream);
}

static void pack_phase(sg_ctx *c, void *out, size_t out_img_stride, unsigned int out_cap, unsigned int *out_counts,
                       int mode, int sorted, int B)
{
    unsigned int *n_feat = c->d_counts + 2 * c->B;
    StageScope sc(c, SG_ST_SORT);
    launch_pack(c->d_kp, c->cand_cap, c->d_kpsrc, c->feat_cap, c->d_fsrc, c->feat_cap, n_feat, c->d_ori,
                (size_t)c->feat_cap * kOriBins, c->d_desc, (size_t)c->feat_cap * 128, c->d_rank, c->feat_cap, out,
                out_img_stride, out_cap, out_counts, c->d_overflow + 3, mode, sorted, c->feat_cap, B, c->stream);
}

// blur src -> dst through the context's row-pass scratch plane
// dec (optional): receives the 2:1 decimated copy of dst when the fused kernel can write it; returns whether it did.
static bool blur_planes(sg_ctx *c, const Plane &src, const Plane &dst, int B, const double *taps, int n, int slot,
                        const Plane *dec = nullptr, hipStream_t side = nullptr)
{
    if (c->fused_blur && blur_fused_supported(src, n)) {
        StageScope sc(c, SG_ST_BLUR_COL0 + slot);
        const bool with_dec = dec && blur_fused_decimates(src, n, c->blur_prec());
        if (launch_blur_fused(src, dst, B, taps, n, c->blur_prec(), side ? side : c->stream, with_dec ? dec : nullptr))
            return with_dec;
    }
    Plane tmp = src;
    tmp.p = c->d_tmp;
    tmp.pitch = src.pitch;
    tmp.img_stride = (size_t)src.rows * src.pitch;
    // With the fused kernel in use the two-pass fallback (tiny planes) is timed under the row slot as a whole,
    // so that the column slot holds the fused kernel's launches only.
    const void *d_taps = taps == c->taps[slot] ? c->d_taps[slot] : nullptr; // the context's own levels only
    {
        StageScope sc(c, SG_ST_BLUR_ROW0 + slot);
        if (!launch_blur_row(src, tmp, B, taps, n, c->blur_prec(), c->stream, d_taps))
            c->launch_failed = true;
    }
    {
        StageScope sc(c, (c->fused_blur ? SG_ST_BLUR_ROW0 : SG_ST_BLUR_COL0) + slot);
        if (!launch_blur_col(tmp, dst, B, taps, n, c->blur_prec(), c->stream, d_taps))
            c->launch_failed = true;
    }
    return false;
}

static int run_pipeline(sg_ctx *c, const uint8_t *d_imgs, int B, int w, int h, int ch, size_t stride, size_t img_stride)
{
    int rc = set_geometry(c, w, h, B);
    if (rc != SG_OK)
        return rc;
    if (ch < 3)
        return fail(c, SG_E_INVALID, "input needs at least 3 interleaved channels (b,g,r)");
    hipStream_t s = c->stream;
    const int prec = c->prm.precision;
    c->launch_failed = false;
    invalidate_pending(c);
    // d_overflow is not cleared here: the words are sticky until check_overflow has read them, so that a second batch
    // enqueued before sg_sync cannot erase the first one's report
    if (c->octaves == 0) {
        (void)hipMemsetAsync(c->d_counts, 0, sizeof(unsigned int) * 3 * c->B, s);
        (void)hipMemsetAsync(c->d_ctr, 0, sizeof(unsigned long long) * 8 * c->B, s);
        return SG_OK;
    }
    Plane up;
    up.p = c->d_up;
    up.rows = 2 * h;
    up.cols = 2 * w;
    up.pitch = pitch_of(2 * w);
    up.img_stride = (size_t)up.rows * up.pitch;
    // Base image (CreateInitSmoothGray, sift.cpp:335-346).  Preferred form: the gray plane at input resolution
    // (always double, so that a float pyramid rounds once) and one blur launch that forms the upsampled pixels
    // itself; otherwise gray + upsample into a plane, then the blur.
    bool base_done = false;
    if (c->fused_blur && h >= 2 && w >= 2) {
        Plane gray;
        gray.p = c->d_up; // 2h x 2w elements of T hold h x w doubles
        gray.rows = h;
        gray.cols = w;
        gray.pitch = pitch_of(w);
        gray.img_stride = (size_t)gray.rows * gray.pitch;
        {
            StageScope sc(c, SG_ST_PREP);
            launch_gray(d_imgs, h, w, ch, stride, img_stride, gray, B, SG_F64, s);
        }
        StageScope sc(c, SG_ST_BLUR_COL0);
        base_done = launch_blur_fused_upsampled(gray, c->level[0][0], B, c->taps[0], c->ntaps[0], c->blur_prec(), s);
    }
    if (!base_done) {
        {
            StageScope sc(c, SG_ST_PREP);
            launch_gray_upsample(d_imgs, h, w, ch, stride, img_stride, up, B, prec, s);
        }
        blur_planes(c, up, c->level[0][0], B, c->taps[0], c->ntaps[0], 0);
    }
    bool have_base = true; // level 0 of the octave is in place
    bool forked = false;
    int side_octaves = 0; // octaves 0..side_octaves-1 have their levels 4 and 5 on the side stream
    c->pair_octaves = 0;
    // The smallest octaves (planes that fit a workgroup's LDS) come from one launch after the loop.
    const double *tail_taps[kLevels - 1];
    int tail_ntaps[kLevels - 1];
    for (int l = 1; l < kLevels; l++) {
        tail_taps[l - 1] = c->taps[l];
        tail_ntaps[l - 1] = c->ntaps[l];
    }
    const int tail = c->fused_blur ? pyr_tail_first(c->tab, tail_ntaps, c->blur_prec()) : c->octaves;
    c->tail_octave = c->octaves;
    // A frame alone: the tail launch keeps one compute unit busy for about as long as the scan of the octaves before
    // it takes on the rest of the chip, so that scan runs beside it, on the side stream behind those octaves' levels 4
    // and 5.  The work area it adds to is then zeroed here, ahead of everything.
    const bool scan_beside_tail = tail >= 1 && tail < c->octaves && c->side && !c->prof &&
                                  (size_t)B * c->rows[0] * c->cols[0] <= kSideStreamPixels;
    int scanned_before = 0;
    if (scan_beside_tail)
        zero_work_area(c);
    for (int o = 0; o < c->octaves; o++) {
        if (!have_base) {
            StageScope sc(c, SG_ST_DOWNSAMPLE);
            if (c->rows[o] == c->rows[o - 1]) // a dimension <= 1: DownSample copies (sift.cpp:64-68)
                (void)hipMemcpyAsync(c->level[o][0].p, c->level[o - 1][3].p,
                                     c->esize * c->level[o][0].img_stride * B, hipMemcpyDeviceToDevice, s);
            else
                launch_downsample(c->level[o - 1][3], c->level[o][0], B, prec, s);
        }
        if (o == tail) { // level 0 of the first tail octave is in place: one launch takes the pyramid from here
            StageScope sc(c, SG_ST_PYR_TAIL);
            if (launch_pyr_tail(c->tab, tail, B, tail_taps, tail_ntaps, c->blur_prec(), s)) {
                c->tail_octave = tail;
                if (scan_beside_tail && side_octaves == tail) { // every octave before the tail has its levels 4, 5 there
                    launch_detect(c->tab, B, c->d_words, c->words_img, c->d_chunks, c->chunks_total, c->d_ctr,
                                  kDetectThreshold, c->prm.nointerval, prec, c->side, 0, tail);
                    scanned_before = tail;
                }
                break;
            }
            sc.cancel();
        }
        have_base = false;
        // Levels 4 and 5 feed nothing but the extrema scan, while the next octave starts from level 3.  Where a launch
        // cannot fill the chip (a frame alone, the small octaves of a small batch) the time goes into the chain of
        // dependent launches, so those two leave the chain: they run on the side stream, beside the next octave's
        // levels.  (Not while stage events are recorded: they time launches on the main stream, one after another.)
        const bool side_ok = c->side && !c->prof && c->fused_blur && o + 1 < c->octaves &&
                             (size_t)B * c->rows[o] * c->cols[o] <= kSideStreamPixels &&
                             blur_fused_supported(c->level[o][3], c->ntaps[4]) &&
                             blur_fused_supported(c->level[o][4], c->ntaps[5]);
        for (int l = 1; l < kLevels; l++) {
            if (l == 1 && c->fused_blur) { // levels 1 and 2 in one launch where that form applies
                StageScope sc(c, SG_ST_BLUR_PAIR);
                if (launch_blur_pair(c->level[o][0], c->level[o][1], c->level[o][2], B, c->taps[1], c->ntaps[1], c->taps[2],
                                     c->ntaps[2], c->blur_prec(), s)) {
                    c->pair_octaves |= 1u << o;
                    l = 2;
                    continue;
                }
                sc.cancel();
            }
            // the blur that produces level 3 also writes the next octave's level 0 (its 2:1 decimation) when it can
            const bool want_dec = l == 3 && o + 1 < c->octaves && c->rows[o + 1] != c->rows[o];
            if (l == 4 && side_ok) {
                (void)hipEventRecord(c->side_ev[0], s);
                (void)hipStreamWaitEvent(c->side, c->side_ev[0], 0);
                forked = true;
                if (side_octaves == o)
                    side_octaves = o + 1;
            }
            const bool wrote = blur_planes(c, c->level[o][l - 1], c->level[o][l], B, c->taps[l], c->ntaps[l], l,
                                           want_dec ? &c->level[o + 1][0] : nullptr, l >= 4 && side_ok ? c->side : nullptr);
            if (want_dec && wrote)
                have_base = true;
        }
    }
    if (forked) { // the scan reads every level
        (void)hipEventRecord(c->side_ev[1], c->side);
        (void)hipStreamWaitEvent(s, c->side_ev[1], 0);
    }
    if (c->launch_failed)
        return fail(c, SG_E_NOMEM, "out of device memory (blur taps)");
    detect_phase(c, c->tab, c->n_words, B, c->prm.nointerval, scanned_before);
    orient_phase(c, c->tab, B, ori_max_radius(c->prm));
    descr_phase(c, c->tab, B, descr_max_radius(c->prm));
    rank_phase(c, B);
    HIPCHK(c, hipGetLastError());
    return SG_OK;
}

static int check_overflow(sg_ctx *c)
{
    unsigned int ov[4];
    HIPCHK(c, hipMemcpyAsync(ov, c->d_overflow, sizeof ov, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (ov[0] || ov[1] || ov[2] || ov[3]) {
        memcpy(c->last_ov, ov, sizeof ov);
        // read and reported: cleared for the calls that follow (stream order puts this after everything enqueued so far)
        (void)hipMemsetAsync(c->d_overflow, 0, sizeof(unsigned int) * 4, c->stream);
        char buf[256];
        snprintf(buf, sizeof buf,
                 "capacity exceeded: candidates %u (cap %u), keypoints %u, features %u (cap %u), output %u",
                 ov[0], c->cand_cap, ov[1], ov[2], c->feat_cap, ov[3]);
        c->err = buf;
        return SG_E_CAPACITY;
    }
    return SG_OK;
}

static int ensure_in(sg_ctx *c, size_t bytes)
{
    if (bytes <= c->in_bytes)
        return SG_OK;
    if (c->d_in)
        (void)hipFree(c->d_in);
    c->d_in = nullptr;
    c->in_bytes = 0;
    HIPCHK(c, hipMalloc((void **)&c->d_in, bytes));
    c->in_bytes = bytes;
    return SG_OK;
}

static int ensure_pinned_in(sg_ctx *c, size_t bytes)
{
    if (bytes <= c->h_in_bytes)
        return SG_OK;
    if (c->h_in) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipHostFree(c->h_in);
    }
    c->h_in = nullptr;
    c->h_in_bytes = 0;
    if (hipHostMalloc((void **)&c->h_in, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        c->h_in = nullptr;
        return SG_E_NOMEM;
    }
    c->h_in_bytes = bytes;
    return SG_OK;
}

static int ensure_out(sg_ctx *c, size_t bytes)
{
    if (bytes <= c->out_bytes)
        return SG_OK;
    if (c->d_out)
        (void)hipFree(c->d_out);
    c->d_out = nullptr;
    c->out_bytes = 0;
    HIPCHK(c, hipMalloc(&c->d_out, bytes));
    c->out_bytes = bytes;
    return SG_OK;
}

// Feature records for the host: the device packs WireRec records (224 bytes each), they come down in pieces into
// pinned memory, two pieces in flight alternately, and the CPU expands each landed piece into the caller's 1120-byte
// sg_keypoint records while the next one is on the bus.  (The record's 128 descriptor doubles hold integers 0..255:
// moving them as bytes cuts the transfer five-fold, and the caller's memory, usually pageable, is written once, by
// the CPU.)  Synchronises the stream.
constexpr size_t kStagePiece = 4u << 20;
constexpr size_t kUploadPiece = 1u << 20; // pieces in which a pageable image goes through the page-locked buffer
constexpr size_t kStageRecs = kStagePiece / sizeof(WireRec);

static int ensure_stage(sg_ctx *c)
{
    if (c->h_stage)
        return SG_OK;
    HIPCHK(c, hipHostMalloc((void **)&c->h_stage, 2 * kStagePiece, hipHostMallocDefault));
    for (int i = 0; i < 2; i++)
        if (!c->stage_ev[i])
            HIPCHK(c, hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming));
    return SG_OK;
}

static_assert(sizeof(sg_wire_feature) == sizeof(WireRec) && offsetof(sg_wire_feature, desc) == offsetof(WireRec, desc) &&
                  offsetof(sg_wire_feature, ori) == offsetof(WireRec, ori) &&
                  offsetof(sg_wire_feature, val) == offsetof(WireRec, val) &&
                  offsetof(sg_wire_feature, octave_scale) == offsetof(WireRec, oscale),
              "the public wire record is the device's WireRec");

// wire records -> reference-layout records: sg_expand_wire (sg_wire.cpp)
static void expand_records(const WireRec *w, size_t n, sg_keypoint *out)
{
    sg_expand_wire(reinterpret_cast<const sg_wire_feature *>(w), n, out);
}

// d_wire: n WireRec records on the device -> out[0..n)
static int download_records(sg_ctx *c, sg_keypoint *out, const WireRec *d_wire, size_t n)
{
    int rc = ensure_stage(c);
    if (rc != SG_OK)
        return rc;
    const size_t pieces = (n + kStageRecs - 1) / kStageRecs;
    for (size_t i = 0; i <= pieces; i++) {
        if (i < pieces) {
            const size_t first = i * kStageRecs, cnt = std::min(kStageRecs, n - first);
            HIPCHK(c, hipMemcpyAsync(c->h_stage + (i & 1) * kStagePiece, d_wire + first, cnt * sizeof(WireRec),
                                     hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipEventRecord(c->stage_ev[i & 1], c->stream));
        }
        if (i >= 1) { // piece i-1 has landed once its event has passed; its half is refilled by piece i+1, issued after this
            const size_t first = (i - 1) * kStageRecs, cnt = std::min(kStageRecs, n - first);
            HIPCHK(c, hipEventSynchronize(c->stage_ev[(i - 1) & 1]));
            expand_records(reinterpret_cast<const WireRec *>(c->h_stage + ((i - 1) & 1) * kStagePiece), cnt, out + first);
        }
    }
    return SG_OK;
}

// =========================================================================================== C ABI

extern "C" {

void sg_default_params(sg_params *p)
{
    if (!p)
        return;
    memset(p, 0, sizeof *p);
    p->sigma = 1.6;
    p->intervals = 3;
    p->precision = SG_F64;
    p->nointerval = 0;
    p->max_width = 1920;
    p->max_height = 1080;
    p->max_batch = 1;
    p->max_features = 0;
    p->device = 0;
    p->verbose = 0;
    p->keep_prequant = 0;
}

const char *sg_strerror(int status)
{
    switch (status) {
    case SG_OK: return "ok";
    case SG_E_INVALID: return "invalid argument";
    case SG_E_HIP: return "HIP runtime error";
    case SG_E_CAPACITY: return "capacity exceeded";
    case SG_E_UNSUPPORTED: return "unsupported parameter";
    case SG_E_NOMEM: return "out of memory";
    default: return "unknown status";
    }
}

const char *sg_last_error(const sg_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

int sg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int sg_destroy(sg_ctx *c)
{
    if (!c)
        return SG_OK;
    (void)hipSetDevice(c->prm.device);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    void *bufs[] = {c->d_in, c->d_up, c->d_tmp, c->d_levels, c->d_words, c->d_vwords, c->d_pwords, c->d_zero,
                    c->d_cand, c->d_kpsrc, c->d_fsrc, c->d_rank, c->d_kp, c->d_ori, c->d_pq, c->d_fscale, c->d_desc,
                    c->d_overflow, c->d_out, c->d_outcounts, c->d_bwork, c->d_blist, c->d_rtot, c->d_frec};
    for (void *b : bufs)
        if (b)
            (void)hipFree(b);
    void *mbufs[] = {c->d_mrows[0], c->d_mrows[1], c->d_mnorm[0], c->d_mnorm[1], c->d_mbad, c->d_mpart,
                     c->d_taps[0], c->d_taps[1], c->d_taps[2], c->d_taps[3], c->d_taps[4], c->d_taps[5], c->d_xhdr};
    for (void *b : mbufs)
        if (b)
            (void)hipFree(b);
    if (c->h_stage)
        (void)hipHostFree(c->h_stage);
    if (c->h_in)
        (void)hipHostFree(c->h_in);
    for (hipEvent_t e : c->stage_ev)
        if (e)
            (void)hipEventDestroy(e);
    if (c->ev.created)
        for (int i = 0; i < kEventPairs; i++) {
            (void)hipEventDestroy(c->ev.start[i]);
            (void)hipEventDestroy(c->ev.stop[i]);
        }
    for (hipEvent_t e : c->chunk_ev)
        if (e)
            (void)hipEventDestroy(e);
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
    }
    for (hipEvent_t e : c->side_ev)
        if (e)
            (void)hipEventDestroy(e);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
    return SG_OK;
}

int sg_create(const sg_params *p_in, sg_ctx **out)
{
    if (!p_in || !out)
        return SG_E_INVALID;
    *out = nullptr;
    // SG_F64_FMA is SG_F64 everywhere (storage, keypoint stages) except in the arithmetic of the blur kernels
    sg_params prm_local = *p_in;
    const bool blur_fma = prm_local.precision == SG_F64_FMA;
    if (blur_fma)
        prm_local.precision = SG_F64;
    const sg_params *p = &prm_local;
    if (p->intervals != 3)
        return SG_E_UNSUPPORTED;
    if (!(p->sigma > 1.0) || p->max_width < 2 || p->max_height < 2 || p->max_batch < 1)
        return SG_E_INVALID; // sigma <= 2*INIT_SIGMA makes the initial blur's sigma imaginary (sift.cpp:343)
    if (p->precision != SG_F64 && p->precision != SG_F32)
        return SG_E_INVALID;
    {
        // The keypoint kernels address a pixel of a pyramid level by a 32-bit byte offset from the level's first pixel:
        // the base plane (2W x 2H elements of the pyramid's type, rows padded to 4 elements) must stay below 4 GB,
        // i.e. about 134 megapixels of input in double precision (an 8K frame has 33).
        const unsigned long long base_bytes = 2ull * (unsigned long long)p->max_height *
                                              (unsigned long long)((2ll * p->max_width + 3) & ~3ll) *
                                              (p->precision == SG_F64 ? 8ull : 4ull);
        if (p->max_width > (1 << 29) || p->max_height > (1 << 29) || base_bytes >= (1ull << 32))
            return SG_E_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p->device < 0 || p->device >= ndev)
        return SG_E_HIP;
    if (hipSetDevice(p->device) != hipSuccess)
        return SG_E_HIP;
    sg_ctx *c = new sg_ctx;
    c->prm = *p;
    c->B = p->max_batch;
    c->esize = p->precision == SG_F64 ? 8 : 4;
    c->fused_blur = env().blur_mode != 1; // the switches are read here, once per process
    c->blur_fma = blur_fma;
    memset(c->stage_ms, 0, sizeof c->stage_ms);
    memset(c->stage_launches, 0, sizeof c->stage_launches);

    // taps: initial blur sigma (sift.cpp:343) and the pyramid's incremental sigmas (sift.cpp:361-372)
    double sig[kLevels];
    sig[0] = sqrt(p->sigma * p->sigma - (0.5 * 2) * (0.5 * 2));
    {
        double k = pow(2.0, 1.0 / p->intervals);
        for (int i = 1; i < kLevels; i++) {
            double sig_prev = pow(k, i - 1) * p->sigma;
            double sig_total = sig_prev * k;
            sig[i] = sqrt(sig_total * sig_total - sig_prev * sig_prev);
        }
    }
    if (!(p->sigma > 1.0)) { // the base blur is sqrt(sigma^2 - (2*INIT_SIGMA)^2), sift.cpp:343
        delete c;
        return SG_E_UNSUPPORTED;
    }
    for (int i = 0; i < kLevels; i++) {
        c->ntaps[i] = gaussian_taps(sig[i], c->taps[i], kMaxAnyTaps);
        if (c->ntaps[i] < 0) {
            delete c;
            return SG_E_UNSUPPORTED; // more taps than the any-radius kernels take
        }
    }

    const int W = p->max_width, H = p->max_height, B = c->B;
    size_t px = (size_t)W * H;
    c->feat_cap = p->max_features > 0 ? (unsigned int)p->max_features : (unsigned int)(px / 40 + 4096);
    c->cand_cap = 3 * c->feat_cap + 4096;
    c->cand_cap = (c->cand_cap + 63) & ~63u;

    Geometry g;
    int rc = make_geometry(2 * H, 2 * W, std::max(0, num_octaves(2 * H, 2 * W)), &g);
    if (rc != SG_OK) {
        delete c;
        return rc;
    }
    // a smaller image can have pitch padding that a max-size layout lacks: keep a margin
    c->levels_elems = levels_elems_for(g, B) + (size_t)kLevels * kMaxOctaves * B * 256;
    c->plane0_elems = (size_t)(2 * H) * pitch_of(2 * W) + 64;
    c->words_img = g.wbase[g.octaves] + 2048;
    c->vwords_img = c->cand_cap / 64;
    c->pwords_img = c->feat_cap;
    c->chunks1 = (c->words_img + kChunkWords - 1) / kChunkWords + 1;
    c->chunks2 = (c->vwords_img + kChunkWords - 1) / kChunkWords + 1;
    c->chunks3 = (c->pwords_img + kChunkWords - 1) / kChunkWords + 1;
    c->chunks_total = c->chunks1 + c->chunks2 + c->chunks3;

#define ALLOC(ptr, bytes)                                                  \
    do {                                                                   \
        hipError_t e_ = hipMalloc((void **)&(ptr), (bytes));               \
        if (e_ != hipSuccess) {                                            \
            sg_destroy(c);                                                 \
            return e_ == hipErrorOutOfMemory ? SG_E_NOMEM : SG_E_HIP;      \
        }                                                                  \
    } while (0)
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return SG_E_HIP;
    }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->side_ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->side_ev[1], hipEventDisableTiming) != hipSuccess) {
        sg_destroy(c);
        return SG_E_HIP;
    }
    {
        hipError_t e = hipSuccess;
        c->static_scratch = static_scratch_for(p->device, &e); // p->device is current (hipSetDevice above)
        if (!c->static_scratch) {
            sg_destroy(c);
            return SG_E_HIP;
        }
    }
    ALLOC(c->d_up, c->esize * c->plane0_elems * B);
    ALLOC(c->d_tmp, c->esize * c->plane0_elems * B);
    ALLOC(c->d_levels, c->esize * c->levels_elems);
    ALLOC(c->d_words, sizeof(unsigned long long) * c->words_img * B);
    ALLOC(c->d_vwords, sizeof(unsigned long long) * c->vwords_img * B);
    ALLOC(c->d_pwords, sizeof(unsigned long long) * c->pwords_img * B);
    {
        const size_t ctr_b = sizeof(unsigned long long) * 8 * B, cnt_b = (sizeof(unsigned int) * 3 * B + 15) & ~(size_t)15;
        const size_t trips_b = sizeof(unsigned long long) * kTripWordsPerImage * B;
        c->zero_bytes = ctr_b + trips_b + cnt_b + sizeof(unsigned int) * c->chunks_total * B;
        ALLOC(c->d_zero, c->zero_bytes);
        c->d_ctr = (unsigned long long *)c->d_zero;
        c->d_trips = (unsigned long long *)((char *)c->d_zero + ctr_b);
        c->d_counts = (unsigned int *)((char *)c->d_zero + ctr_b + trips_b);
        c->d_chunks = (unsigned int *)((char *)c->d_zero + ctr_b + trips_b + cnt_b);
    }
    ALLOC(c->d_cand, sizeof(unsigned int) * (size_t)c->cand_cap * B);
    ALLOC(c->d_kp, sizeof(KpRec) * (size_t)c->cand_cap * B);
    ALLOC(c->d_kpsrc, sizeof(unsigned int) * (size_t)c->feat_cap * B);
    ALLOC(c->d_fsrc, sizeof(unsigned int) * (size_t)c->feat_cap * B);
    ALLOC(c->d_rank, sizeof(unsigned int) * (size_t)c->feat_cap * B);
    ALLOC(c->d_ori, sizeof(double) * (size_t)c->feat_cap * kOriBins * B);
    ALLOC(c->d_fscale, sizeof(double) * (size_t)c->feat_cap * B);
    ALLOC(c->d_frec, sizeof(FeatRec) * (size_t)c->feat_cap * B);
    ALLOC(c->d_desc, (size_t)c->feat_cap * 128 * B);
    if (p->keep_prequant)
        ALLOC(c->d_pq, sizeof(double) * (size_t)c->feat_cap * 128 * B);
    ALLOC(c->d_overflow, sizeof(unsigned int) * 4);
    ALLOC(c->d_outcounts, sizeof(unsigned int) * B);
    ALLOC(c->d_bwork, sizeof(unsigned int) * 2 * kRankBuckets * B);
    ALLOC(c->d_blist, sizeof(RankEntry) * (size_t)c->feat_cap * B);
    ALLOC(c->d_rtot, sizeof(unsigned int) * (B + 1));
    for (int i = 0; i < kLevels; i++) {
        if (c->ntaps[i] <= kMaxTaps)
            continue; // served by the unrolled kernels, taps travel as a kernel argument
        ALLOC(c->d_taps[i], c->esize * (size_t)c->ntaps[i]);
        if (c->esize == 8) {
            (void)hipMemcpy(c->d_taps[i], c->taps[i], sizeof(double) * c->ntaps[i], hipMemcpyHostToDevice);
        } else {
            std::vector<float> tf(c->taps[i], c->taps[i] + c->ntaps[i]);
            (void)hipMemcpy(c->d_taps[i], tf.data(), sizeof(float) * c->ntaps[i], hipMemcpyHostToDevice);
        }
    }
#undef ALLOC
    (void)hipMemset(c->d_words, 0, sizeof(unsigned long long) * c->words_img * B);
    (void)hipMemset(c->d_vwords, 0, sizeof(unsigned long long) * c->vwords_img * B);
    (void)hipMemset(c->d_pwords, 0, sizeof(unsigned long long) * c->pwords_img * B);
    (void)hipMemset(c->d_zero, 0, c->zero_bytes);
    (void)hipMemset(c->d_overflow, 0, sizeof(unsigned int) * 4);
    (void)hipDeviceSynchronize();
    *out = c;
    return SG_OK;
}

int sg_required_features(const sg_ctx *c, size_t *max_features)
{
    if (!c || !max_features)
        return SG_E_INVALID;
    // feat_cap holds keypoints and features, cand_cap = 3 * feat_cap + 4096 holds the 3x3x3 extrema.  When the extrema
    // overflowed, the later counts were taken on a truncated list: scale them by the truncation and add a margin.
    // A stage that overflowed handed a truncated list to the stages after it: their counts are scaled up by the
    // truncation (keypoints -> features, extrema -> both), and a margin is added.
    double kps = c->last_ov[1], feats = c->last_ov[2];
    if (kps > c->feat_cap)
        feats *= kps / c->feat_cap;
    double need = std::max(kps, feats);
    if (c->last_ov[0] > c->cand_cap) {
        const double grow = (double)c->last_ov[0] / c->cand_cap;
        need = std::max(need * grow, (c->last_ov[0] - 4096.0) / 3.0);
        need = std::max(need, (double)c->feat_cap * grow);
    }
    *max_features = need > 0 ? (size_t)(need * 1.25) + 1024 : 0;
    return SG_OK;
}

int sg_sync(sg_ctx *c)
{
    if (!c)
        return SG_E_INVALID;
    return check_overflow(c);
}

// ---- whole path ---------------------------------------------------------------------------------------

int sg_extract_batch_device(sg_ctx *c, const uint8_t *d_imgs, int n_img, int width, int height, int channels,
                            size_t stride, size_t img_stride, sg_packed_feature *d_out, size_t cap_per_image,
                            uint32_t *d_counts)
{
    if (!c || !d_imgs || !d_out || !d_counts || n_img < 1)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    int rc = run_pipeline(c, d_imgs, n_img, width, height, channels, stride, img_stride);
    if (rc != SG_OK)
        return rc;
    unsigned int cap = cap_per_image > 0xffffffffu ? 0xffffffffu : (unsigned int)cap_per_image;
    pack_phase(c, d_out, cap_per_image, cap, d_counts, 0, 1, n_img);
    HIPCHK(c, hipGetLastError());
    return SG_OK;
}

int sg_extract_batch(sg_ctx *c, const uint8_t *const *imgs, int n_img, int width, int height, int channels,
                     size_t stride, sg_packed_feature *out, size_t cap_per_image, uint32_t *counts)
{
    if (!c || !imgs || !out || !counts || n_img < 1 || n_img > c->B)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    if (width < 1 || height < 1 || channels < 3 || stride < (size_t)width * channels)
        return fail(c, SG_E_INVALID, "stride is smaller than a row of pixels");
    size_t img_bytes = stride * (size_t)height;
    int rc = ensure_in(c, img_bytes * n_img);
    if (rc != SG_OK)
        return rc;
    rc = ensure_out(c, sizeof(sg_packed_feature) * cap_per_image * n_img);
    if (rc != SG_OK)
        return rc;
    for (int i = 0; i < n_img; i++) // the last row of a padded view ends after its pixels, not after its stride
        HIPCHK(c, hipMemcpyAsync(c->d_in + img_bytes * i, imgs[i], image_bytes(width, height, channels, stride),
                                 hipMemcpyHostToDevice, c->stream));
    rc = sg_extract_batch_device(c, c->d_in, n_img, width, height, channels, stride, img_bytes,
                                 (sg_packed_feature *)c->d_out, cap_per_image, c->d_outcounts);
    if (rc != SG_OK)
        return rc;
    HIPCHK(c, hipMemcpyAsync(counts, c->d_outcounts, sizeof(uint32_t) * n_img, hipMemcpyDeviceToHost, c->stream));
    rc = check_overflow(c);
    if (rc != SG_OK)
        return rc;
    for (int i = 0; i < n_img; i++)
        HIPCHK(c, hipMemcpyAsync(out + cap_per_image * i, (sg_packed_feature *)c->d_out + cap_per_image * i,
                                 sizeof(sg_packed_feature) * counts[i], hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SG_OK;
}

int sg_extract_begin(sg_ctx *c, const uint8_t *px, int width, int height, int channels, size_t stride, size_t *n)
{
    if (!c || !px || !n)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    c->pending_counts.clear();
    c->have_pending = false;
    if (width < 1 || height < 1 || channels < 3 || stride < (size_t)width * channels)
        return fail(c, SG_E_INVALID, "stride is smaller than a row of pixels");
    size_t img_bytes = stride * (size_t)height;
    int rc = ensure_in(c, img_bytes);
    if (rc != SG_OK)
        return rc;
    {
        // The caller's image is pageable as a rule: it goes through a pinned buffer of the context, so that the
        // transfer itself is one asynchronous DMA (a pageable source makes the runtime stage the copy under its own lock,
        // which serialises host threads that each drive a context).
        const size_t nbytes = image_bytes(width, height, channels, stride);
        const bool pinned = ensure_pinned_in(c, nbytes) == SG_OK;
        if (!pinned)
            HIPCHK(c, hipMemcpyAsync(c->d_in, px, nbytes, hipMemcpyHostToDevice, c->stream));
        for (size_t at = 0; pinned && at < nbytes; at += kUploadPiece) { // a piece is on the bus while the next is copied
            const size_t len = std::min(kUploadPiece, nbytes - at);
            copy_to_staging(c->h_in + at, px + at, len);
            HIPCHK(c, hipMemcpyAsync(c->d_in + at, c->h_in + at, len, hipMemcpyHostToDevice, c->stream));
        }
    }
    rc = run_pipeline(c, c->d_in, 1, width, height, channels, stride, img_bytes);
    if (rc != SG_OK)
        return rc;
    unsigned int nf = 0;
    HIPCHK(c, hipMemcpyAsync(&nf, c->d_counts + 2 * c->B, sizeof nf, hipMemcpyDeviceToHost, c->stream));
    rc = check_overflow(c); // synchronises
    if (rc != SG_OK)
        return rc;
    *n = nf;
    c->pending_features = nf;
    c->have_pending = true;
    return SG_OK;
}

int sg_extract_fetch(sg_ctx *c, sg_keypoint *out, size_t cap, size_t *n)
{
    if (!c || (cap && !out))
        return SG_E_INVALID;
    if (!c->have_pending)
        return fail(c, SG_E_INVALID, "sg_extract_fetch without a successful sg_extract_begin before it");
    (void)hipSetDevice(c->prm.device);
    const size_t nf = c->pending_features;
    if (n)
        *n = nf;
    if (nf > cap)
        return fail(c, SG_E_CAPACITY, "output array too small");
    if (nf == 0)
        return SG_OK;
    int rc = ensure_out(c, sizeof(WireRec) * nf);
    if (rc != SG_OK)
        return rc;
    pack_phase(c, c->d_out, nf, (unsigned int)nf, nullptr, 3, 1, 1);
    HIPCHK(c, hipGetLastError());
    return download_records(c, out, (const WireRec *)c->d_out, nf);
}

int sg_extract_batch_begin(sg_ctx *c, const uint8_t *const *imgs, int n_img, int width, int height, int channels,
                           size_t stride, int pinned, uint32_t *counts)
{
    if (!c || !imgs || !counts || n_img < 1 || n_img > c->B)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    c->pending_counts.clear();
    c->have_pending = false;
    if (width < 1 || height < 1 || channels < 3 || stride < (size_t)width * channels)
        return fail(c, SG_E_INVALID, "stride is smaller than a row of pixels");
    for (int i = 0; i < n_img; i++)
        if (!imgs[i])
            return SG_E_INVALID;
    const size_t img_bytes = stride * (size_t)height, nbytes = image_bytes(width, height, channels, stride);
    int rc = ensure_in(c, img_bytes * n_img);
    if (rc != SG_OK)
        return rc;
    const bool staged = !pinned && ensure_pinned_in(c, img_bytes * n_img) == SG_OK;
    for (int i = 0; i < n_img; i++) {
        if (!staged) {
            HIPCHK(c, hipMemcpyAsync(c->d_in + img_bytes * i, imgs[i], nbytes, hipMemcpyHostToDevice, c->stream));
            continue;
        }
        // through the page-locked buffer a piece at a time: the transfer of one piece runs while the next is copied
        for (size_t at = 0; at < nbytes; at += kUploadPiece) {
            const size_t len = std::min(kUploadPiece, nbytes - at);
            copy_to_staging(c->h_in + img_bytes * i + at, imgs[i] + at, len);
            HIPCHK(c, hipMemcpyAsync(c->d_in + img_bytes * i + at, c->h_in + img_bytes * i + at, len,
                                     hipMemcpyHostToDevice, c->stream));
        }
    }
    rc = run_pipeline(c, c->d_in, n_img, width, height, channels, stride, img_bytes);
    if (rc != SG_OK)
        return rc;
    HIPCHK(c, hipMemcpyAsync(counts, c->d_counts + 2 * c->B, sizeof(uint32_t) * n_img, hipMemcpyDeviceToHost,
                             c->stream));
    rc = check_overflow(c); // synchronises
    if (rc != SG_OK)
        return rc;
    c->pending_counts.assign(counts, counts + n_img);
    return SG_OK;
}

int sg_extract_batch_fetch_wire(sg_ctx *c, sg_wire_feature *out, size_t cap_total)
{
    if (!c || (cap_total && !out))
        return SG_E_INVALID;
    if (c->pending_counts.empty())
        return fail(c, SG_E_INVALID, "sg_extract_batch_fetch_wire without a successful sg_extract_batch_begin before it");
    (void)hipSetDevice(c->prm.device);
    const int n_img = (int)c->pending_counts.size();
    size_t total = 0, widest = 0;
    for (uint32_t v : c->pending_counts) {
        total += v;
        widest = std::max<size_t>(widest, v);
    }
    if (total > cap_total)
        return fail(c, SG_E_CAPACITY, "output array too small");
    if (total == 0)
        return SG_OK;
    // every image packs into a slot of `widest` records; the slots come down back to back
    int rc = ensure_out(c, sizeof(WireRec) * widest * n_img);
    if (rc != SG_OK)
        return rc;
    pack_phase(c, c->d_out, widest, (unsigned int)widest, nullptr, 3, 1, n_img);
    HIPCHK(c, hipGetLastError());
    size_t at = 0;
    for (int i = 0; i < n_img; i++) {
        const size_t cnt = c->pending_counts[i];
        if (cnt)
            HIPCHK(c, hipMemcpyAsync(out + at, (const WireRec *)c->d_out + widest * i, cnt * sizeof(WireRec),
                                     hipMemcpyDeviceToHost, c->stream));
        at += cnt;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SG_OK;
}


// The same download in pieces the caller can wait for one by one (and expand while the next is on the bus).
int sg_extract_batch_fetch_wire_begin(sg_ctx *c, sg_wire_feature *out, size_t cap_total, size_t piece_records, int *n_pieces)
{
    if (!c || (cap_total && !out) || !n_pieces || piece_records == 0)
        return SG_E_INVALID;
    *n_pieces = 0;
    c->chunks_pending = 0;
    if (c->pending_counts.empty())
        return fail(c, SG_E_INVALID, "sg_extract_batch_fetch_wire_begin without a successful sg_extract_batch_begin before it");
    (void)hipSetDevice(c->prm.device);
    const int n_img = (int)c->pending_counts.size();
    size_t total = 0, widest = 0;
    for (uint32_t v : c->pending_counts) {
        total += v;
        widest = std::max<size_t>(widest, v);
    }
    if (total > cap_total)
        return fail(c, SG_E_CAPACITY, "output array too small");
    if (total == 0)
        return SG_OK;
    int rc = ensure_out(c, sizeof(WireRec) * widest * n_img);
    if (rc != SG_OK)
        return rc;
    pack_phase(c, c->d_out, widest, (unsigned int)widest, nullptr, 3, 1, n_img);
    HIPCHK(c, hipGetLastError());
    // Pieces follow the output order: image after image, every image cut into pieces of at most piece_records.
    size_t at = 0;
    int k = 0;
    for (int i = 0; i < n_img; i++) {
        const size_t cnt = c->pending_counts[i];
        for (size_t off = 0; off < cnt; off += piece_records) {
            const size_t m = std::min(piece_records, cnt - off);
            HIPCHK(c, hipMemcpyAsync(out + at + off, (const WireRec *)c->d_out + widest * i + off, m * sizeof(WireRec),
                                     hipMemcpyDeviceToHost, c->stream));
            if ((size_t)k >= c->chunk_ev.size()) {
                hipEvent_t e = nullptr;
                HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                c->chunk_ev.push_back(e);
            }
            HIPCHK(c, hipEventRecord(c->chunk_ev[(size_t)k], c->stream));
            k++;
        }
        at += cnt;
    }
    c->chunks_pending = k;
    *n_pieces = k;
    return SG_OK;
}

int sg_extract_batch_fetch_wire_wait(sg_ctx *c, int piece)
{
    if (!c || piece < 0 || piece >= c->chunks_pending)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    HIPCHK(c, hipEventSynchronize(c->chunk_ev[(size_t)piece]));
    return SG_OK;
}

int sg_host_alloc(size_t bytes, void **ptr)
{
    if (!ptr || bytes == 0)
        return SG_E_INVALID;
    *ptr = nullptr;
    // portable: usable by the contexts of every device of the process, whichever device is current here
    if (hipHostMalloc(ptr, bytes, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        *ptr = nullptr;
        return SG_E_NOMEM;
    }
    return SG_OK;
}

int sg_host_free(void *ptr)
{
    if (ptr && hipHostFree(ptr) != hipSuccess) {
        (void)hipGetLastError();
        return SG_E_HIP;
    }
    return SG_OK;
}

int sg_extract(sg_ctx *c, const uint8_t *px, int width, int height, int channels, size_t stride, sg_keypoint *out,
               size_t cap, size_t *n)
{
    if (!c || !px || !n || (cap && !out))
        return SG_E_INVALID;
    int rc = sg_extract_begin(c, px, width, height, channels, stride, n);
    if (rc != SG_OK)
        return rc;
    return sg_extract_fetch(c, out, cap, n);
}

// ---- inspection -------------------------------------------------------------------------------------------

int sg_last_octaves(const sg_ctx *c) { return c ? c->octaves : 0; }

unsigned int sg_last_pair_octaves(const sg_ctx *c) { return c ? c->pair_octaves : 0; }

size_t sg_match_scratch_bytes(const sg_ctx *c)
{
    if (!c)
        return 0;
    size_t bytes = c->mpart_bytes + (c->d_mbad ? sizeof(unsigned int) : 0);
    for (int i = 0; i < 2; i++)
        bytes += (size_t)c->m_cap[i] * (128 + sizeof(int));
    return bytes;
}
int sg_last_tail_octave(const sg_ctx *c) { return c ? c->tail_octave : 0; }

int sg_level_dims(const sg_ctx *c, int octave, int *rows, int *cols)
{
    if (!c || octave < 0 || octave >= c->octaves)
        return SG_E_INVALID;
    if (rows)
        *rows = c->rows[octave];
    if (cols)
        *cols = c->cols[octave];
    return SG_OK;
}

static int download_plane(sg_ctx *c, const Plane &p, int image, double *dst)
{
    const size_t n = (size_t)p.rows * p.pitch;
    std::vector<unsigned char> host(n * c->esize);
    HIPCHK(c, hipMemcpyAsync(host.data(), (const char *)p.p + (size_t)image * p.img_stride * c->esize, n * c->esize,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int y = 0; y < p.rows; y++)
        for (int x = 0; x < p.cols; x++)
            dst[(size_t)y * p.cols + x] = c->esize == 8 ? ((const double *)host.data())[(size_t)y * p.pitch + x]
                                                        : (double)((const float *)host.data())[(size_t)y * p.pitch + x];
    return SG_OK;
}

int sg_read_level(sg_ctx *c, int image, int octave, int level, double *dst)
{
    if (!c || !dst || octave < 0 || octave >= c->octaves || level < 0 || level >= kLevels || image < 0 ||
        image >= c->last_batch)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    return download_plane(c, c->level[octave][level], image, dst);
}

int sg_read_counters(sg_ctx *c, int image, sg_counters *out)
{
    if (!c || !out || image < 0 || image >= c->B)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    unsigned long long v[8];
    HIPCHK(c, hipMemcpyAsync(v, c->d_ctr + 8 * image, sizeof v, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out->cc00 = v[0]; out->cc0 = v[1]; out->cc1 = v[2]; out->cc2 = v[3]; out->cc3 = v[4];
    return SG_OK;
}

int sg_read_work_counters(sg_ctx *c, int n_img, unsigned long long *descr_trips, unsigned long long *orient_trips)
{
    if (!c || n_img < 1 || n_img > c->B)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    std::vector<unsigned long long> v(kTripWordsPerImage * n_img);
    HIPCHK(c, hipMemcpyAsync(v.data(), c->d_trips, sizeof(unsigned long long) * v.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long d = 0, o = 0;
    for (int i = 0; i < n_img; i++)
        for (int k = 0; k < kTripSlots; k++) {
            d += v[kTripWordsPerImage * i + (size_t)k * kTripStride];
            o += v[kTripWordsPerImage * i + (size_t)(kTripSlots + k) * kTripStride];
        }
    if (descr_trips)
        *descr_trips = d;
    if (orient_trips)
        *orient_trips = o;
    return SG_OK;
}

static void kprec_to_keypoint(const KpRec &r, sg_keypoint *k)
{
    memset(k, 0, sizeof *k);
    k->octave = r.octave; k->interval = r.interval; k->offset_interval = r.offi;
    k->x = r.x; k->y = r.y; k->scale = r.scale; k->dx = r.dx; k->dy = r.dy;
    k->offset_x = r.offx; k->offset_y = r.offy; k->octave_scale = r.oscale; k->val = r.val;
}

static void keypoint_to_kprec(const sg_keypoint &k, KpRec *r)
{
    r->x = k.x; r->y = k.y; r->octave = k.octave; r->interval = k.interval;
    r->offx = k.offset_x; r->offy = k.offset_y; r->offi = k.offset_interval;
    r->dx = k.dx; r->dy = k.dy; r->scale = k.scale; r->oscale = k.octave_scale; r->val = k.val;
}

static int download_keypoints(sg_ctx *c, int image, std::vector<KpRec> *recs)
{
    unsigned int nk = 0;
    HIPCHK(c, hipMemcpyAsync(&nk, c->d_counts + c->B + image, sizeof nk, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    recs->resize(nk);
    if (!nk)
        return SG_OK;
    std::vector<unsigned int> src(nk);
    HIPCHK(c, hipMemcpyAsync(src.data(), c->d_kpsrc + (size_t)image * c->feat_cap, sizeof(unsigned int) * nk,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned int hi = *std::max_element(src.begin(), src.end());
    std::vector<KpRec> all(hi + 1);
    HIPCHK(c, hipMemcpyAsync(all.data(), c->d_kp + (size_t)image * c->cand_cap, sizeof(KpRec) * (hi + 1),
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (unsigned int i = 0; i < nk; i++)
        (*recs)[i] = all[src[i]];
    return SG_OK;
}

int sg_read_keypoints(sg_ctx *c, int image, sg_keypoint *out, size_t cap, size_t *n)
{
    if (!c || !n || image < 0 || image >= c->B)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    std::vector<KpRec> recs;
    int rc = download_keypoints(c, image, &recs);
    if (rc != SG_OK)
        return rc;
    *n = recs.size();
    if (recs.size() > cap)
        return SG_E_CAPACITY;
    for (size_t i = 0; i < recs.size(); i++)
        kprec_to_keypoint(recs[i], &out[i]);
    return SG_OK;
}

int sg_read_features_unsorted(sg_ctx *c, int image, sg_keypoint *out, size_t cap, size_t *n, double *prequant)
{
    if (!c || !n || image < 0 || image >= c->B)
        return SG_E_INVALID;
    if (prequant && !c->d_pq)
        return fail(c, SG_E_INVALID, "context was created without keep_prequant");
    (void)hipSetDevice(c->prm.device);
    unsigned int nf = 0;
    HIPCHK(c, hipMemcpyAsync(&nf, c->d_counts + 2 * c->B + image, sizeof nf, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n = nf;
    if (nf > cap)
        return SG_E_CAPACITY;
    if (!nf)
        return SG_OK;
    int rc = ensure_out(c, sizeof(sg_keypoint) * (size_t)nf);
    if (rc != SG_OK)
        return rc;
    // pack image `image` alone, unsorted: shift every per-image base by hand
    {
        size_t b = (size_t)image;
        launch_pack(c->d_kp + b * c->cand_cap, 0, c->d_kpsrc + b * c->feat_cap, 0, c->d_fsrc + b * c->feat_cap, 0,
                    c->d_counts + 2 * c->B + image, c->d_ori + b * c->feat_cap * kOriBins, 0,
                    c->d_desc + b * c->feat_cap * 128, 0, c->d_rank + b * c->feat_cap, 0, c->d_out, 0, nf, nullptr,
                    c->d_overflow + 3, 1, 0, c->feat_cap, 1, c->stream);
    }
    HIPCHK(c, hipMemcpyAsync(out, c->d_out, sizeof(sg_keypoint) * (size_t)nf, hipMemcpyDeviceToHost, c->stream));
    if (prequant)
        HIPCHK(c, hipMemcpyAsync(prequant, c->d_pq + (size_t)image * c->feat_cap * 128, sizeof(double) * 128 * (size_t)nf,
                                 hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SG_OK;
}

// ---- timing ---------------------------------------------------------------------------------------------------

int sg_set_profiling(sg_ctx *c, int on)
{
    if (!c)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    if (on && !c->ev.created) {
        for (int i = 0; i < kEventPairs; i++) {
            HIPCHK(c, hipEventCreate(&c->ev.start[i]));
            HIPCHK(c, hipEventCreate(&c->ev.stop[i]));
        }
        c->ev.created = true;
    }
    if (!on)
        flush_events(c);
    c->prof = on != 0;
    return SG_OK;
}

int sg_get_stage_times(sg_ctx *c, double *ms, unsigned long long *launches)
{
    if (!c)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    flush_events(c);
    for (int i = 0; i < SG_ST_COUNT; i++) {
        if (ms)
            ms[i] = c->stage_ms[i];
        if (launches)
            launches[i] = c->stage_launches[i];
    }
    return SG_OK;
}

int sg_reset_stage_times(sg_ctx *c)
{
    if (!c)
        return SG_E_INVALID;
    flush_events(c);
    memset(c->stage_ms, 0, sizeof c->stage_ms);
    memset(c->stage_launches, 0, sizeof c->stage_launches);
    return SG_OK;
}

// ---- device memory, synthetic input -----------------------------------------------------------------------------

int sg_malloc(sg_ctx *c, size_t bytes, void **d_ptr)
{
    if (!c || !d_ptr)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    HIPCHK(c, hipMalloc(d_ptr, bytes));
    return SG_OK;
}

int sg_free(sg_ctx *c, void *d_ptr)
{
    if (!c)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(d_ptr));
    return SG_OK;
}

int sg_memcpy_h2d(sg_ctx *c, void *d_dst, const void *src, size_t bytes)
{
    if (!c)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    HIPCHK(c, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SG_OK;
}

int sg_memcpy_d2h(sg_ctx *c, void *dst, const void *d_src, size_t bytes)
{
    if (!c)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    HIPCHK(c, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SG_OK;
}

// ---- descriptor matching (demo main, sift/源.cpp:88-156) ----

static unsigned int match_pad(size_t n) { return n == 0 ? 32u : (unsigned int)((n + 31) / 32 * 32); }

// Makes room for n features in scratch set `which` and fills it from `feats` (device records of the given mode).
static int match_prepare(sg_ctx *c, int which, const void *d_feats, int mode, size_t n)
{
    const unsigned int n_pad = match_pad(n);
    if (n_pad > c->m_cap[which]) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_mrows[which])
            (void)hipFree(c->d_mrows[which]);
        if (c->d_mnorm[which])
            (void)hipFree(c->d_mnorm[which]);
        c->d_mrows[which] = nullptr;
        c->d_mnorm[which] = nullptr;
        c->m_cap[which] = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_mrows[which], (size_t)n_pad * 128));
        HIPCHK(c, hipMalloc((void **)&c->d_mnorm[which], (size_t)n_pad * sizeof(int)));
        c->m_cap[which] = n_pad;
    }
    if (!c->d_mbad) {
        HIPCHK(c, hipMalloc((void **)&c->d_mbad, sizeof(unsigned int)));
        HIPCHK(c, hipMemsetAsync(c->d_mbad, 0, sizeof(unsigned int), c->stream));
    }
    launch_match_prep(d_feats, mode, (unsigned int)n, n_pad, c->d_mrows[which], c->d_mnorm[which], c->d_mbad, c->stream);
    if (which == 0) { // room for the partial results of n features of image 1
        const size_t need = (size_t)n_pad * kMatchMaxSlices * 12;
        if (need > c->mpart_bytes) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->d_mpart)
                (void)hipFree(c->d_mpart);
            c->d_mpart = nullptr;
            c->mpart_bytes = 0;
            HIPCHK(c, hipMalloc(&c->d_mpart, need));
            c->mpart_bytes = need;
        }
    }
    return SG_OK;
}

int sg_match_packed_device(sg_ctx *c, const sg_packed_feature *d_f1, uint32_t n1, const sg_packed_feature *d_f2,
                           uint32_t n2, double ratio, int32_t *d_index2, uint32_t *d_dist2)
{
    if (!c || (n1 && (!d_f1 || !d_index2)) || (n2 && !d_f2))
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    int rc = match_prepare(c, 0, d_f1, 0, n1);
    if (rc != SG_OK)
        return rc;
    rc = match_prepare(c, 1, d_f2, 0, n2);
    if (rc != SG_OK)
        return rc;
    launch_match(c->d_mrows[0], c->d_mnorm[0], n1, c->d_mrows[1], c->d_mnorm[1], n2, ratio, d_index2, d_dist2, c->d_mpart,
                 match_slices(n1, n2), c->stream);
    HIPCHK(c, hipGetLastError());
    return SG_OK;
}

int sg_match_packed_device_noalloc(sg_ctx *c, const sg_packed_feature *d_f1, uint32_t n1,
                                   const sg_packed_feature *d_f2, uint32_t n2, double ratio, int32_t *d_index2,
                                   uint32_t *d_dist2)
{
    if (!c || (n1 && (!d_f1 || !d_index2)) || (n2 && !d_f2))
        return SG_E_INVALID;
    if (n1 > kStaticMatchMax || n2 > kStaticMatchMax)
        return fail(c, SG_E_CAPACITY, "sg_match_packed_device_noalloc takes at most 32768 features per set");
    (void)hipSetDevice(c->prm.device);
    unsigned char *sp = c->static_scratch; // resolved by sg_create on this device
    signed char *rows[2] = {(signed char *)(sp + kSsOffRows0), (signed char *)(sp + kSsOffRows1)};
    int *norms[2] = {(int *)(sp + kSsOffNorm0), (int *)(sp + kSsOffNorm1)};
    unsigned int *bad = (unsigned int *)(sp + kSsOffBad);
    // The array keeps the previous call's contents.  Rows and norms 0 .. n_pad-1 of both sets and the partial results
    // of features 0 .. n1-1 are rewritten by this call before they are read; only the flag word must start at zero.
    HIPCHK(c, hipMemsetAsync(bad, 0, sizeof(unsigned int), c->stream));
    launch_match_prep(d_f1, 0, n1, match_pad(n1), rows[0], norms[0], bad, c->stream);
    launch_match_prep(d_f2, 0, n2, match_pad(n2), rows[1], norms[1], bad, c->stream);
    launch_match(rows[0], norms[0], n1, rows[1], norms[1], n2, ratio, d_index2, d_dist2, sp + kSsOffPart,
                 match_slices(n1, n2), c->stream);
    HIPCHK(c, hipGetLastError());
    return SG_OK;
}

int sg_match(sg_ctx *c, const sg_keypoint *f1, size_t n1, const sg_keypoint *f2, size_t n2, double ratio,
             sg_match_point *out, size_t cap, size_t *n)
{
    if (!c || !n || (n1 && !f1) || (n2 && !f2) || (cap && !out) || n1 > 0x7fffffffu || n2 > 0x7fffffffu)
        return SG_E_INVALID;
    *n = 0;
    if (n1 == 0)
        return SG_OK;
    (void)hipSetDevice(c->prm.device);
    hipStream_t s = c->stream;
    sg_keypoint *d_f = nullptr;
    int *d_idx = nullptr;
    const size_t nmax = n1 > n2 ? n1 : n2;
    int rc = SG_OK;
    std::vector<int> idx(n1);
    unsigned int bad = 0;
    do {
        if (hipMalloc((void **)&d_f, sizeof(sg_keypoint) * nmax) != hipSuccess ||
            hipMalloc((void **)&d_idx, sizeof(int) * n1) != hipSuccess) {
            rc = fail(c, SG_E_HIP, "hipMalloc failed (matching)");
            break;
        }
        // the two record sets pass through one staging buffer, one after the other
        if (hipMemcpyAsync(d_f, f1, sizeof(sg_keypoint) * n1, hipMemcpyHostToDevice, s) != hipSuccess) {
            rc = fail(c, SG_E_HIP, "upload failed (matching)");
            break;
        }
        if ((rc = match_prepare(c, 0, d_f, 1, n1)) != SG_OK)
            break;
        if (n2 && hipMemcpyAsync(d_f, f2, sizeof(sg_keypoint) * n2, hipMemcpyHostToDevice, s) != hipSuccess) {
            rc = fail(c, SG_E_HIP, "upload failed (matching)");
            break;
        }
        if ((rc = match_prepare(c, 1, d_f, 1, n2)) != SG_OK)
            break;
        launch_match(c->d_mrows[0], c->d_mnorm[0], (unsigned int)n1, c->d_mrows[1], c->d_mnorm[1], (unsigned int)n2,
                     ratio, d_idx, nullptr, c->d_mpart, match_slices((unsigned int)n1, (unsigned int)n2), s);
        if (hipMemcpyAsync(idx.data(), d_idx, sizeof(int) * n1, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(&bad, c->d_mbad, sizeof bad, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            rc = fail(c, SG_E_HIP, "matching failed");
            break;
        }
    } while (0);
    if (d_f)
        (void)hipFree(d_f);
    if (d_idx)
        (void)hipFree(d_idx);
    if (rc != SG_OK)
        return rc;
    if (bad) {
        (void)hipMemsetAsync(c->d_mbad, 0, sizeof(unsigned int), s);
        return fail(c, SG_E_INVALID, "descriptor values must be integers in 0..255");
    }
    size_t m = 0;
    for (size_t i = 0; i < n1; i++) {
        if (idx[i] < 0)
            continue;
        if (m < cap) {
            out[m].index_1 = (int)i;
            out[m].index_2 = idx[i];
        }
        m++;
    }
    *n = m;
    return m > cap ? SG_E_CAPACITY : SG_OK;
}

// ---- multi-GPU exchange ------------------------------------------------------------------------------------------

int sg_comm_unique_id(void *id)
{
    if (!id)
        return SG_E_INVALID;
    return rccl_unique_id(id) == 0 ? SG_OK : SG_E_UNSUPPORTED;
}

int sg_comm_create(sg_ctx *c, const void *id, int rank, int n_ranks, void **comm)
{
    if (!c || !id || !comm || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return SG_E_INVALID;
    *comm = nullptr;
    if (const char *why = rccl_unavailable_reason())
        return fail(c, SG_E_UNSUPPORTED, why);
    (void)hipSetDevice(c->prm.device);
    const int rc = rccl_comm_create(id, rank, n_ranks, comm);
    if (rc != 0) {
        c->err = std::string("ncclCommInitRank: ") + rccl_error_string(rc);
        return SG_E_HIP;
    }
    return SG_OK;
}

int sg_comm_destroy(void *comm)
{
    if (!comm)
        return SG_OK;
    return rccl_comm_destroy(comm) == 0 ? SG_OK : SG_E_HIP;
}

int sg_compact_features(sg_ctx *c, const sg_packed_feature *d_feats, const uint32_t *d_counts, int n_img,
                        size_t cap_per_image, sg_packed_feature *d_out)
{
    if (!c || !d_feats || !d_counts || !d_out || n_img < 1)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    launch_compact_feats(d_feats, d_counts, n_img, cap_per_image, d_out, c->stream);
    HIPCHK(c, hipGetLastError());
    return SG_OK;
}

int sg_allgather_features(sg_ctx *c, void *comm, const sg_packed_feature *d_feats, const uint32_t *d_counts, int n_img,
                          size_t cap_per_image, sg_packed_feature *d_all, size_t cap_all, uint32_t *d_all_counts,
                          size_t *rank_first)
{
    if (!c || !d_feats || !d_counts || !d_all || !d_all_counts || !rank_first || n_img < 1)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    hipStream_t s = c->stream;
    int rank = 0, n_ranks = 1;
    if (comm) {
        if (const char *why = rccl_unavailable_reason())
            return fail(c, SG_E_UNSUPPORTED, why);
        const int rc = rccl_comm_shape(comm, &rank, &n_ranks);
        if (rc != 0 || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
            c->err = std::string("ncclCommCount/ncclCommUserRank: ") + rccl_error_string(rc);
            return SG_E_HIP;
        }
    }
    // 0. the ranks agree on the call's shape before any of them depends on it: a four-word header per rank (cap_all,
    // n_img, cap_per_image) is all-gathered first.  Every rank then sees the same headers and takes the same decision:
    // differing n_img or cap_per_image -> SG_E_INVALID everywhere (the counts all-gather below would not match up);
    // the capacity test of step 2 uses the smallest cap_all of all ranks, so that either every rank enters the
    // broadcasts or none does.  (A HIP or RCCL error on one rank alone cannot be agreed on: after any other non-OK
    // return the communicator must be taken as broken and destroyed.)
    size_t cap_all_min = cap_all;
    if (comm) {
        if (!c->d_xhdr)
            HIPCHK(c, hipMalloc((void **)&c->d_xhdr, sizeof(uint32_t) * 4 * (size_t)(kMaxCommRanks + 1)));
        if (n_ranks > kMaxCommRanks)
            return fail(c, SG_E_UNSUPPORTED, "more ranks than sg_allgather_features takes");
        uint32_t hdr[4] = {(uint32_t)(cap_all & 0xffffffffu), (uint32_t)((unsigned long long)cap_all >> 32), (uint32_t)n_img,
                           (uint32_t)std::min<size_t>(cap_per_image, 0xffffffffu)};
        HIPCHK(c, hipMemcpyAsync(c->d_xhdr, hdr, sizeof hdr, hipMemcpyHostToDevice, s));
        int rc = rccl_allgather_u32(comm, c->d_xhdr, c->d_xhdr + 4, 4, s);
        if (rc != 0) {
            c->err = std::string("ncclAllGather (header): ") + rccl_error_string(rc);
            return SG_E_HIP;
        }
        std::vector<uint32_t> all_hdr(4 * (size_t)n_ranks);
        HIPCHK(c, hipMemcpyAsync(all_hdr.data(), c->d_xhdr + 4, sizeof(uint32_t) * all_hdr.size(), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        for (int r = 0; r < n_ranks; r++) {
            const uint32_t *h = &all_hdr[4 * (size_t)r];
            if (h[2] != hdr[2] || h[3] != hdr[3])
                return fail(c, SG_E_INVALID, "the ranks disagree on n_img or cap_per_image");
            cap_all_min = std::min<size_t>(cap_all_min, (size_t)(((unsigned long long)h[1] << 32) | h[0]));
        }
    }
    // 1. every rank learns every frame's count
    if (comm) {
        const int rc = rccl_allgather_u32(comm, d_counts, d_all_counts, (size_t)n_img, s);
        if (rc != 0) {
            c->err = std::string("ncclAllGather: ") + rccl_error_string(rc);
            return SG_E_HIP;
        }
    } else {
        HIPCHK(c, hipMemcpyAsync(d_all_counts, d_counts, sizeof(uint32_t) * n_img, hipMemcpyDeviceToDevice, s));
    }
    std::vector<uint32_t> counts((size_t)n_ranks * n_img);
    HIPCHK(c, hipMemcpyAsync(counts.data(), d_all_counts, sizeof(uint32_t) * counts.size(), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    // 2. the layout all ranks derive from the same counts: rank r's features start at rank_first[r]
    std::vector<size_t> first_byte(n_ranks + 1);
    rank_first[0] = 0;
    for (int r = 0; r < n_ranks; r++) {
        size_t total = 0;
        for (int i = 0; i < n_img; i++)
            total += std::min<size_t>(counts[(size_t)r * n_img + i], cap_per_image);
        rank_first[r + 1] = rank_first[r] + total;
    }
    for (int r = 0; r <= n_ranks; r++)
        first_byte[r] = rank_first[r] * sizeof(sg_packed_feature);
    if (rank_first[n_ranks] > cap_all_min) // the same verdict on every rank
        return fail(c, SG_E_CAPACITY, "gathered features do not fit the smallest cap_all of the ranks");
    // 3. this rank's features into its own slot, without padding; 4. every slot from its owner
    launch_compact_feats(d_feats, d_counts, n_img, cap_per_image, d_all + rank_first[rank], s);
    HIPCHK(c, hipGetLastError());
    if (comm && n_ranks > 1) {
        const int rc = rccl_allgatherv_bytes(comm, reinterpret_cast<char *>(d_all), first_byte.data(), rank, n_ranks, s);
        if (rc != 0) {
            c->err = std::string("all-gather-v (ncclBroadcast or ncclSend/ncclRecv group): ") + rccl_error_string(rc);
            return SG_E_HIP;
        }
    }
    return SG_OK;
}

int sg_synth_vnoise5(sg_ctx *c, uint8_t *d_dst, int width, int height, uint64_t seed)
{
    if (!c || !d_dst || width < 1 || height < 1)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    launch_vnoise5(d_dst, height, width, seed, c->stream);
    HIPCHK(c, hipGetLastError());
    return SG_OK;
}

} // extern "C"

// ===================================================================== stage entry points (host pointers)
//
// Each uploads its operands into temporary device planes, runs the same kernels the whole path uses
// (batch of one) and downloads the result.  They exist for drop-in use of the reference's stage-level
// functions and for parity tests of single stages; they allocate and synchronise freely.

namespace {

struct TmpPlane {
    Plane pl;
    sg_ctx *c;
    TmpPlane(sg_ctx *ctx, int rows, int cols) : c(ctx)
    {
        pl.rows = rows;
        pl.cols = cols;
        pl.pitch = pitch_of(cols);
        pl.img_stride = (size_t)rows * pl.pitch;
        if (hipMalloc(&pl.p, c->esize * (pl.img_stride + 64)) != hipSuccess)
            pl.p = nullptr;
    }
    ~TmpPlane()
    {
        if (pl.p)
            (void)hipFree(pl.p);
    }
    bool ok() const { return pl.p != nullptr; }
    int upload(const double *src)
    {
        std::vector<unsigned char> host(c->esize * pl.img_stride, 0);
        for (int y = 0; y < pl.rows; y++)
            for (int x = 0; x < pl.cols; x++) {
                double v = src[(size_t)y * pl.cols + x];
                if (c->esize == 8)
                    ((double *)host.data())[(size_t)y * pl.pitch + x] = v;
                else
                    ((float *)host.data())[(size_t)y * pl.pitch + x] = (float)v;
            }
        HIPCHK(c, hipMemcpyAsync(pl.p, host.data(), host.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return SG_OK;
    }
    int download(double *dst) { return download_plane(c, pl, 0, dst); }
};

int finish(sg_ctx *c)
{
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SG_OK;
}

int blur_tmp(sg_ctx *c, const Plane &src, const Plane &dst, double sigma)
{
    std::vector<double> taps(1024);
    int n = gaussian_taps(sigma, taps.data(), 1024);
    if (n < 0)
        return fail(c, SG_E_UNSUPPORTED, "sigma too large");
    if (c->fused_blur && launch_blur_fused(src, dst, 1, taps.data(), n, c->blur_prec(), c->stream))
        return finish(c);
    TmpPlane tmp(c, src.rows, src.cols);
    if (!tmp.ok())
        return fail(c, SG_E_NOMEM, "out of device memory");
    if (!launch_blur_row(src, tmp.pl, 1, taps.data(), n, c->blur_prec(), c->stream) ||
        !launch_blur_col(tmp.pl, dst, 1, taps.data(), n, c->blur_prec(), c->stream))
        return fail(c, SG_E_NOMEM, "out of device memory (blur taps)");
    return finish(c);
}

// uploads a pyramid of `per_octave` levels per octave and fills an OctaveTab over it
struct TmpPyramid {
    std::vector<TmpPlane *> planes;
    OctaveTab tab;
    unsigned int n_words = 0;
    ~TmpPyramid()
    {
        for (TmpPlane *p : planes)
            delete p;
    }
    int build(sg_ctx *c, const double *const *levels, int width, int height, int octaves, int per_octave, bool from_dog)
    {
        Geometry g;
        int rc = make_geometry(height, width, octaves, &g);
        if (rc != SG_OK)
            return fail(c, rc, "too many octaves");
        memset(&tab, 0, sizeof tab);
        for (int o = 0; o < octaves; o++) {
            for (int l = 0; l < per_octave; l++) {
                TmpPlane *p = new TmpPlane(c, g.rows[o], g.cols[o]);
                planes.push_back(p);
                if (!p->ok())
                    return fail(c, SG_E_NOMEM, "out of device memory");
                rc = p->upload(levels[o * per_octave + l]);
                if (rc != SG_OK)
                    return rc;
                tab.lv[o][l] = p->pl.p;
            }
            tab.rows[o] = g.rows[o];
            tab.cols[o] = g.cols[o];
            tab.pitch[o] = g.pitch[o];
            tab.img_stride[o] = 0;
            tab.nrows[o] = g.nrows[o];
            tab.wpr[o] = g.wpr[o];
        }
        for (int o = 0; o <= kMaxOctaves; o++)
            tab.wbase[o] = g.wbase[o];
        tab.octaves = octaves;
        tab.from_dog = from_dog ? 1 : 0;
        n_words = g.wbase[octaves];
        return SG_OK;
    }
};

} // namespace

extern "C" {

int sg_convert_to_gray(sg_ctx *c, const uint8_t *px, int width, int height, int channels, size_t stride, double *dst)
{
    if (!c || !px || !dst || width < 1 || height < 1 || channels < 3 || stride < (size_t)width * channels)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    int rc = ensure_in(c, stride * (size_t)height);
    if (rc != SG_OK)
        return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_in, px, image_bytes(width, height, channels, stride), hipMemcpyHostToDevice, c->stream));
    TmpPlane out(c, height, width);
    if (!out.ok())
        return fail(c, SG_E_NOMEM, "out of device memory");
    launch_gray(c->d_in, height, width, channels, stride, 0, out.pl, 1, c->prm.precision, c->stream);
    rc = finish(c);
    return rc != SG_OK ? rc : out.download(dst);
}

int sg_upsample(sg_ctx *c, const double *src, int width, int height, double *dst)
{
    if (!c || !src || !dst || width < 2 || height < 2)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    TmpPlane in(c, height, width), out(c, 2 * height, 2 * width);
    if (!in.ok() || !out.ok())
        return fail(c, SG_E_NOMEM, "out of device memory");
    int rc = in.upload(src);
    if (rc != SG_OK)
        return rc;
    launch_upsample(in.pl, out.pl, 1, c->prm.precision, c->stream);
    rc = finish(c);
    return rc != SG_OK ? rc : out.download(dst);
}

int sg_downsample(sg_ctx *c, const double *src, int width, int height, double *dst)
{
    if (!c || !src || !dst || width < 1 || height < 1)
        return SG_E_INVALID;
    if (width <= 1 || height <= 1) { // DownSample copies such a source (sift.cpp:64-68)
        memcpy(dst, src, sizeof(double) * (size_t)width * height);
        return SG_OK;
    }
    (void)hipSetDevice(c->prm.device);
    TmpPlane in(c, height, width), out(c, height / 2, width / 2);
    if (!in.ok() || !out.ok())
        return fail(c, SG_E_NOMEM, "out of device memory");
    int rc = in.upload(src);
    if (rc != SG_OK)
        return rc;
    launch_downsample(in.pl, out.pl, 1, c->prm.precision, c->stream);
    rc = finish(c);
    return rc != SG_OK ? rc : out.download(dst);
}

int sg_gaussian_smooth(sg_ctx *c, const double *src, double *dst, int width, int height, double sigma)
{
    if (!c || !src || !dst || width < 1 || height < 1 || !(sigma > 0))
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    TmpPlane in(c, height, width), out(c, height, width);
    if (!in.ok() || !out.ok())
        return fail(c, SG_E_NOMEM, "out of device memory");
    int rc = in.upload(src);
    if (rc != SG_OK)
        return rc;
    rc = blur_tmp(c, in.pl, out.pl, sigma);
    return rc != SG_OK ? rc : out.download(dst);
}

int sg_create_init_smooth_gray(sg_ctx *c, const uint8_t *px, int width, int height, int channels, size_t stride,
                               double sigma, double *dst)
{
    if (!c || !px || !dst || width < 2 || height < 2 || channels < 3 || !(sigma > 1.0) ||
        stride < (size_t)width * channels)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    int rc = ensure_in(c, stride * (size_t)height);
    if (rc != SG_OK)
        return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_in, px, image_bytes(width, height, channels, stride), hipMemcpyHostToDevice, c->stream));
    TmpPlane up(c, 2 * height, 2 * width), out(c, 2 * height, 2 * width);
    if (!up.ok() || !out.ok())
        return fail(c, SG_E_NOMEM, "out of device memory");
    launch_gray_upsample(c->d_in, height, width, channels, stride, 0, up.pl, 1, c->prm.precision, c->stream);
    double sigma_init = sqrt(sigma * sigma - (0.5 * 2) * (0.5 * 2));
    rc = blur_tmp(c, up.pl, out.pl, sigma_init);
    return rc != SG_OK ? rc : out.download(dst);
}

int sg_gaussian_pyramid(sg_ctx *c, const double *base, int width, int height, int octaves, int intervals, double sigma,
                        double *const *levels)
{
    if (!c || !base || !levels || width < 1 || height < 1 || octaves < 0)
        return SG_E_INVALID;
    if (intervals != 3)
        return SG_E_UNSUPPORTED;
    (void)hipSetDevice(c->prm.device);
    Geometry g;
    int rc = make_geometry(height, width, octaves, &g);
    if (rc != SG_OK)
        return rc;
    double sig[kLevels];
    double k = pow(2.0, 1.0 / intervals);
    sig[0] = sigma;
    for (int i = 1; i < kLevels; i++) {
        double sig_prev = pow(k, i - 1) * sigma;
        double sig_total = sig_prev * k;
        sig[i] = sqrt(sig_total * sig_total - sig_prev * sig_prev);
    }
    TmpPlane *prev3 = nullptr;
    std::vector<TmpPlane *> keep;
    auto cleanup = [&]() {
        for (TmpPlane *p : keep)
            delete p;
    };
    for (int o = 0; o < octaves; o++) {
        TmpPlane *lv[kLevels];
        for (int l = 0; l < kLevels; l++) {
            lv[l] = new TmpPlane(c, g.rows[o], g.cols[o]);
            keep.push_back(lv[l]);
            if (!lv[l]->ok()) {
                cleanup();
                return fail(c, SG_E_NOMEM, "out of device memory");
            }
        }
        if (o == 0)
            rc = lv[0]->upload(base);
        else if (g.rows[o] == g.rows[o - 1]) {
            rc = hipMemcpy(lv[0]->pl.p, prev3->pl.p, c->esize * lv[0]->pl.img_stride, hipMemcpyDeviceToDevice) == hipSuccess
                     ? SG_OK : SG_E_HIP;
        } else {
            launch_downsample(prev3->pl, lv[0]->pl, 1, c->prm.precision, c->stream);
            rc = finish(c);
        }
        for (int l = 1; l < kLevels && rc == SG_OK; l++)
            rc = blur_tmp(c, lv[l - 1]->pl, lv[l]->pl, sig[l]);
        for (int l = 0; l < kLevels && rc == SG_OK; l++)
            rc = lv[l]->download(levels[o * kLevels + l]);
        if (rc != SG_OK) {
            cleanup();
            return rc;
        }
        prev3 = lv[3];
    }
    cleanup();
    return SG_OK;
}

int sg_sub(sg_ctx *c, const double *a, const double *b, double *out, int width, int height)
{
    if (!c || !a || !b || !out || width < 1 || height < 1)
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    TmpPlane pa(c, height, width), pb(c, height, width), pc(c, height, width);
    if (!pa.ok() || !pb.ok() || !pc.ok())
        return fail(c, SG_E_NOMEM, "out of device memory");
    int rc = pa.upload(a);
    if (rc == SG_OK)
        rc = pb.upload(b);
    if (rc != SG_OK)
        return rc;
    launch_sub(pa.pl, pb.pl, pc.pl, 1, c->prm.precision, c->stream);
    rc = finish(c);
    return rc != SG_OK ? rc : pc.download(out);
}

int sg_dog_pyramid(sg_ctx *c, const double *const *gauss, int width, int height, int octaves, int intervals,
                   double *const *dog)
{
    if (!c || !gauss || !dog || octaves < 0)
        return SG_E_INVALID;
    if (intervals != 3)
        return SG_E_UNSUPPORTED;
    Geometry g;
    int rc = make_geometry(height, width, octaves, &g);
    if (rc != SG_OK)
        return rc;
    for (int o = 0; o < octaves; o++)
        for (int i = 1; i < kLevels; i++) {
            rc = sg_sub(c, gauss[o * kLevels + i], gauss[o * kLevels + i - 1], dog[o * kDogLevels + i - 1], g.cols[o],
                        g.rows[o]);
            if (rc != SG_OK)
                return rc;
        }
    return SG_OK;
}

int sg_detect_local_extrema(sg_ctx *c, const double *const *dog, int width, int height, int octaves, int intervals,
                            int nointerval, sg_keypoint *out, size_t cap, size_t *n, sg_counters *counters)
{
    if (!c || !dog || !n || octaves < 0 || (cap && !out))
        return SG_E_INVALID;
    if (intervals != 3)
        return SG_E_UNSUPPORTED;
    (void)hipSetDevice(c->prm.device);
    TmpPyramid pyr;
    int rc = pyr.build(c, dog, width, height, octaves, kDogLevels, true);
    if (rc != SG_OK)
        return rc;
    if (pyr.n_words > c->words_img)
        return fail(c, SG_E_CAPACITY, "pyramid larger than the context was sized for");
    invalidate_pending(c);
    detect_phase(c, pyr.tab, pyr.n_words, 1, nointerval);
    rc = check_overflow(c);
    if (rc != SG_OK)
        return rc;
    std::vector<KpRec> recs;
    rc = download_keypoints(c, 0, &recs);
    if (rc != SG_OK)
        return rc;
    if (counters)
        sg_read_counters(c, 0, counters);
    *n = recs.size();
    if (recs.size() > cap)
        return SG_E_CAPACITY;
    for (size_t i = 0; i < recs.size(); i++) {
        kprec_to_keypoint(recs[i], &out[i]);
        // the reference's detector leaves scale fields unset and keeps base-image coordinates (sift.cpp:895-907)
        out[i].dx = (out[i].x + out[i].offset_x) * pow(2.0, out[i].octave);
        out[i].dy = (out[i].y + out[i].offset_y) * pow(2.0, out[i].octave);
        out[i].scale = 0;
        out[i].octave_scale = 0;
    }
    return SG_OK;
}

int sg_calculate_scale(sg_keypoint *kp, size_t n, double sigma, int intervals)
{
    if (!kp && n)
        return SG_E_INVALID;
    for (size_t i = 0; i < n; i++) {
        double intvl = kp[i].interval + kp[i].offset_interval;
        // the routine the device path uses (2^y correctly rounded): the stage chain gives the driver's records exactly
        kp[i].scale = sigma * sg::sg_exp2_cr(kp[i].octave + intvl / intervals);
        kp[i].octave_scale = sigma * sg::sg_exp2_cr(intvl / intervals);
    }
    return SG_OK;
}

int sg_half_features(sg_keypoint *kp, size_t n)
{
    if (!kp && n)
        return SG_E_INVALID;
    for (size_t i = 0; i < n; i++) {
        kp[i].dx /= 2;
        kp[i].dy /= 2;
        kp[i].scale /= 2;
    }
    return SG_OK;
}

// Records come from the caller (hand-built, or read from a text file that leaves most fields zero): the kernels index
// level [octave][interval] at (x, y) and size their windows from octave_scale, so everything they index by is checked.
static int upload_records(sg_ctx *c, const sg_keypoint *recs, size_t n, bool with_ori, const OctaveTab &tab)
{
    if (n > c->feat_cap)
        return fail(c, SG_E_CAPACITY, "more records than the context's feature capacity");
    for (size_t i = 0; i < n; i++) {
        const sg_keypoint &k = recs[i];
        const bool ok = k.octave >= 0 && k.octave < tab.octaves && k.interval >= 0 && k.interval < kLevels &&
                        k.x >= 0 && k.x < tab.cols[k.octave] && k.y >= 0 && k.y < tab.rows[k.octave] &&
                        k.octave_scale > 0 && k.octave_scale <= 256.0 && (!with_ori || std::isfinite(k.ori));
        if (!ok) // (a NaN octave_scale fails both comparisons)
            return fail(c, SG_E_INVALID, "record with octave, interval, x, y, octave_scale or ori outside the pyramid's range");
    }
    std::vector<KpRec> kp(n);
    std::vector<unsigned int> ident(n), fsrc(n);
    std::vector<double> ori(with_ori ? n * kOriBins : 0);
    for (size_t i = 0; i < n; i++) {
        keypoint_to_kprec(recs[i], &kp[i]);
        ident[i] = (unsigned int)i;
        fsrc[i] = (unsigned int)i * 64u;
        if (with_ori)
            ori[i * kOriBins] = recs[i].ori;
    }
    unsigned int cnt = (unsigned int)n;
    hipStream_t s = c->stream;
    if (n) {
        HIPCHK(c, hipMemcpyAsync(c->d_kp, kp.data(), sizeof(KpRec) * n, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->d_kpsrc, ident.data(), sizeof(unsigned int) * n, hipMemcpyHostToDevice, s));
        if (with_ori) {
            HIPCHK(c, hipMemcpyAsync(c->d_fsrc, fsrc.data(), sizeof(unsigned int) * n, hipMemcpyHostToDevice, s));
            HIPCHK(c, hipMemcpyAsync(c->d_ori, ori.data(), sizeof(double) * ori.size(), hipMemcpyHostToDevice, s));
        }
    }
    HIPCHK(c, hipMemcpyAsync(c->d_counts + c->B, &cnt, sizeof cnt, hipMemcpyHostToDevice, s));
    if (with_ori)
        HIPCHK(c, hipMemcpyAsync(c->d_counts + 2 * c->B, &cnt, sizeof cnt, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return SG_OK;
}

int sg_orientation_assignment(sg_ctx *c, const sg_keypoint *extrema, size_t n_extrema, const double *const *gauss,
                              int width, int height, int octaves, sg_keypoint *out, size_t cap, size_t *n)
{
    if (!c || !gauss || !n || (n_extrema && !extrema) || (cap && !out))
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    TmpPyramid pyr;
    int rc = pyr.build(c, gauss, width, height, octaves, kLevels, false);
    if (rc != SG_OK)
        return rc;
    rc = upload_records(c, extrema, n_extrema, false, pyr.tab);
    if (rc != SG_OK)
        return rc;
    (void)hipMemsetAsync(c->d_chunks + c->chunks1 + c->chunks2, 0, sizeof(unsigned int) * c->chunks3, c->stream);
    int max_radius = 0;
    for (size_t i = 0; i < n_extrema; i++)
        max_radius = std::max(max_radius, ori_radius(extrema[i].octave_scale));
    invalidate_pending(c);
    orient_phase(c, pyr.tab, 1, max_radius);
    rc = check_overflow(c);
    if (rc != SG_OK)
        return rc;
    unsigned int nf = 0;
    HIPCHK(c, hipMemcpy(&nf, c->d_counts + 2 * c->B, sizeof nf, hipMemcpyDeviceToHost));
    *n = nf;
    if (nf > cap)
        return SG_E_CAPACITY;
    if (!nf)
        return SG_OK;
    std::vector<unsigned int> fsrc(nf);
    std::vector<double> ori((size_t)n_extrema * kOriBins);
    HIPCHK(c, hipMemcpy(fsrc.data(), c->d_fsrc, sizeof(unsigned int) * nf, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ori.data(), c->d_ori, sizeof(double) * ori.size(), hipMemcpyDeviceToHost));
    for (unsigned int f = 0; f < nf; f++) {
        unsigned int j = fsrc[f] >> 6, bin = fsrc[f] & 63;
        out[f] = extrema[j]; // CopyKeypoint (sift.cpp:1249-1267); descriptor fields are not meaningful yet
        out[f].descr_length = 0;
        memset(out[f].descriptor, 0, sizeof out[f].descriptor);
        out[f].ori = ori[(size_t)j * kOriBins + bin];
    }
    return SG_OK;
}

int sg_descriptor_representation(sg_ctx *c, sg_keypoint *features, size_t n, const double *const *gauss, int width,
                                 int height, int octaves)
{
    if (!c || !gauss || (n && !features))
        return SG_E_INVALID;
    (void)hipSetDevice(c->prm.device);
    TmpPyramid pyr;
    int rc = pyr.build(c, gauss, width, height, octaves, kLevels, false);
    if (rc != SG_OK)
        return rc;
    rc = upload_records(c, features, n, true, pyr.tab);
    if (rc != SG_OK)
        return rc;
    int max_radius = 0;
    for (size_t i = 0; i < n; i++)
        max_radius = std::max(max_radius, descr_radius(features[i].octave_scale));
    // The descriptor kernel adds into 64-bit fixed-point bins that hold values below 2^14 (kernels_describe.hip).  The
    // path's own pyramids (pixel values 0..1, window radius at most 42) stay below 300; planes and scales handed in
    // from outside are checked.  A bin's sum is bounded by (sum of a cell's bilinear weights over the sample grid,
    // about shw^2 = (r / 3.54)^2, taken as 0.1 r^2) x (largest gradient magnitude, sqrt(2) x the pixel range); the
    // planes must be finite.
    if (n) {
        Geometry g;
        if (make_geometry(height, width, octaves, &g) != SG_OK)
            return fail(c, SG_E_UNSUPPORTED, "too many octaves");
        double vmin = 1.7976931348623157e308, vmax = -1.7976931348623157e308;
        bool finite = true;
        for (int o = 0; o < octaves; o++)
            for (int l = 1; l <= 3; l++) { // the levels descriptors read (interval 1..3)
                const double *pl = gauss[o * kLevels + l];
                const size_t npx = (size_t)g.rows[o] * g.cols[o];
                for (size_t i = 0; i < npx; i++) {
                    const double v = pl[i];
                    finite = finite && fabs(v) <= 1.7976931348623157e308; // false for NaN and infinities
                    vmin = v < vmin ? v : vmin;
                    vmax = v > vmax ? v : vmax;
                }
            }
        if (!finite)
            return fail(c, SG_E_INVALID, "the pyramid holds values that are not finite");
        if (vmax > vmin && 0.1 * (double)max_radius * (double)max_radius * 1.4142135623730951 * (vmax - vmin) >= 16000.0)
            return fail(c, SG_E_UNSUPPORTED,
                        "pixel range times window area beyond what the descriptor histograms hold ((max - min pixel) * "
                        "radius^2 must stay below 113000; the path's own pyramids have pixels 0..1 and radius <= 42)");
    }
    invalidate_pending(c);
    descr_phase(c, pyr.tab, 1, max_radius);
    rc = finish(c);
    if (rc != SG_OK || !n)
        return rc;
    std::vector<uint8_t> desc(n * 128);
    HIPCHK(c, hipMemcpy(desc.data(), c->d_desc, desc.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
        features[i].descr_length = 128;
        for (int k = 0; k < 128; k++)
            features[i].descriptor[k] = desc[i * 128 + k];
    }
    return SG_OK;
}

int sg_sort_features(sg_ctx *c, sg_keypoint *features, size_t n)
{
    if (!c || (n && !features))
        return SG_E_INVALID;
    if (n > c->feat_cap)
        return fail(c, SG_E_CAPACITY, "more records than the context's feature capacity");
    if (n < 2)
        return SG_OK;
    (void)hipSetDevice(c->prm.device);
    std::vector<double> sc(n);
    for (size_t i = 0; i < n; i++)
        sc[i] = features[i].scale;
    unsigned int cnt = (unsigned int)n;
    HIPCHK(c, hipMemcpyAsync(c->d_fscale, sc.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_counts + 2 * c->B, &cnt, sizeof cnt, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    invalidate_pending(c);
    rank_phase(c, 1);
    int rc = finish(c);
    if (rc != SG_OK)
        return rc;
    std::vector<unsigned int> rank(n);
    HIPCHK(c, hipMemcpy(rank.data(), c->d_rank, sizeof(unsigned int) * n, hipMemcpyDeviceToHost));
    std::vector<sg_keypoint> tmp(features, features + n);
    for (size_t i = 0; i < n; i++)
        features[rank[i]] = tmp[i];
    return SG_OK;
}

} // extern "C"
