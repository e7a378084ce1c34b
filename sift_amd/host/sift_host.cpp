// sift_host.cpp -- host C++ layer: the reference's free functions (sift/sift.h, sift/sift.cpp) implemented
// as thin calls into the C ABI of libsiftgpu.  No image arithmetic happens here; this file only
// marshals cv::Mat / std::vector arguments, manages one GPU context per thread and reproduces the
// reference's calling conventions (void returns, outputs appended, silent return on bad arguments).
#include "sift_gpu.h"

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>

#include "siftgpu.h"

static_assert(sizeof(Keypoint) == sizeof(sg_keypoint), "Keypoint must match the C ABI record");

namespace {

struct ThreadState {
    sg_ctx *ctx = nullptr;
    int max_w = 0, max_h = 0;
    double sigma = 0;
    int nointerval = -1;
    int precision = -1; // the sg_precision the context was built with
    size_t max_features = 0; // 0: the library's default density; raised after an image overflowed it
    int device = 0;
    int status = SG_OK;
    std::string error;
    bool verbose = false;
    sg_wire_feature *wire = nullptr; // page-locked landing area of the feature records of a call
    size_t wire_cap = 0;
    ~ThreadState()
    {
        if (wire)
            sg_host_free(wire);
        if (ctx)
            sg_destroy(ctx);
    }
};

// The pyramid arithmetic of contexts created from now on (SiftSetPrecision; -1: the environment decides on first use).
std::atomic<int> g_precision(-1);
int precision_now()
{
    int p = g_precision.load();
    if (p < 0) {
        const char *e = std::getenv("SIFT_GPU_PRECISION");
        p = SG_F64;
        if (e && (!std::strcmp(e, "fused") || !std::strcmp(e, "f64_fma")))
            p = SG_F64_FMA;
        else if (e && (!std::strcmp(e, "single") || !std::strcmp(e, "f32")))
            p = SG_F32;
        int expected = -1;
        g_precision.compare_exchange_strong(expected, p);
        p = g_precision.load();
    }
    return p;
}

// Wire records into the caller's vector, behind what it holds.  resize() would zero 1120 bytes per new record before
// they are filled (a Keypoint is a plain struct); instead the records are expanded a few dozen at a time into a buffer
// that stays in cache and appended from there, so that the vector's memory is written once.
void append_expanded(std::vector<Keypoint> &features, const sg_wire_feature *w, size_t n)
{
    features.reserve(features.size() + n);
    static thread_local Keypoint block[32];
    for (size_t i = 0; i < n; i += 32) {
        const size_t m = n - i < 32 ? n - i : 32;
        sg_expand_wire(w + i, m, reinterpret_cast<sg_keypoint *>(block));
        features.insert(features.end(), block, block + m);
    }
}

thread_local ThreadState tls;

void set_status(int rc, const char *what = nullptr)
{
    tls.status = rc;
    if (rc == SG_OK) {
        tls.error.clear();
        return;
    }
    tls.error = what ? what : "";
    if (tls.ctx && tls.error.empty())
        tls.error = sg_last_error(tls.ctx);
    if (tls.error.empty())
        tls.error = sg_strerror(rc);
}

// A context able to take input images up to w x h (i.e. base planes up to 2w x 2h) with room for at least
// min_features features per image (0: whatever the current context has).
sg_ctx *context_for(int w, int h, double sigma, int nointerval, size_t min_features = 0)
{
    if (w < 2)
        w = 2;
    if (h < 2)
        h = 2;
    const int precision = precision_now();
    const bool fits = tls.ctx && w <= tls.max_w && h <= tls.max_h && sigma == tls.sigma &&
                      nointerval == tls.nointerval && precision == tls.precision && min_features <= tls.max_features;
    if (fits)
        return tls.ctx;
    if (min_features > tls.max_features)
        tls.max_features = min_features;
    if (tls.ctx) {
        sg_destroy(tls.ctx);
        tls.ctx = nullptr;
    }
    sg_params p;
    sg_default_params(&p);
    p.sigma = sigma;
    p.nointerval = nointerval;
    p.precision = precision;
    p.max_width = w > tls.max_w ? w : tls.max_w;
    p.max_height = h > tls.max_h ? h : tls.max_h;
    p.max_batch = 1;
    p.device = tls.device;
    p.max_features = tls.max_features > 0x7fffffff ? 0x7fffffff : (int)tls.max_features;
    int rc = sg_create(&p, &tls.ctx);
    if (rc != SG_OK) {
        tls.ctx = nullptr;
        tls.max_w = tls.max_h = 0;
        set_status(rc, "sg_create failed (is a HIP device present? there is no CPU fallback)");
        return nullptr;
    }
    tls.max_w = p.max_width;
    tls.max_h = p.max_height;
    tls.sigma = sigma;
    tls.nointerval = nointerval;
    tls.precision = precision;
    return tls.ctx;
}

// After a call returned SG_E_CAPACITY: if an internal buffer of the context overflowed (an image denser than the
// default sizing assumes; the reference has no such limit), a context sized from the reported counts replaces it
// and true is returned: the caller runs the call again.  False when the overflow was the caller's own output array.
bool grow_context(int w, int h, double sigma, int nointerval)
{
    size_t need = 0;
    if (!tls.ctx || sg_required_features(tls.ctx, &need) != SG_OK || need == 0 || need <= tls.max_features)
        return false;
    return context_for(w, h, sigma, nointerval, need) != nullptr;
}

// context for stage calls on a plane of cols x rows used as (at most) a base image
sg_ctx *context_for_plane(int cols, int rows)
{
    double sigma = tls.ctx ? tls.sigma : SIGMA;
    int ni = tls.ctx ? tls.nointerval : 0;
    return context_for((cols + 1) / 2, (rows + 1) / 2, sigma, ni);
}

bool is_f64_plane(const Mat &m) { return !m.empty() && m.type() == CV_64FC1; }

// contiguous double view of a CV_64FC1 Mat (copies only when the rows are padded)
struct PlaneView {
    const double *p = nullptr;
    std::vector<double> copy;
    explicit PlaneView(const Mat &m)
    {
        if ((size_t)m.step == (size_t)m.cols * sizeof(double)) {
            p = reinterpret_cast<const double *>(m.data);
        } else {
            copy.resize((size_t)m.rows * m.cols);
            for (int y = 0; y < m.rows; y++)
                std::memcpy(&copy[(size_t)y * m.cols], m.data + m.step * (size_t)y, sizeof(double) * (size_t)m.cols);
            p = copy.data();
        }
    }
};

void octave_dims(int rows, int cols, int octaves, std::vector<int> &r, std::vector<int> &c)
{
    r.resize(octaves);
    c.resize(octaves);
    for (int o = 0; o < octaves; o++) {
        if (o == 0) {
            r[0] = rows;
            c[0] = cols;
        } else if (r[o - 1] <= 1 || c[o - 1] <= 1) {
            r[o] = r[o - 1];
            c[o] = c[o - 1];
        } else {
            r[o] = r[o - 1] / 2;
            c[o] = c[o - 1] / 2;
        }
    }
}

// ---- one context shared by the calling threads ---------------------------------------------------------------
// With SiftSetBatching(n > 1) the Sift() calls of all threads on a device go through one context of n images per batch.
// A call copies its image into a page-locked slot and queues a request; whichever caller finds the context idle
// becomes the leader: it takes the queued requests of one image size (its own need not be among them), runs them as
// one batch and hands every request its slice of the downloaded wire records; each caller then expands its own
// slice into its own vector while the next leader already runs the next batch.  Nobody waits for a batch to fill:
// a caller alone gets a batch of one, sixteen busy callers settle into batches of about eight.
struct Request {
    const uint8_t *px = nullptr; // page-locked, rows back to back
    int w = 0, h = 0, ch = 0;
    double sigma = 0;
    int nointerval = 0;
    bool want_counters = false;
    bool solo = false; // a batch this request was part of failed: it runs again as a batch of one
    // filled by the leader
    bool done = false;
    int rc = SG_OK;
    std::string error;
    const sg_wire_feature *wire = nullptr;
    size_t n = 0;
    int wire_buf = -1;
    sg_counters counters = {};
    int octaves = 0;
};

class Batcher {
public:
    Batcher(int device, int max_batch) : device_(device), B_(max_batch), slots_(2 * (size_t)max_batch) {}

    // a page-locked buffer of at least `bytes` for one caller's image; waits while all are taken
    int acquire_slot(size_t bytes, uint8_t **p)
    {
        std::unique_lock<std::mutex> lk(m_);
        size_t i;
        for (;;) {
            for (i = 0; i < slots_.size() && slots_[i].used; i++) {
            }
            if (i < slots_.size())
                break;
            cv_.wait(lk);
        }
        slots_[i].used = true;
        lk.unlock(); // the slot is this caller's alone from here
        if (slots_[i].cap < bytes) {
            if (slots_[i].p)
                sg_host_free(slots_[i].p);
            slots_[i].p = nullptr;
            slots_[i].cap = 0;
            void *q = nullptr;
            if (sg_host_alloc(bytes + bytes / 8, &q) != SG_OK) {
                release_slot((int)i);
                return -1;
            }
            slots_[i].p = static_cast<uint8_t *>(q);
            slots_[i].cap = bytes + bytes / 8;
        }
        *p = slots_[i].p;
        return (int)i;
    }

    void release_slot(int i)
    {
        std::lock_guard<std::mutex> lk(m_);
        slots_[(size_t)i].used = false;
        cv_.notify_all();
    }

    void release_wire(int b)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (--wire_[b].users == 0)
            cv_.notify_all();
    }

    // queues r and returns once it has been served
    void run(Request &r)
    {
        std::unique_lock<std::mutex> lk(m_);
        queue_.push_back(&r);
        while (!r.done) {
            if (busy_) {
                cv_.wait(lk);
                continue;
            }
            busy_ = true;
            std::vector<Request *> batch;
            const Request *head = queue_.front();
            for (auto it = queue_.begin(); it != queue_.end() && (int)batch.size() < B_;) {
                const Request *q = *it;
                const bool joins = q == head || (!head->solo && !q->solo && q->w == head->w && q->h == head->h &&
                                                 q->ch == head->ch && q->sigma == head->sigma &&
                                                 q->nointerval == head->nointerval);
                if (joins) {
                    batch.push_back(*it);
                    it = queue_.erase(it);
                } else {
                    ++it;
                }
            }
            lk.unlock();
            int wb = -1;
            std::vector<Request *> back; // requests that run again: queued in front of the others
            failed_ = false;
            try {
                wb = execute(batch, back);
            } catch (...) { // out of host memory in the bookkeeping: the batch fails, the context stays usable
                fail_all(batch, SG_E_NOMEM, "out of host memory");
                for (Request *q : batch)
                    q->wire_buf = -1;
                wb = held_wire_;
            }
            held_wire_ = -1;
            // A batch of several images failed (one image too dense for any context, no memory left for the whole
            // batch, ...): the reference serves every call on its own, so one caller's image must not fail another's
            // call.  Every request of the batch runs again as a batch of one; only the offender keeps the error.
            std::vector<Request *> finished;
            if (failed_ && batch.size() > 1) {
                for (Request *q : batch) {
                    q->solo = true;
                    q->rc = SG_OK;
                    q->error.clear();
                    q->wire = nullptr;
                    q->n = 0;
                    q->wire_buf = -1;
                    back.push_back(q);
                }
            } else {
                finished = batch;
            }
            lk.lock();
            if (wb >= 0) {
                int users = 0;
                for (Request *q : finished)
                    users += q->wire_buf == wb;
                wire_[wb].users = users; // 0: free again at once
            }
            for (Request *q : finished)
                q->done = true;
            queue_.insert(queue_.begin(), back.begin(), back.end());
            busy_ = false;
            cv_.notify_all();
        }
    }

    void destroy()
    {
        std::lock_guard<std::mutex> lk(m_);
        for (Keyed &k : ctxs_)
            if (k.ctx)
                sg_destroy(k.ctx);
        ctxs_.clear();
        ctx_ = nullptr;
        for (WireBuf &w : wire_) {
            if (w.p)
                sg_host_free(w.p);
            w = WireBuf();
        }
        for (Slot &sl : slots_) {
            if (sl.p)
                sg_host_free(sl.p);
            sl = Slot();
        }
    }

private:
    struct Slot {
        uint8_t *p = nullptr;
        size_t cap = 0;
        bool used = false;
    };
    struct WireBuf {
        sg_wire_feature *p = nullptr;
        size_t cap = 0;
        int users = 0; // callers still expanding out of it; -1: held by the leader
    };

    // One context per (sigma, detector) pair in use, so that callers of Sift() and Sift_nointerval(), or callers with
    // different sigma, who take turns on this lane do not tear a context down and build another on every turn.  At
    // most kMaxKeyed are kept (each holds a batch's pyramids); the one unused for longest goes first.
    struct Keyed {
        sg_ctx *ctx = nullptr;
        double sigma = 0;
        int ni = -1;
        int max_w = 0, max_h = 0, batch = 0;
        size_t max_features = 0;
        unsigned long long last_use = 0;
    };
    static constexpr size_t kMaxKeyed = 4;

    // makes ctx_ a context for images up to w x h with these parameters and room for min_features features per image
    bool make_context(int w, int h, double sigma, int ni, size_t min_features)
    {
        Keyed *k = nullptr;
        for (Keyed &c : ctxs_)
            if (c.sigma == sigma && c.ni == ni)
                k = &c;
        if (!k) {
            if (ctxs_.size() >= kMaxKeyed) {
                size_t oldest = 0;
                for (size_t i = 1; i < ctxs_.size(); i++)
                    if (ctxs_[i].last_use < ctxs_[oldest].last_use)
                        oldest = i;
                if (ctxs_[oldest].ctx)
                    sg_destroy(ctxs_[oldest].ctx);
                ctxs_.erase(ctxs_.begin() + (ptrdiff_t)oldest);
            }
            ctxs_.push_back(Keyed());
            k = &ctxs_.back();
            k->sigma = sigma;
            k->ni = ni;
        }
        k->last_use = ++use_clock_;
        ctx_ = nullptr;
        const bool fits = k->ctx && w <= k->max_w && h <= k->max_h && min_features <= k->max_features && k->batch == B_;
        if (fits) {
            ctx_ = k->ctx;
            max_features_ = k->max_features;
            return true;
        }
        if (min_features > k->max_features)
            k->max_features = min_features;
        if (k->ctx)
            sg_destroy(k->ctx);
        k->ctx = nullptr;
        sg_params p;
        sg_default_params(&p);
        p.sigma = sigma;
        p.nointerval = ni;
        p.precision = precision_now(); // (SiftSetPrecision drops the shared contexts, so a lane never mixes two)
        p.max_width = w > k->max_w ? w : k->max_w;
        p.max_height = h > k->max_h ? h : k->max_h;
        p.device = device_;
        p.max_features = k->max_features > 0x7fffffff ? 0x7fffffff : (int)k->max_features;
        // a batch of large images may not fit the device (an 8K frame takes 10 GB): first the other contexts of this
        // lane go, then fewer images per batch
        for (;;) {
            p.max_batch = B_;
            if (sg_create(&p, &k->ctx) == SG_OK)
                break;
            k->ctx = nullptr;
            bool freed = false;
            for (Keyed &c : ctxs_)
                if (&c != k && c.ctx) {
                    sg_destroy(c.ctx);
                    c.ctx = nullptr;
                    c.max_w = c.max_h = 0;
                    freed = true;
                }
            if (freed)
                continue;
            if (B_ <= 1) {
                k->max_w = k->max_h = 0;
                return false;
            }
            B_ = (B_ + 1) / 2;
        }
        k->max_w = p.max_width;
        k->max_h = p.max_height;
        k->batch = B_;
        ctx_ = k->ctx;
        max_features_ = k->max_features;
        return true;
    }

    // the context in use is dropped (its batch did not fit beside its own transfer buffers)
    void drop_current_context()
    {
        for (Keyed &c : ctxs_)
            if (c.ctx == ctx_ && ctx_) {
                sg_destroy(c.ctx);
                c.ctx = nullptr;
                c.max_w = c.max_h = 0;
            }
        ctx_ = nullptr;
    }

    void fail_all(const std::vector<Request *> &batch, int rc, const char *what)
    {
        failed_ = true;
        for (Request *q : batch) {
            q->rc = rc;
            q->error = what && *what ? what : sg_strerror(rc);
        }
    }

    // runs one batch; returns the wire buffer its results lie in (-1: none)
    int execute(std::vector<Request *> &batch, std::vector<Request *> &back)
    {
        const Request &h = *batch[0];
        if (!make_context(h.w < 2 ? 2 : h.w, h.h < 2 ? 2 : h.h, h.sigma, h.nointerval, 0)) {
            fail_all(batch, SG_E_HIP, "sg_create failed (is a HIP device present? there is no CPU fallback)");
            return -1;
        }
        auto trim = [&] { // the context for this image size came out smaller than the batch collected
            if ((int)batch.size() > B_) {
                back.insert(back.begin(), batch.begin() + B_, batch.end());
                batch.resize((size_t)B_);
            }
        };
        trim();
        int n = (int)batch.size();
        std::vector<const uint8_t *> imgs(batch.size());
        for (int i = 0; i < n; i++)
            imgs[(size_t)i] = batch[(size_t)i]->px;
        std::vector<uint32_t> counts(batch.size());
        int rc = SG_OK;
        for (int attempt = 0; attempt < 8; attempt++) {
            rc = sg_extract_batch_begin(ctx_, imgs.data(), n, h.w, h.h, h.ch, (size_t)h.w * h.ch, 1, counts.data());
            if (rc == SG_E_NOMEM && B_ > 1) {
                // the context fitted the device but left no room for the batch's transfer buffers: half the batch
                drop_current_context();
                B_ = (B_ + 1) / 2;
                if (!make_context(h.w, h.h, h.sigma, h.nointerval, 0))
                    break;
                trim();
                n = (int)batch.size();
                continue;
            }
            if (rc != SG_E_CAPACITY)
                break;
            size_t need = 0; // an image denser than the context was sized for: a larger context, and again
            if (sg_required_features(ctx_, &need) != SG_OK || need <= max_features_ ||
                !make_context(h.w, h.h, h.sigma, h.nointerval, need))
                break;
            trim();
            n = (int)batch.size();
        }
        if (rc != SG_OK) {
            fail_all(batch, rc, ctx_ ? sg_last_error(ctx_) : "");
            return -1;
        }
        size_t total = 0;
        for (uint32_t v : counts)
            total += v;
        for (int i = 0; i < n; i++) {
            Request *q = batch[(size_t)i];
            q->octaves = sg_last_octaves(ctx_);
            if (q->want_counters)
                sg_read_counters(ctx_, i, &q->counters);
        }
        if (total == 0)
            return -1; // rc OK, n 0 everywhere
        int wb;
        {
            std::unique_lock<std::mutex> lk(m_);
            for (;;) {
                wb = wire_[0].users == 0 ? 0 : wire_[1].users == 0 ? 1 : -1;
                if (wb >= 0)
                    break;
                cv_.wait(lk);
            }
            wire_[wb].users = -1;
            held_wire_ = wb;
        }
        WireBuf &w = wire_[wb];
        if (w.cap < total) {
            if (w.p)
                sg_host_free(w.p);
            w.p = nullptr;
            w.cap = 0;
            void *q = nullptr;
            const size_t cap = total + total / 4;
            if (sg_host_alloc(cap * sizeof(sg_wire_feature), &q) != SG_OK) {
                fail_all(batch, SG_E_NOMEM, "no page-locked memory for the feature records");
                return wb; // users counts to 0 in run(): free again
            }
            w.p = static_cast<sg_wire_feature *>(q);
            w.cap = cap;
        }
        rc = sg_extract_batch_fetch_wire(ctx_, w.p, w.cap);
        if (rc != SG_OK) {
            fail_all(batch, rc, sg_last_error(ctx_));
            return wb;
        }
        size_t at = 0;
        for (int i = 0; i < n; i++) {
            Request *q = batch[(size_t)i];
            q->n = counts[(size_t)i];
            if (q->n) {
                q->wire = w.p + at;
                q->wire_buf = wb;
            }
            at += q->n;
        }
        return wb;
    }

    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Request *> queue_;
    bool busy_ = false;
    const int device_;
    int B_; // images per batch: as asked for, halved while a context of that size does not fit the device
    sg_ctx *ctx_ = nullptr;        // the context of the batch being run (one of ctxs_)
    size_t max_features_ = 0;      // its per-image feature capacity as asked for (0: the library's default)
    std::vector<Keyed> ctxs_;      // leader only
    unsigned long long use_clock_ = 0;
    bool failed_ = false;          // the batch being run was failed as a whole (leader only)
    WireBuf wire_[2];
    int held_wire_ = -1; // wire buffer the running batch has taken (leader only)
    std::vector<Slot> slots_;
};

std::atomic<int> g_batching(-1); // -1: not set yet (the environment decides on first use)
std::mutex g_batchers_mutex;
std::map<int, Batcher *> g_batchers; // per (device, lane); live until SiftSetBatching() drops them

int batching()
{
    int b = g_batching.load();
    if (b < 0) {
        const char *e = std::getenv("SIFT_GPU_BATCH");
        b = e ? std::atoi(e) : 0;
        if (b < 0)
            b = 0;
        if (b > 64)
            b = 64;
        int expected = -1;
        g_batching.compare_exchange_strong(expected, b);
        b = g_batching.load();
    }
    return b;
}

// Two shared contexts per device, taken in turn by the calls: while one batch is on the bus (images up, records
// down) or between leaders, the other one's kernels keep the GPU busy.
int lanes()
{
    static const int n = [] {
        const char *e = std::getenv("SIFT_GPU_LANES");
        const int v = e ? std::atoi(e) : 2;
        return v < 1 ? 1 : (v > 4 ? 4 : v);
    }();
    return n;
}
std::atomic<unsigned int> g_next_lane(0);

Batcher *batcher_for(int device, int max_batch)
{
    const int lane = (int)(g_next_lane.fetch_add(1) % (unsigned int)lanes());
    std::lock_guard<std::mutex> lk(g_batchers_mutex);
    Batcher *&b = g_batchers[device * 4 + lane];
    if (!b)
        b = new Batcher(device, max_batch);
    return b;
}

void print_counters(const Mat &src, const sg_counters &c, int octaves, size_t n)
{ // the reference's progress lines (sift.cpp:984-986, 1654-1706)
    std::cout << "rows = " << 2 * src.rows << "  cols = " << 2 * src.cols << "  octaves = " << octaves << std::endl;
    std::cout << "cc00: " << c.cc00 << ", cc0: " << c.cc0 << ", cc1: " << c.cc1 << ", cc2: " << c.cc2
              << ", cc3: " << c.cc3 << std::endl;
    std::cout << "keypoints cout: " << c.cc3 << std::endl;
    std::cout << "features count: " << n << std::endl;
}

void run_sift_shared(const Mat &src, std::vector<Keypoint> &features, double sigma, int nointerval, int max_batch)
{
    Batcher *b = batcher_for(tls.device, max_batch);
    const size_t row = (size_t)src.cols * src.channels();
    uint8_t *slot_mem = nullptr;
    const int slot = b->acquire_slot(row * (size_t)src.rows, &slot_mem);
    if (slot < 0) {
        set_status(SG_E_NOMEM, "no page-locked memory for the input image");
        return;
    }
    for (int y = 0; y < src.rows; y++)
        std::memcpy(slot_mem + row * (size_t)y, src.data + src.step * (size_t)y, row);
    Request r;
    r.px = slot_mem;
    r.w = src.cols;
    r.h = src.rows;
    r.ch = src.channels();
    r.sigma = sigma;
    r.nointerval = nointerval;
    r.want_counters = tls.verbose;
    b->run(r);
    b->release_slot(slot);
    if (r.rc != SG_OK) {
        set_status(r.rc, r.error.c_str());
        return;
    }
    if (r.n) {
        try {
            append_expanded(features, r.wire, r.n);
        } catch (...) { // the caller's vector could not grow: the shared buffer is handed back all the same
            b->release_wire(r.wire_buf);
            throw;
        }
        b->release_wire(r.wire_buf);
    }
    set_status(SG_OK);
    if (tls.verbose)
        print_counters(src, r.counters, r.octaves, r.n);
}

void run_sift(const Mat &src, std::vector<Keypoint> &features, double sigma, int intervals, int nointerval)
{
    if (src.empty() || src.depth() != CV_8U || src.channels() < 3) {
        set_status(SG_E_INVALID, "Sift needs an 8-bit image with at least 3 interleaved channels (b,g,r)");
        return;
    }
    if (intervals != INTERVALS) {
        set_status(SG_E_UNSUPPORTED, "only intervals == 3 is supported (the reference hard-wires INTERVALS)");
        return;
    }
    const int shared = batching();
    if (shared > 1) {
        run_sift_shared(src, features, sigma, nointerval, shared);
        return;
    }
    sg_ctx *ctx = context_for(src.cols, src.rows, sigma, nointerval);
    if (!ctx)
        return;
    // The path runs and reports its count; the records come down as 224-byte wire records into a page-locked buffer
    // of this thread and are expanded behind the features the caller already holds (the reference appends).
    uint32_t count = 0;
    const uint8_t *one[1] = {src.data};
    int rc = SG_OK;
    for (int attempt = 0; attempt < 4; attempt++) {
        rc = sg_extract_batch_begin(ctx, one, 1, src.cols, src.rows, src.channels(), src.step, 0, &count);
        if (rc != SG_E_CAPACITY || !grow_context(src.cols, src.rows, sigma, nointerval))
            break;
        ctx = tls.ctx;
    }
    set_status(rc);
    if (rc != SG_OK)
        return;
    const size_t n = count;
    if (n > tls.wire_cap) {
        if (tls.wire)
            sg_host_free(tls.wire);
        tls.wire = nullptr;
        tls.wire_cap = 0;
        void *q = nullptr;
        const size_t cap = n + n / 4 + 1024;
        if (sg_host_alloc(cap * sizeof(sg_wire_feature), &q) != SG_OK) {
            set_status(SG_E_NOMEM, "no page-locked memory for the feature records");
            return;
        }
        tls.wire = static_cast<sg_wire_feature *>(q);
        tls.wire_cap = cap;
    }
    if (n) {
        // The records come down in pieces; a piece is expanded into the caller's vector while the next is on the bus.
        const size_t piece = 4096; // records (0.9 MB on the wire, 4.6 MB expanded)
        int n_pieces = 0;
        rc = sg_extract_batch_fetch_wire_begin(ctx, tls.wire, tls.wire_cap, piece, &n_pieces);
        set_status(rc);
        if (rc != SG_OK)
            return;
        const size_t n_before = features.size();
        features.reserve(n_before + n);
        for (int k = 0; k < n_pieces; k++) {
            rc = sg_extract_batch_fetch_wire_wait(ctx, k);
            if (rc != SG_OK) { // a failed call leaves the caller's vector as it was
                features.resize(n_before);
                set_status(rc);
                return;
            }
            const size_t at = (size_t)k * piece;
            append_expanded(features, tls.wire + at, n - at < piece ? n - at : piece);
        }
    }
    if (tls.verbose) {
        sg_counters c;
        sg_read_counters(ctx, 0, &c);
        print_counters(src, c, sg_last_octaves(ctx), n);
    }
}

void run_detect(const std::vector<Mat> &dog_pyr, std::vector<Keypoint> &extrema, int octaves, int intervals, int ni)
{
    if (intervals != INTERVALS || octaves < 0 || dog_pyr.size() < (size_t)octaves * (INTERVALS + 2)) {
        set_status(SG_E_INVALID, "DetectionLocalExtrema: bad pyramid");
        return;
    }
    if (octaves == 0) {
        set_status(SG_OK);
        return;
    }
    const int rows = dog_pyr[0].rows, cols = dog_pyr[0].cols;
    sg_ctx *ctx = context_for_plane(cols, rows);
    if (!ctx)
        return;
    std::vector<PlaneView> views;
    views.reserve((size_t)octaves * (INTERVALS + 2));
    std::vector<const double *> ptrs;
    for (int i = 0; i < octaves * (INTERVALS + 2); i++) {
        if (!is_f64_plane(dog_pyr[i])) {
            set_status(SG_E_INVALID, "DetectionLocalExtrema: levels must be CV_64FC1");
            return;
        }
        views.emplace_back(dog_pyr[i]);
        ptrs.push_back(views.back().p);
    }
    size_t cap = (size_t)rows * cols / 40 + 4096, n = 0;
    std::vector<Keypoint> out(cap);
    sg_counters c;
    int rc = SG_OK;
    for (int attempt = 0; attempt < 4; attempt++) {
        rc = sg_detect_local_extrema(ctx, ptrs.data(), cols, rows, octaves, intervals, ni,
                                     reinterpret_cast<sg_keypoint *>(out.data()), cap, &n, &c);
        if (rc != SG_E_CAPACITY)
            break;
        if (n > cap) { // the output array was the small one
            cap = n;
            out.resize(cap);
        } else if (grow_context((cols + 1) / 2, (rows + 1) / 2, tls.sigma, tls.nointerval)) {
            ctx = tls.ctx;
        } else {
            break;
        }
    }
    set_status(rc);
    if (rc != SG_OK)
        return;
    if (tls.verbose)
        std::cout << "cc00: " << c.cc00 << ", cc0: " << c.cc0 << ", cc1: " << c.cc1 << ", cc2: " << c.cc2
                  << ", cc3: " << c.cc3 << std::endl;
    extrema.insert(extrema.end(), out.begin(), out.begin() + (long)n);
}

bool gauss_ptrs(const std::vector<Mat> &gauss_pyr, int octaves, std::vector<PlaneView> &views,
                std::vector<const double *> &ptrs)
{
    views.reserve((size_t)octaves * (INTERVALS + 3));
    for (int i = 0; i < octaves * (INTERVALS + 3); i++) {
        if (!is_f64_plane(gauss_pyr[i]))
            return false;
        views.emplace_back(gauss_pyr[i]);
        ptrs.push_back(views.back().p);
    }
    return true;
}

} // namespace

// ------------------------------------------------------------------------------------------ additions

int SiftLastStatus() { return tls.status; }
const char *SiftLastError() { return tls.error.c_str(); }
void SiftSetDevice(int device)
{
    if (device != tls.device)
        SiftReleaseContext();
    tls.device = device;
}
void SiftSetVerbose(bool on) { tls.verbose = on; }
void SiftSetBatching(int max_batch)
{
    if (max_batch < 0)
        max_batch = 0;
    if (max_batch > 64)
        max_batch = 64;
    std::lock_guard<std::mutex> lk(g_batchers_mutex);
    for (auto &kv : g_batchers) {
        kv.second->destroy();
        delete kv.second;
    }
    g_batchers.clear();
    g_batching.store(max_batch);
}
void SiftSetPrecision(int precision)
{
    const int p = precision == SIFT_FUSED ? SG_F64_FMA : (precision == SIFT_SINGLE ? SG_F32 : SG_F64);
    std::lock_guard<std::mutex> lk(g_batchers_mutex);
    for (auto &kv : g_batchers) {
        kv.second->destroy();
        delete kv.second;
    }
    g_batchers.clear();
    g_precision.store(p); // per-thread contexts notice at their next call
}
void SiftReleaseContext()
{
    if (tls.ctx)
        sg_destroy(tls.ctx);
    tls.ctx = nullptr;
    tls.max_w = tls.max_h = 0;
    tls.max_features = 0;
}

// ------------------------------------------------------------------------------------------ stages

// sift/sift.cpp:20-51; dst is created only when empty, like the reference
void ConvertToGray(const Mat &src, Mat &dst)
{
    if (src.empty() || src.depth() != CV_8U || src.channels() < 3) {
        set_status(SG_E_INVALID, "ConvertToGray needs an 8-bit image with at least 3 channels");
        return;
    }
    if (dst.empty())
        dst.create(src.rows, src.cols, CV_64FC1);
    if (dst.rows != src.rows || dst.cols != src.cols || dst.type() != CV_64FC1 || !dst.isContinuous()) {
        set_status(SG_E_INVALID, "ConvertToGray: dst has the wrong shape");
        return;
    }
    sg_ctx *ctx = context_for_plane(src.cols, src.rows);
    if (!ctx)
        return;
    set_status(sg_convert_to_gray(ctx, src.data, src.cols, src.rows, src.channels(), src.step,
                                  reinterpret_cast<double *>(dst.data)));
}

// sift/sift.cpp:59-103
void DownSample(const Mat &src, Mat &dst)
{
    if (src.empty() || src.channels() != 1 || src.type() != CV_64FC1)
        return;
    if (src.cols <= 1 || src.rows <= 1) {
        src.copyTo(dst);
        return;
    }
    dst.create(src.rows / 2, src.cols / 2, CV_64FC1);
    sg_ctx *ctx = context_for_plane(src.cols, src.rows);
    if (!ctx)
        return;
    PlaneView v(src);
    set_status(sg_downsample(ctx, v.p, src.cols, src.rows, reinterpret_cast<double *>(dst.data)));
}

// sift/sift.cpp:111-175
void UpSample(const Mat &src, Mat &dst)
{
    if (src.empty() || src.channels() != 1 || src.type() != CV_64FC1)
        return;
    dst.create(src.rows * 2, src.cols * 2, CV_64FC1);
    if (src.rows < 2 || src.cols < 2) { // the reference leaves such a dst unwritten; keep it defined
        std::memset(dst.data, 0, (size_t)(dst.dataend - dst.data));
        return;
    }
    sg_ctx *ctx = context_for_plane(2 * src.cols, 2 * src.rows);
    if (!ctx)
        return;
    PlaneView v(src);
    set_status(sg_upsample(ctx, v.p, src.cols, src.rows, reinterpret_cast<double *>(dst.data)));
}

// sift/sift.cpp:323-326
void GaussianSmooth(const Mat &src, Mat &dst, double sigma)
{
    if (!is_f64_plane(src) || !(sigma > 0)) {
        set_status(SG_E_INVALID, "GaussianSmooth needs a CV_64FC1 image and sigma > 0");
        return;
    }
    sg_ctx *ctx = context_for_plane(src.cols, src.rows);
    if (!ctx)
        return;
    PlaneView v(src); // taken before dst.create so that in-place calls (src and dst the same Mat) work
    Mat out(src.rows, src.cols, CV_64FC1);
    int rc = sg_gaussian_smooth(ctx, v.p, reinterpret_cast<double *>(out.data), src.cols, src.rows, sigma);
    set_status(rc);
    if (rc == SG_OK)
        dst = out;
}

// sift/sift.cpp:335-346
void CreateInitSmoothGray(const Mat &src, Mat &dst, double sigma)
{
    if (src.empty() || src.depth() != CV_8U || src.channels() < 3 || src.rows < 2 || src.cols < 2) {
        set_status(SG_E_INVALID, "CreateInitSmoothGray needs an 8-bit image with at least 3 channels");
        return;
    }
    sg_ctx *ctx = context_for(src.cols, src.rows, tls.ctx ? tls.sigma : SIGMA, tls.ctx ? tls.nointerval : 0);
    if (!ctx)
        return;
    Mat out(2 * src.rows, 2 * src.cols, CV_64FC1);
    int rc = sg_create_init_smooth_gray(ctx, src.data, src.cols, src.rows, src.channels(), src.step, sigma,
                                        reinterpret_cast<double *>(out.data));
    set_status(rc);
    if (rc == SG_OK)
        dst = out;
}

// sift/sift.cpp:357-407; appends octaves*(intervals+3) levels
void GaussianPyramid(const Mat &src, std::vector<Mat> &gauss_pyr, int octaves, int intervals, double sigma)
{
    if (!is_f64_plane(src) || octaves < 0 || intervals != INTERVALS) {
        set_status(SG_E_INVALID, "GaussianPyramid needs a CV_64FC1 base and intervals == 3");
        return;
    }
    sg_ctx *ctx = context_for_plane(src.cols, src.rows);
    if (!ctx)
        return;
    std::vector<int> r, c;
    octave_dims(src.rows, src.cols, octaves, r, c);
    std::vector<Mat> levels;
    std::vector<double *> ptrs;
    for (int o = 0; o < octaves; o++)
        for (int i = 0; i < intervals + 3; i++) {
            levels.emplace_back(r[o], c[o], CV_64FC1);
            ptrs.push_back(reinterpret_cast<double *>(levels.back().data));
        }
    PlaneView v(src);
    int rc = sg_gaussian_pyramid(ctx, v.p, src.cols, src.rows, octaves, intervals, sigma, ptrs.data());
    set_status(rc);
    if (rc == SG_OK)
        gauss_pyr.insert(gauss_pyr.end(), levels.begin(), levels.end());
}

// sift/sift.cpp:416-437; like the reference, returns silently on a shape mismatch or a non-empty c
void Sub(const Mat &a, const Mat &b, Mat &c)
{
    if (a.rows != b.rows || a.cols != b.cols || a.type() != b.type())
        return;
    if (!c.empty())
        return;
    if (!is_f64_plane(a)) {
        set_status(SG_E_INVALID, "Sub needs CV_64FC1 images");
        return;
    }
    sg_ctx *ctx = context_for_plane(a.cols, a.rows);
    if (!ctx)
        return;
    c.create(a.rows, a.cols, CV_64FC1);
    PlaneView va(a), vb(b);
    set_status(sg_sub(ctx, va.p, vb.p, reinterpret_cast<double *>(c.data), a.cols, a.rows));
}

// sift/sift.cpp:447-461; appends octaves*(intervals+2) levels
void DogPyramid(const std::vector<Mat> &gauss_pyr, std::vector<Mat> &dog_pyr, int octaves, int intervals)
{
    if (intervals != INTERVALS || octaves < 0 || gauss_pyr.size() < (size_t)octaves * (INTERVALS + 3)) {
        set_status(SG_E_INVALID, "DogPyramid: bad pyramid");
        return;
    }
    for (int o = 0; o < octaves; o++)
        for (int i = 1; i < intervals + 3; i++) {
            Mat m;
            Sub(gauss_pyr[o * (intervals + 3) + i], gauss_pyr[o * (intervals + 3) + i - 1], m);
            if (tls.status != SG_OK)
                return;
            dog_pyr.push_back(m);
        }
}

// sift/sift.cpp:920-987 and 989-1056
void DetectionLocalExtrema(const std::vector<Mat> &dog_pyr, std::vector<Keypoint> &extrema, int octaves, int intervals)
{
    run_detect(dog_pyr, extrema, octaves, intervals, 0);
}

void DetectionLocalExtrema_nointerval(const std::vector<Mat> &dog_pyr, std::vector<Keypoint> &extrema, int octaves,
                                      int intervals)
{
    run_detect(dog_pyr, extrema, octaves, intervals, 1);
}

// sift/sift.cpp:1065-1077
void CalculateScale(std::vector<Keypoint> &features, double sigma, int intervals)
{
    set_status(sg_calculate_scale(reinterpret_cast<sg_keypoint *>(features.data()), features.size(), sigma, intervals));
}

// sift/sift.cpp:1085-1096
void HalfFeatures(std::vector<Keypoint> &features)
{
    set_status(sg_half_features(reinterpret_cast<sg_keypoint *>(features.data()), features.size()));
}

// sift/sift.cpp:1317-1339; appends one record per orientation peak
void OrientationAssignment(std::vector<Keypoint> &extrema, std::vector<Keypoint> &features,
                           const std::vector<Mat> &gauss_pyr)
{
    const int octaves = (int)(gauss_pyr.size() / (INTERVALS + 3));
    if (octaves == 0 || extrema.empty()) {
        set_status(SG_OK);
        return;
    }
    sg_ctx *ctx = context_for_plane(gauss_pyr[0].cols, gauss_pyr[0].rows);
    if (!ctx)
        return;
    std::vector<PlaneView> views;
    std::vector<const double *> ptrs;
    if (!gauss_ptrs(gauss_pyr, octaves, views, ptrs)) {
        set_status(SG_E_INVALID, "OrientationAssignment: levels must be CV_64FC1");
        return;
    }
    size_t cap = 2 * extrema.size() + 64, n = 0;
    std::vector<Keypoint> out(cap);
    int rc = SG_OK;
    for (int attempt = 0; attempt < 4; attempt++) {
        rc = sg_orientation_assignment(ctx, reinterpret_cast<const sg_keypoint *>(extrema.data()), extrema.size(),
                                       ptrs.data(), gauss_pyr[0].cols, gauss_pyr[0].rows, octaves,
                                       reinterpret_cast<sg_keypoint *>(out.data()), cap, &n);
        if (rc != SG_E_CAPACITY)
            break;
        if (n > cap) {
            cap = n;
            out.resize(cap);
        } else if (grow_context((gauss_pyr[0].cols + 1) / 2, (gauss_pyr[0].rows + 1) / 2, tls.sigma, tls.nointerval)) {
            ctx = tls.ctx;
        } else if (extrema.size() > tls.max_features &&
                   context_for((gauss_pyr[0].cols + 1) / 2, (gauss_pyr[0].rows + 1) / 2, tls.sigma, tls.nointerval,
                               2 * extrema.size() + 64)) {
            ctx = tls.ctx; // more records than the context's feature capacity: sized from the input itself
        } else {
            break;
        }
    }
    set_status(rc);
    if (rc == SG_OK)
        features.insert(features.end(), out.begin(), out.begin() + (long)n);
}

// sift/sift.cpp:1607-1631
void DescriptorRepresentation(std::vector<Keypoint> &features, const std::vector<Mat> &gauss_pyr, int bins, int width)
{
    if (bins != DESCR_HIST_BINS || width != DESCR_WINDOW_WIDTH) {
        set_status(SG_E_UNSUPPORTED, "DescriptorRepresentation supports the 4x4x8 layout only");
        return;
    }
    const int octaves = (int)(gauss_pyr.size() / (INTERVALS + 3));
    if (octaves == 0 || features.empty()) {
        set_status(SG_OK);
        return;
    }
    sg_ctx *ctx = context_for_plane(gauss_pyr[0].cols, gauss_pyr[0].rows);
    if (!ctx)
        return;
    std::vector<PlaneView> views;
    std::vector<const double *> ptrs;
    if (!gauss_ptrs(gauss_pyr, octaves, views, ptrs)) {
        set_status(SG_E_INVALID, "DescriptorRepresentation: levels must be CV_64FC1");
        return;
    }
    set_status(sg_descriptor_representation(ctx, reinterpret_cast<sg_keypoint *>(features.data()), features.size(),
                                            ptrs.data(), gauss_pyr[0].cols, gauss_pyr[0].rows, octaves));
}

// sift/sift.cpp:1640-1643
bool FeatureCmp(Keypoint &f1, Keypoint &f2) { return f1.scale < f2.scale; }

// ------------------------------------------------------------------------------------------ drivers

// sift/sift.cpp:1646-1707
void Sift(const Mat &src, std::vector<Keypoint> &features, double sigma, int intervals)
{
    run_sift(src, features, sigma, intervals, 0);
}

// sift/sift.cpp:1709-1770
void Sift_nointerval(const Mat &src, std::vector<Keypoint> &features, double sigma, int intervals)
{
    run_sift(src, features, sigma, intervals, 1);
}

// ------------------------------------------------------------------------------------------ text files

// sift/sift.cpp:1886-1908: "n dims", then per feature "dy dx scale ori" and dims descriptor values
void read_features(std::vector<Keypoint> &features, const char *file)
{
    std::ifstream in(file);
    int n = 0, dims = 0;
    in >> n >> dims;
    if (!in || dims < 0 || dims > FEATURE_ELEMENT_LENGTH) {
        set_status(SG_E_INVALID, "read_features: bad header");
        return;
    }
    for (int i = 0; i < n; i++) {
        Keypoint key;
        std::memset(&key, 0, sizeof key);
        in >> key.dy >> key.dx >> key.scale >> key.ori;
        for (int j = 0; j < dims; j++)
            in >> key.descriptor[j];
        if (!in) {
            set_status(SG_E_INVALID, "read_features: truncated file");
            return;
        }
        key.descr_length = dims;
        features.push_back(key);
    }
    set_status(SG_OK);
}

// sift/源.cpp:88-156 (the demo's matching loop); appends to `matches`
void MatchFeatures(const std::vector<Keypoint> &features1, const std::vector<Keypoint> &features2,
                   std::vector<MatchPoint> &matches, double ratio)
{
    if (features1.empty()) {
        set_status(SG_OK);
        return;
    }
    sg_ctx *ctx = tls.ctx ? tls.ctx : context_for(2, 2, SIGMA, 0);
    if (!ctx)
        return;
    static_assert(sizeof(MatchPoint) == sizeof(sg_match_point), "MatchPoint layout");
    std::vector<MatchPoint> found(features1.size());
    size_t n = 0;
    int rc = sg_match(ctx, reinterpret_cast<const sg_keypoint *>(features1.data()), features1.size(),
                      reinterpret_cast<const sg_keypoint *>(features2.data()), features2.size(), ratio,
                      reinterpret_cast<sg_match_point *>(found.data()), found.size(), &n);
    set_status(rc, rc == SG_OK ? nullptr : sg_last_error(ctx));
    if (rc == SG_OK)
        matches.insert(matches.end(), found.begin(), found.begin() + n);
}

// sift/sift.cpp:1910-1931: default stream formatting, a line break before every 20th descriptor value
void write_features(const std::vector<Keypoint> &features, const char *file)
{
    std::ofstream out(file);
    out << features.size() << " " << FEATURE_ELEMENT_LENGTH << std::endl;
    for (size_t i = 0; i < features.size(); i++) {
        const Keypoint &f = features[i];
        out << f.dy << " " << f.dx << " " << f.scale << " " << f.ori << std::endl;
        for (int j = 0; j < FEATURE_ELEMENT_LENGTH; j++) {
            if (j % 20 == 0)
                out << std::endl;
            out << f.descriptor[j] << " ";
        }
        out << std::endl << std::endl;
    }
    set_status(out ? SG_OK : SG_E_INVALID, out ? nullptr : "write_features: cannot write file");
}
