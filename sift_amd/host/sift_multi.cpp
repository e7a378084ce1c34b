// sift_multi.cpp -- a batch of frames over the GPUs of one node, in host C++ (SiftBatchMultiGpu of sift_gpu.h).
//
// The reference processes one image per Sift() call (sift/sift.cpp:1646-1707, called from sift/源.cpp:21-22) and a
// frame's features depend on nothing but its pixels, so a batch shards by frame: GPU g of G takes the contiguous
// block of frames [g * per, (g + 1) * per), per = ceil(F / G), one host thread per GPU, no communication while the
// shards are extracted.  The one exchange step follows: the ranks all-gather their packed feature sets
// (sg_allgather_features: RCCL all-gather-v over xGMI), after which every GPU holds the features of every frame in
// frame order; the host reads them from GPU 0.
//
// Per rank the frames go through the device-pointer form of the path in batches: while batch k is being extracted,
// the rows of batch k + 1 are copied into page-locked memory and uploaded on a second stream (a small context of
// its own), into the other of two device image buffers.
#include "sift_gpu.h"

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "siftgpu.h"

static_assert(sizeof(SiftPackedFeature) == sizeof(sg_packed_feature) && sizeof(SiftPackedFeature) == 144,
              "SiftPackedFeature is the C ABI's packed feature");

namespace {

// all ranks meet; the last one to arrive releases the others
class Barrier {
public:
    explicit Barrier(int n) : n_(n) {}
    void wait()
    {
        std::unique_lock<std::mutex> lk(m_);
        const unsigned long long gen = gen_;
        if (++count_ == n_) {
            count_ = 0;
            gen_++;
            cv_.notify_all();
            return;
        }
        cv_.wait(lk, [&] { return gen_ != gen; });
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    const int n_;
    int count_ = 0;
    unsigned long long gen_ = 0;
};

struct Shared {
    const std::vector<Mat> *frames = nullptr;
    SiftMultiGpuOptions opt;
    int ranks = 1, n_dev = 1, per = 0, w = 0, h = 0, ch = 0;
    unsigned char comm_id[SG_COMM_ID_BYTES];
    bool pool_comm_ok = false; // every rank still holds a communicator of this many ranks: kept
    Barrier *barrier = nullptr;
    // what the ranks tell each other through host memory
    std::vector<int> status;            // per rank, of the phase just finished
    std::vector<std::string> error;
    std::vector<size_t> need;           // per rank: per-frame capacity an overflowing shard asked for (0: none)
    std::vector<size_t> total;          // per rank: features of its shard
    std::vector<std::vector<uint32_t>> seen_counts; // per rank: all frames' counts as gathered on that rank
    std::vector<std::vector<SiftPackedFeature>> seen_last; // per rank: last feature of every frame as gathered there
    std::vector<std::vector<SiftPackedFeature>> rehearsal_feats; // rehearsal: every rank's compacted shard
    SiftBatchFeatures *out = nullptr;
};

// What a rank holds on its GPU.  Kept between calls in a pool (creating a context of 32 frames, its streams and events,
// and an RCCL communicator costs far more than extracting a small batch), keyed by rank; a call whose shape differs
// (frame size, batch, capacity, parameters, number of ranks) rebuilds what depends on it.
struct Rank {
    sg_ctx *ctx = nullptr, *up = nullptr;
    void *comm = nullptr;
    void *d_img[2] = {nullptr, nullptr}, *d_out = nullptr, *d_cnt = nullptr, *d_all = nullptr, *d_all_cnt = nullptr;
    void *pinned[2] = {nullptr, nullptr};
    // shape of what is held
    int device = -1, comm_ranks = 0, w = 0, h = 0, ch = 0, bsz = 0, per = 0, nointerval = -1, precision = -1, all_cnt_ranks = 0;
    double sigma = 0;
    size_t cap = 0, all_cap = 0, pinned_bytes = 0;
    void free_shard()
    {
        if (ctx) {
            void *bufs[] = {d_img[0], d_img[1], d_out, d_cnt, d_all, d_all_cnt};
            for (void *b : bufs)
                if (b)
                    sg_free(ctx, b);
            sg_destroy(ctx);
        }
        ctx = nullptr;
        d_img[0] = d_img[1] = d_out = d_cnt = d_all = d_all_cnt = nullptr;
        all_cap = 0;
        all_cnt_ranks = 0;
        cap = 0;
    }
    ~Rank()
    {
        if (comm)
            sg_comm_destroy(comm);
        for (void *p : pinned)
            if (p)
                sg_host_free(p);
        free_shard();
        if (up)
            sg_destroy(up);
    }
};

std::mutex g_call_mutex;          // one SiftBatchMultiGpu call at a time
std::vector<Rank *> g_pool;       // by rank; touched under g_call_mutex (each rank thread takes its own entry)

void SiftMultiGpuRelease_locked()
{
    for (Rank *r : g_pool)
        delete r;
    g_pool.clear();
}

// true when every rank of a call of `ranks` ranks still holds its communicator of that size
bool pool_has_comm(int ranks)
{
    if ((int)g_pool.size() < ranks)
        return false;
    for (int g = 0; g < ranks; g++)
        if (!g_pool[(size_t)g] || !g_pool[(size_t)g]->comm || g_pool[(size_t)g]->comm_ranks != ranks)
            return false;
    return true;
}

// The ranks meet and learn whether all of them are well: true on every rank or on none.  (Two meetings: nobody writes
// its next status before everybody has read this one.)
bool all_ok(Shared &s)
{
    s.barrier->wait();
    bool ok = true;
    for (int v : s.status)
        ok = ok && v == SG_OK;
    s.barrier->wait();
    return ok;
}

void rank_main(Shared &s, int rank)
{
    auto fail = [&](int rc, const char *what, sg_ctx *c) {
        s.status[(size_t)rank] = rc;
        std::string e = what;
        if (c && *sg_last_error(c))
            e += std::string(": ") + sg_last_error(c);
        s.error[(size_t)rank] = e;
    };
    const int F = (int)s.frames->size();
    const int first = std::min(F, rank * s.per), stop = std::min(F, first + s.per), mine = stop - first;
    const int device = s.opt.rehearsal ? rank % s.n_dev : rank;
    const int bsz = std::max(1, std::min(s.opt.batch, s.per));
    const size_t frame_bytes = (size_t)s.w * s.h * s.ch;
    const bool rccl = s.ranks > 1 && !s.opt.rehearsal;
    size_t cap = s.opt.max_features_per_frame ? (size_t)s.opt.max_features_per_frame : (size_t)s.w * s.h / 40 + 4096;

    // ---- set-up: what the pool holds for this rank is kept where its shape fits ----
    Rank *&slot = g_pool[(size_t)rank];
    if (slot && slot->device != device) {
        delete slot;
        slot = nullptr;
    }
    if (!slot)
        slot = new Rank;
    Rank &r = *slot;
    r.device = device;
    sg_params p;
    int rc = SG_OK;
    if (!r.up) {
        sg_default_params(&p);
        p.device = device;
        p.max_width = 64;
        p.max_height = 64;
        rc = sg_create(&p, &r.up);
        if (rc != SG_OK)
            fail(rc, "sg_create (upload context)", nullptr);
    }
    const bool keep_comm = s.pool_comm_ok; // decided once for all ranks, before the threads started
    if (r.comm && !(rccl && keep_comm)) {
        sg_comm_destroy(r.comm);
        r.comm = nullptr;
        r.comm_ranks = 0;
    }
    if (!all_ok(s))
        return;
    if (rccl && !r.comm) { // all ranks enter ncclCommInitRank together
        rc = sg_comm_create(r.up, s.comm_id, rank, s.ranks, &r.comm);
        if (rc != SG_OK)
            fail(rc, "sg_comm_create", r.up);
        else
            r.comm_ranks = s.ranks;
    }
    if (!all_ok(s))
        return;

    // ---- extraction of the shard; every rank repeats it with a larger per-frame capacity if any rank overflowed ----
    for (int attempt = 0; attempt < 4; attempt++) {
        s.need[(size_t)rank] = 0;
        const int ni = s.opt.nointerval ? 1 : 0;
        const int prec = s.opt.precision == SIFT_FUSED ? SG_F64_FMA : (s.opt.precision == SIFT_SINGLE ? SG_F32 : SG_F64);
        const bool fits = r.ctx && r.w == s.w && r.h == s.h && r.ch == s.ch && r.bsz == bsz && r.per == s.per &&
                          r.cap == cap && r.sigma == s.opt.sigma && r.nointerval == ni && r.precision == prec;
        if (!fits) { // another shape (or a larger capacity after an overflow): everything sized from it goes
            r.free_shard();
            sg_default_params(&p);
            p.device = device;
            p.sigma = s.opt.sigma;
            p.nointerval = ni;
            p.precision = prec;
            p.max_width = s.w;
            p.max_height = s.h;
            p.max_batch = bsz;
            p.max_features = (int)std::min<size_t>(cap, 0x7fffffff);
            rc = sg_create(&p, &r.ctx);
            if (rc != SG_OK) {
                r.ctx = nullptr;
                fail(rc, "sg_create", nullptr);
            } else {
                rc = sg_malloc(r.ctx, (size_t)s.per * cap * sizeof(sg_packed_feature), &r.d_out);
                if (rc == SG_OK)
                    rc = sg_malloc(r.ctx, sizeof(uint32_t) * (size_t)s.per, &r.d_cnt);
                for (int b = 0; b < 2 && rc == SG_OK; b++)
                    rc = sg_malloc(r.ctx, frame_bytes * (size_t)bsz, &r.d_img[b]);
                if (rc == SG_OK && r.pinned_bytes < frame_bytes * (size_t)bsz) {
                    for (int b = 0; b < 2; b++) {
                        if (r.pinned[b])
                            sg_host_free(r.pinned[b]);
                        r.pinned[b] = nullptr;
                    }
                    r.pinned_bytes = 0;
                    for (int b = 0; b < 2 && rc == SG_OK; b++)
                        rc = sg_host_alloc(frame_bytes * (size_t)bsz, &r.pinned[b]);
                    if (rc == SG_OK)
                        r.pinned_bytes = frame_bytes * (size_t)bsz;
                }
                if (rc != SG_OK) {
                    fail(rc, "device or page-locked memory for the shard", r.ctx);
                    r.free_shard();
                } else {
                    r.w = s.w;
                    r.h = s.h;
                    r.ch = s.ch;
                    r.bsz = bsz;
                    r.per = s.per;
                    r.cap = cap;
                    r.sigma = s.opt.sigma;
                    r.precision = prec;
                    r.nointerval = ni;
                }
            }
        }
        if (s.status[(size_t)rank] == SG_OK) {
            // frames this rank does not have (the last shard may be short) count zero features
            std::vector<uint32_t> zeros((size_t)s.per, 0u);
            rc = sg_memcpy_h2d(r.ctx, r.d_cnt, zeros.data(), sizeof(uint32_t) * zeros.size());
            for (int k0 = 0; k0 < mine && rc == SG_OK; k0 += bsz) {
                const int n = std::min(bsz, mine - k0), b = (k0 / bsz) & 1;
                unsigned char *stage = static_cast<unsigned char *>(r.pinned[b]);
                for (int i = 0; i < n; i++) {
                    const Mat &m = (*s.frames)[(size_t)(first + k0 + i)];
                    for (int y = 0; y < s.h; y++)
                        std::memcpy(stage + frame_bytes * (size_t)i + (size_t)s.w * s.ch * (size_t)y,
                                    m.data + m.step * (size_t)y, (size_t)s.w * s.ch);
                }
                // the upload runs on the small context's stream, beside the extraction of the batch before
                rc = sg_memcpy_h2d(r.up, r.d_img[b], stage, frame_bytes * (size_t)n);
                if (rc != SG_OK)
                    break;
                rc = sg_sync(r.ctx); // the batch before is complete (and the buffer of the one before that free)
                if (rc != SG_OK)
                    break;
                rc = sg_extract_batch_device(r.ctx, static_cast<const uint8_t *>(r.d_img[b]), n, s.w, s.h, s.ch,
                                             (size_t)s.w * s.ch, frame_bytes,
                                             static_cast<sg_packed_feature *>(r.d_out) + (size_t)k0 * cap, cap,
                                             static_cast<uint32_t *>(r.d_cnt) + k0);
            }
            if (rc == SG_OK)
                rc = sg_sync(r.ctx);
            if (rc == SG_E_CAPACITY) {
                size_t need = 0;
                if (sg_required_features(r.ctx, &need) == SG_OK && need > cap)
                    s.need[(size_t)rank] = need;
                else
                    fail(rc, "a frame has more features than any capacity tried", r.ctx);
            } else if (rc != SG_OK) {
                fail(rc, "extraction", r.ctx);
            }
        }
        if (!all_ok(s))
            return;
        size_t agreed = 0;
        for (size_t v : s.need)
            agreed = std::max(agreed, v);
        s.barrier->wait(); // everybody has read `need` before anybody clears it again
        if (agreed == 0)
            break;
        if (attempt == 3) {
            fail(SG_E_CAPACITY, "a frame has more features than any capacity tried", nullptr);
            break;
        }
        cap = agreed + agreed / 8;
    }
    if (!all_ok(s))
        return;

    // ---- the exchange: every rank tells its total, all size the gathered buffer from the sum ----
    std::vector<uint32_t> my_counts((size_t)s.per);
    rc = sg_memcpy_d2h(r.ctx, my_counts.data(), r.d_cnt, sizeof(uint32_t) * my_counts.size());
    if (rc != SG_OK)
        fail(rc, "reading the counts", r.ctx);
    size_t my_total = 0;
    for (uint32_t v : my_counts)
        my_total += std::min<size_t>(v, cap);
    s.total[(size_t)rank] = my_total;
    if (!all_ok(s))
        return;
    size_t total = 0;
    for (size_t v : s.total)
        total += v;
    const size_t gather_total = s.opt.rehearsal ? my_total : total; // without a communicator a rank gathers itself
    const int gather_ranks = rccl ? s.ranks : 1;
    if (r.all_cap < std::max<size_t>(gather_total, 1)) {
        if (r.d_all)
            sg_free(r.ctx, r.d_all);
        r.d_all = nullptr;
        r.all_cap = 0;
        const size_t want = std::max<size_t>(gather_total + gather_total / 8, 1);
        rc = sg_malloc(r.ctx, want * sizeof(sg_packed_feature), &r.d_all);
        if (rc == SG_OK)
            r.all_cap = want;
    }
    if (rc == SG_OK && r.all_cnt_ranks < gather_ranks) {
        if (r.d_all_cnt)
            sg_free(r.ctx, r.d_all_cnt);
        r.d_all_cnt = nullptr;
        r.all_cnt_ranks = 0;
        rc = sg_malloc(r.ctx, sizeof(uint32_t) * (size_t)s.per * (size_t)gather_ranks, &r.d_all_cnt);
        if (rc == SG_OK)
            r.all_cnt_ranks = gather_ranks;
    }
    std::vector<size_t> rank_first((size_t)gather_ranks + 1, 0);
    if (rc != SG_OK)
        fail(rc, "device memory for the gathered features", r.ctx);
    if (!all_ok(s)) // no rank enters the collective unless every rank can
        return;
    rc = sg_allgather_features(r.ctx, rccl ? r.comm : nullptr, static_cast<const sg_packed_feature *>(r.d_out),
                               static_cast<const uint32_t *>(r.d_cnt), s.per, cap, static_cast<sg_packed_feature *>(r.d_all),
                               r.all_cap, static_cast<uint32_t *>(r.d_all_cnt), rank_first.data());
    if (rc == SG_OK)
        rc = sg_sync(r.ctx);
    if (rc != SG_OK)
        fail(rc, "sg_allgather_features", r.ctx);
    else if (rank_first[(size_t)gather_ranks] != gather_total)
        fail(SG_E_HIP, "the gathered total differs from the sum of the ranks' totals", nullptr);
    if (!all_ok(s))
        return;

    // ---- to the host ----
    if (s.opt.rehearsal) { // ranks share devices here and RCCL cannot run: the sets meet in host memory
        std::vector<SiftPackedFeature> &mine_feats = s.rehearsal_feats[(size_t)rank];
        mine_feats.resize(my_total);
        if (my_total)
            rc = sg_memcpy_d2h(r.ctx, mine_feats.data(), r.d_all, my_total * sizeof(SiftPackedFeature));
        s.seen_counts[(size_t)rank] = my_counts;
        if (rc != SG_OK)
            fail(rc, "download", r.ctx);
        s.barrier->wait();
        return;
    }
    // every rank reports what it sees of the gathered set: all counts and the last feature of every frame
    std::vector<uint32_t> &cnts = s.seen_counts[(size_t)rank];
    cnts.resize((size_t)s.per * (size_t)gather_ranks);
    rc = sg_memcpy_d2h(r.ctx, cnts.data(), r.d_all_cnt, sizeof(uint32_t) * cnts.size());
    std::vector<SiftPackedFeature> &last = s.seen_last[(size_t)rank];
    last.assign(cnts.size(), SiftPackedFeature());
    if (rank == 0) { // rank 0 brings the whole set down, straight into the caller's array; the others sample it
        std::vector<SiftPackedFeature> &all = s.out->features;
        all.resize(total);
        if (rc == SG_OK && total)
            rc = sg_memcpy_d2h(r.ctx, all.data(), r.d_all, total * sizeof(SiftPackedFeature));
        if (rc == SG_OK) {
            s.out->first.assign((size_t)F + 1, 0);
            size_t at = 0;
            for (size_t f = 0; f < cnts.size(); f++) {
                const size_t n = std::min<size_t>(cnts[f], cap);
                if (f < (size_t)F)
                    s.out->first[f] = at;
                if (n)
                    last[f] = all[at + n - 1];
                at += n;
            }
            s.out->first[(size_t)F] = at; // frames beyond F are the padding of the last shard: no features
        }
    } else {
        size_t at = 0;
        for (size_t f = 0; f < cnts.size() && rc == SG_OK; f++) {
            const size_t n = std::min<size_t>(cnts[f], cap);
            if (n)
                rc = sg_memcpy_d2h(r.ctx, &last[f], static_cast<const SiftPackedFeature *>(r.d_all) + (at + n - 1),
                                   sizeof(SiftPackedFeature));
            at += n;
        }
    }
    if (rc != SG_OK)
        fail(rc, "download", r.ctx);
    s.barrier->wait();
}

} // namespace

int SiftBatchMultiGpu(const std::vector<Mat> &frames, SiftBatchFeatures &result, const SiftMultiGpuOptions &options,
                      std::string *error)
{
    auto bad = [&](int rc, const std::string &what) {
        if (error)
            *error = what;
        result.features.clear();
        result.first.assign(frames.size() + 1, 0);
        return rc;
    };
    result.features.clear();
    result.first.assign(frames.size() + 1, 0);
    if (frames.empty())
        return SG_OK;
    Shared s;
    s.frames = &frames;
    s.opt = options;
    s.n_dev = sg_device_count();
    if (s.n_dev < 1)
        return bad(SG_E_HIP, "no HIP device (there is no CPU fallback)");
    s.ranks = options.n_gpus > 0 ? options.n_gpus : s.n_dev;
    if (s.ranks > s.n_dev && !options.rehearsal)
        return bad(SG_E_INVALID, "more GPUs asked for than the node has");
    s.ranks = std::min<int>(s.ranks, (int)frames.size());
    s.w = frames[0].cols;
    s.h = frames[0].rows;
    s.ch = frames[0].channels();
    for (const Mat &m : frames)
        if (m.empty() || m.depth() != CV_8U || m.channels() < 3 || m.cols != s.w || m.rows != s.h || m.channels() != s.ch)
            return bad(SG_E_INVALID, "SiftBatchMultiGpu needs 8-bit frames of one size with at least 3 channels (b,g,r)");
    if (!(options.sigma > 1.0))
        return bad(SG_E_INVALID, "sigma must exceed 1");
    s.per = ((int)frames.size() + s.ranks - 1) / s.ranks;
    Barrier barrier(s.ranks);
    s.barrier = &barrier;
    s.status.assign((size_t)s.ranks, SG_OK);
    s.error.assign((size_t)s.ranks, std::string());
    s.need.assign((size_t)s.ranks, 0);
    s.total.assign((size_t)s.ranks, 0);
    s.seen_counts.resize((size_t)s.ranks);
    s.seen_last.resize((size_t)s.ranks);
    s.rehearsal_feats.resize((size_t)s.ranks);
    s.out = &result;
    // One call at a time: the ranks' contexts, buffers and communicators are kept between calls (SiftMultiGpuRelease
    // frees them) and belong to the call that runs.
    std::lock_guard<std::mutex> one_call(g_call_mutex);
    if (g_pool.size() < (size_t)s.ranks)
        g_pool.resize((size_t)s.ranks, nullptr);
    s.pool_comm_ok = pool_has_comm(s.ranks);
    if (s.ranks > 1 && !options.rehearsal && !s.pool_comm_ok) {
        const int rc = sg_comm_unique_id(s.comm_id);
        if (rc != SG_OK)
            return bad(rc, "sg_comm_unique_id: RCCL is not usable");
    }
    std::vector<std::thread> threads;
    for (int g = 1; g < s.ranks; g++)
        threads.emplace_back(rank_main, std::ref(s), g);
    rank_main(s, 0);
    for (std::thread &t : threads)
        t.join();
    for (int g = 0; g < s.ranks; g++)
        if (s.status[(size_t)g] != SG_OK) {
            SiftMultiGpuRelease_locked(); // whatever state the ranks were left in is not used again
            return bad(s.status[(size_t)g], "GPU " + std::to_string(g) + ": " + s.error[(size_t)g]);
        }
    if (options.rehearsal) {
        result.features.clear();
        for (int g = 0; g < s.ranks; g++) {
            size_t at = 0;
            for (int i = 0; i < s.per; i++) {
                const size_t f = (size_t)g * (size_t)s.per + (size_t)i, n = s.seen_counts[(size_t)g][(size_t)i];
                if (f < frames.size()) {
                    result.first[f] = result.features.size();
                    result.features.insert(result.features.end(), s.rehearsal_feats[(size_t)g].begin() + (ptrdiff_t)at,
                                           s.rehearsal_feats[(size_t)g].begin() + (ptrdiff_t)(at + n));
                }
                at += n;
            }
        }
        result.first[frames.size()] = result.features.size();
        return SG_OK;
    }
    // every GPU must hold the same gathered set: the counts and the sampled features of all ranks agree with rank 0's
    for (int g = 1; g < s.ranks; g++) {
        const bool same = s.seen_counts[(size_t)g] == s.seen_counts[0] &&
                          std::memcmp(s.seen_last[(size_t)g].data(), s.seen_last[0].data(),
                                      s.seen_last[0].size() * sizeof(SiftPackedFeature)) == 0;
        if (!same)
            return bad(SG_E_HIP, "GPU " + std::to_string(g) + " holds a different gathered set than GPU 0");
    }
    return SG_OK;
}

void SiftMultiGpuRelease()
{
    std::lock_guard<std::mutex> one_call(g_call_mutex);
    SiftMultiGpuRelease_locked();
}
