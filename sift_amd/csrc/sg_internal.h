// sg_internal.h -- shared declarations of libsiftgpu's HIP sources (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "siftgpu.h"

namespace sg {

constexpr int kMaxOctaves = 16;
constexpr int kLevels = 6;        // INTERVALS + 3 Gaussian levels per octave
constexpr int kDogLevels = 5;     // INTERVALS + 2
constexpr int kMaxFastRadius = 13;
constexpr int kMaxTaps = 2 * kMaxFastRadius + 1;
constexpr int kMaxAnyTaps = 1024; // widest kernel the any-radius blur takes
constexpr int kBorder = 5;        // IMG_BORDER, sift.h:58
constexpr int kChunkWords = 1024; // bitmap words per compaction chunk
constexpr int kOriBins = 36;
// Trip counts of the histogram kernels (diagnostics: sg_read_work_counters) are added into kTripSlots words per image
// and kernel, 64 bytes apart, the slot chosen by the workgroup's number: a launch's waves all finish at about the same
// time, and their adds to one word would queue up at the memory side behind each other (a frame alone: 5 000 adds, 40 us).
constexpr int kTripSlots = 64, kTripStride = 8; // slots per (image, kernel); 64-bit words from slot to slot
constexpr size_t kTripWordsPerImage = 2 * (size_t)kTripSlots * kTripStride; // [0]: descriptors, [1]: orientation
constexpr int kRankBuckets = 16384; // 16 binades x 1024 mantissa steps of the scale key

// Debug / comparison switches, read from the environment once per process (first sg_create) and never on a launch path:
//   SIFTGPU_BLUR=2pass|ring|win   blur variant (default: the register-window kernels with LDS-DMA staging)
//   SIFTGPU_BLUR_DIRECT=1         input rows loaded by the lanes instead of LDS-DMA
//   SIFTGPU_BLUR_PAIR=0|2         pyramid levels 1 and 2 always as two launches | as one wherever the plane allows
//                                 (default: one launch where a wave's row segment is long)
//   SIFTGPU_DETECT_DIRECT=1       detect rows loaded into registers instead of LDS-DMA
//   SIFTGPU_ORI_WAVES / SIFTGPU_DESCR_WAVES   total waves of the orientation / descriptor launches
struct Env {
    int blur_mode = 0; // 0 default, 1 "2pass", 2 "ring", 3 "win"
    bool blur_direct = false, detect_direct = false;
    int blur_pair = 1; // SIFTGPU_BLUR_PAIR: 0 levels 1 and 2 always as two launches, 1 one launch where segments are
                       // long (default), 2 one launch wherever the plane is large enough (tests)
    int blur_remap = 0;                 // SIFTGPU_BLUR_REMAP=1: XCD-aware tile order of the strip-walking blur launches
    bool pyr_tail = true;               // SIFTGPU_PYR_TAIL=0: the smallest octaves level by level, like the large ones
    int ori_waves = 0, descr_waves = 0; // 0 = library default
    int n_cu = 256;                     // compute units of the current device at first use
};
const Env &env();

// The `prec` argument of the blur launchers: an sg_precision (the pyramid's storage type) in the low byte, and
// kPrecFma where the double filters take fused multiply-adds (sg_params.precision = SG_F64_FMA).
constexpr int kPrecFma = 0x100;
inline int prec_store(int prec) { return prec & 0xff; }
inline bool prec_fma(int prec) { return (prec & kPrecFma) != 0; }

// One image plane (or a batch of them): element pointer, geometry in elements.
struct Plane {
    void *p = nullptr;
    int rows = 0, cols = 0, pitch = 0;
    size_t img_stride = 0; // elements between images of a batch
};

// Filter taps handed to the blur kernels by value (zero-padded to 2*R+1 of the chosen kernel).
template <typename T>
struct Taps {
    T k[kMaxTaps];
};

// Per-octave table handed to the keypoint kernels by value.
struct OctaveTab {
    const void *lv[kMaxOctaves][kLevels]; // Gaussian levels (or DoG levels 0..4 when from_dog)
    int rows[kMaxOctaves], cols[kMaxOctaves], pitch[kMaxOctaves];
    unsigned long long img_stride[kMaxOctaves];
    unsigned int wbase[kMaxOctaves + 1]; // first bitmap word of each octave
    int nrows[kMaxOctaves];              // scanned rows per layer = rows - 2*border (>= 0)
    int wpr[kMaxOctaves];                // bitmap words per scanned row
    int octaves;
    int from_dog;
};

// Device keypoint record (after refinement): 80 bytes.
struct KpRec {
    int x, y, octave, interval;
    double offx, offy, offi;
    double dx, dy, scale, oscale, val;
};

// Per-feature inputs of the descriptor stage, gathered (and cos/sin of the orientation evaluated) once per
// feature by one thread, so that the 256 threads of a descriptor workgroup need a single 64-byte load.
struct FeatRec {
    int x, y, octave, interval;
    double oscale, ori, scale;
    double cos_ori, sin_ori;
    unsigned int kp_slot, pad;
};

// A feature as it crosses the bus towards sg_keypoint records on the host: every field of the record, with the
// descriptor as the 128 bytes it is rather than 128 doubles (224 bytes instead of 1120); expanded by the CPU.
struct WireRec {
    int octave, interval, x, y;
    double offi, scale, dx, dy, offx, offy, oscale, ori, val;
    double pad;
    uint8_t desc[128];
};
static_assert(sizeof(WireRec) == 224 && offsetof(WireRec, desc) == 96, "16-byte pieces");

// ---- image preparation -----------------------------------------------------------------
void launch_gray(const uint8_t *src, int rows, int cols, int ch, size_t stride, size_t img_stride,
                 const Plane &dst, int B, int prec, hipStream_t s);
void launch_upsample(const Plane &src, const Plane &dst, int B, int prec, hipStream_t s);
void launch_gray_upsample(const uint8_t *src, int rows, int cols, int ch, size_t stride, size_t img_stride,
                          const Plane &dst, int B, int prec, hipStream_t s);
void launch_downsample(const Plane &src, const Plane &dst, int B, int prec, hipStream_t s);
void launch_sub(const Plane &a, const Plane &b, const Plane &c, int B, int prec, hipStream_t s);
void launch_vnoise5(uint8_t *dst, int rows, int cols, uint64_t seed, hipStream_t s);

// ---- blur ---------------------------------------------------------------------------------
// taps: n doubles (host), n odd. Row pass src->tmp, column pass tmp->dst. In-place (src==dst) is allowed.
// d_taps (optional): the n taps already on the device in the plane's type; without it a blur wider than the unrolled
// kernels uploads them itself (allocating and synchronising); false (nothing launched) when that upload fails
bool launch_blur_row(const Plane &src, const Plane &dst, int B, const double *taps, int n, int prec, hipStream_t s,
                     const void *d_taps = nullptr);
bool launch_blur_col(const Plane &src, const Plane &dst, int B, const double *taps, int n, int prec, hipStream_t s,
                     const void *d_taps = nullptr);
// Row and column pass in one launch (src != dst). Returns false, launching nothing, where only the two-pass
// kernels apply (1-pixel-wide or -high planes, radius beyond the fast kernels).
// true when launch_blur_fused handles this plane and tap count (else the two-pass kernels must be used)
bool blur_fused_supported(const Plane &src, int n);
bool launch_blur_fused(const Plane &src, const Plane &dst, int B, const double *taps, int n, int prec, hipStream_t s,
                       const Plane *dec = nullptr);
// dst = blur(UpSample(gray)) in one launch; gray is a double plane of half dst's size.  False when the form does not
// apply (the caller then upsamples into a plane and blurs that).
bool launch_blur_fused_upsampled(const Plane &gray, const Plane &dst, int B, const double *taps, int n, int prec,
                                 hipStream_t s);
// dst_a = blur(src, taps_a) and dst_b = blur(dst_a, taps_b) in one launch; false (nothing launched) where that form
// does not apply (the caller then runs the two blurs one after the other)
bool launch_blur_pair(const Plane &src, const Plane &dst_a, const Plane &dst_b, int B, const double *taps_a, int n_a,
                      const double *taps_b, int n_b, int prec, hipStream_t s);
// The smallest octaves of a pyramid in one launch: one workgroup per image holds an octave's plane in LDS and produces
// levels 1..5 of octaves first..tab.octaves-1 (and level 0 of every octave after `first`, by 2:1 decimation of level 3)
// from level 0 of octave `first`, which must be in place.  taps[l-1], ntaps[l-1]: kernel of level l.
// pyr_tail_first: the first octave (>= 1) from which that form applies, tab.octaves when none does.
int pyr_tail_first(const OctaveTab &tab, const int *ntaps, int prec);
bool launch_pyr_tail(const OctaveTab &tab, int first, int B, const double *const *taps, const int *ntaps, int prec,
                     hipStream_t s);
// true when launch_blur_fused can also write the 2:1 decimated copy of its output (DownSample) into `dec`
bool blur_fused_decimates(const Plane &src, int n, int prec);

// ---- detection --------------------------------------------------------------------------------
// all octaves of the table; returns the number of launches
int launch_detect(const OctaveTab &tab, int B, unsigned long long *words, size_t words_img_stride,
                  unsigned int *chunk_sums, size_t chunks_img_stride, unsigned long long *counters,
                  double thresh, int nointerval, int prec, hipStream_t s, int o_from = 0, int o_to = -1);
// exclusive scan of chunk_sums (n_chunks per image) in place; totals[b] = sum (clamped report of overflow)
void launch_scan(unsigned int *chunk_sums, size_t chunks_img_stride, int n_chunks, unsigned int *totals,
                 unsigned int cap, unsigned int *overflow, int B, hipStream_t s);
// expand set bits of words (n_words per image) into out[b][pos] = word*64 + bit, ordered
// host-side copy into a page-locked staging buffer (streaming stores where the CPU has them; sg_wire.cpp)
void copy_to_staging(void *dst, const void *src, size_t len);
// per-chunk popcounts of a bitmap (n_chunks values per image), for producers that leave the counting to this launch
void launch_chunk_popc(const unsigned long long *words, size_t words_img_stride, unsigned int n_words,
                       const unsigned int *dyn_items, unsigned int items_per_word, unsigned int *chunk_sums,
                       size_t chunks_img_stride, int n_chunks, int B, hipStream_t s);
void launch_expand(const unsigned long long *words, size_t words_img_stride, unsigned int n_words,
                   const unsigned int *dyn_items, unsigned int items_per_word, const unsigned int *chunk_offs,
                   size_t chunks_img_stride, unsigned int *out,
                   size_t out_img_stride, unsigned int cap, int B, hipStream_t s, bool dense = false);
void launch_refine(const OctaveTab &tab, const unsigned int *cand, size_t cand_img_stride,
                   const unsigned int *cand_count, KpRec *kp, size_t kp_img_stride,
                   unsigned long long *valid_words, size_t vw_img_stride, unsigned int *chunk_sums,
                   size_t chunks_img_stride, unsigned long long *counters, double sigma, unsigned int cap, int B,
                   int prec, hipStream_t s);

// ---- orientation, descriptors, ordering ---------------------------------------------------------
void launch_orient(const OctaveTab &tab, const KpRec *kp, size_t kp_img_stride, const unsigned int *kp_src,
                   size_t src_img_stride, const unsigned int *kp_count, double *ori_tmp, size_t ori_img_stride,
                   unsigned long long *peak_words, size_t pw_img_stride, unsigned int *chunk_sums,
                   size_t chunks_img_stride, int n_chunks, unsigned int cap, int max_radius, int B, int prec,
                   unsigned long long *trip_slots, hipStream_t s);
// gathers FeatRec[f] (and fscale[f]) from feat_src -> kp_src -> kp / ori_tmp
void launch_feat_gather(const KpRec *kp, size_t kp_img_stride, const unsigned int *kp_src, size_t src_img_stride,
                        const unsigned int *feat_src, size_t fsrc_img_stride, const unsigned int *feat_count,
                        const double *ori_tmp, size_t ori_img_stride, FeatRec *frec, size_t frec_img_stride,
                        double *fscale, size_t fs_img_stride, unsigned int cap, int B, hipStream_t s);
// max_radius: an upper bound of the features' window radii (selects the kernel variant for very wide windows)
void launch_descr(const OctaveTab &tab, const FeatRec *frec, size_t frec_img_stride, const unsigned int *feat_count,
                  uint8_t *desc, size_t desc_img_stride, double *prequant, size_t pq_img_stride, unsigned int cap,
                  int max_radius, int B, int prec, unsigned long long *trip_slots, hipStream_t s);
// An entry of a scale bucket's list: the feature's index with its scale beside it, so that the counting inside a
// bucket reads one contiguous stream instead of following every index to its scale.
struct RankEntry {
    double scale;
    unsigned int index, pad;
};
// bwork: [B][2][kRankBuckets] scratch, blist: [B][cap] scratch, scratch_totals: [B], overflow_dummy: [1]
void launch_rank(const double *fscale, size_t fs_img_stride, const unsigned int *feat_count, unsigned int *rank,
                 size_t rank_img_stride, unsigned int *bwork, RankEntry *blist, unsigned int *scratch_totals,
                 unsigned int *overflow_dummy, unsigned int cap, int B, hipStream_t s);
// mode 0: packed features (sg_packed_feature) at out + b*out_img_stride*144; mode 1: sg_keypoint records;
// mode 3: WireRec records (out 16-byte aligned).
void launch_pack(const KpRec *kp, size_t kp_img_stride, const unsigned int *kp_src, size_t src_img_stride,
                 const unsigned int *feat_src, size_t fsrc_img_stride, const unsigned int *feat_count,
                 const double *ori_tmp, size_t ori_img_stride, const uint8_t *desc, size_t desc_img_stride,
                 const unsigned int *rank, size_t rank_img_stride, void *out, size_t out_img_stride,
                 unsigned int out_cap, unsigned int *out_counts, unsigned int *overflow, int mode, int sorted,
                 unsigned int cap, int B, hipStream_t s);
// descriptor matching (kernels_match.hip).  rows: n_pad x 128 signed bytes (value - 128), norms: n_pad squared norms;
// n_pad = n rounded up to 32 (at least 32).  mode 0: sg_packed_feature records, 1: sg_keypoint records.
void launch_match_prep(const void *feats, int mode, unsigned int n, unsigned int n_pad, signed char *rows, int *norms,
                       unsigned int *bad, hipStream_t s);
// partials: room for n1 (rounded up to 32) * kMatchMaxSlices * 12 bytes; slices from match_slices (1: partials unused)
constexpr unsigned int kMatchMaxSlices = 16;
unsigned int match_slices(unsigned int n1, unsigned int n2);
void launch_match(const signed char *rows1, const int *norms1, unsigned int n1, const signed char *rows2,
                  const int *norms2, unsigned int n2, double ratio, int *index2, unsigned int *dist2, void *partials,
                  unsigned int slices, hipStream_t s);

// ---- multi-GPU exchange (sg_comm.hip) ------------------------------------------------------------------------------
// [n_img][cap] padded feature slots -> frames back to back at out (16-byte aligned)
void launch_compact_feats(const sg_packed_feature *in, const unsigned int *counts, int n_img, size_t cap,
                          sg_packed_feature *out, hipStream_t s);
// RCCL bound at run time; the int results are ncclResult_t values (0 = success), -1 when RCCL cannot be loaded
const char *rccl_unavailable_reason(); // nullptr when RCCL is usable
int rccl_unique_id(void *id128);
int rccl_comm_create(const void *id128, int rank, int n_ranks, void **comm);
int rccl_comm_destroy(void *comm);
int rccl_comm_shape(void *comm, int *rank, int *n_ranks);
const char *rccl_error_string(int rc);
int rccl_allgather_u32(void *comm, const unsigned int *send, unsigned int *recv, size_t count, hipStream_t s);
int rccl_allgatherv_bytes(void *comm, char *base, const size_t *first_byte, int rank, int n_ranks, hipStream_t s);

} // namespace sg
