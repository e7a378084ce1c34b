/*
 * siftgpu.h -- C ABI of libsiftgpu, the MI355X (gfx950) SIFT extraction path.
 *
 * The library replaces the CPU pipeline behind the free functions of the reference's
 * sift/sift.h (Jia-Baos/SIFT).  Each entry point below names the reference interface it
 * stands in for (file:line into the reference tree).  Plain pointers and sizes only; no C++
 * or framework types cross this boundary.  A host-side C++ mirror of sift/sift.h that calls
 * these functions lives in sift_amd/host/.
 *
 * Conventions
 *   - every function returns SG_OK (0) or a negative sg_status; nothing throws, nothing prints
 *     unless sg_params.verbose is set;
 *   - outputs are caller-allocated with an explicit capacity; if a result does not fit the call
 *     returns SG_E_CAPACITY and stores the required count where a count is returned;
 *   - images are row-major; "stride" is in bytes for 8-bit input and in elements for double planes;
 *   - a context belongs to one HIP device and one stream; a context is not thread-safe, distinct
 *     contexts are independent;
 *   - there is no CPU fallback: without a usable HIP device sg_create fails with SG_E_HIP.
 */
#ifndef SIFTGPU_H
#define SIFTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sg_status {
    SG_OK = 0,
    SG_E_INVALID = -1,     /* bad argument */
    SG_E_HIP = -2,         /* HIP runtime error (see sg_last_error) */
    SG_E_CAPACITY = -3,    /* output or internal buffer too small */
    SG_E_UNSUPPORTED = -4, /* parameter combination the path does not implement */
    SG_E_NOMEM = -5
} sg_status;

typedef enum sg_precision {
    SG_F64 = 0, /* pyramid stored and filtered in double, like the reference's pixel_t (sift.h:16) */
    SG_F32 = 1, /* pyramid stored in float; keypoint arithmetic stays in double */
    SG_F64_FMA = 2 /* SG_F64 with fused multiply-adds in the Gaussian filters: one rounding per tap instead of the
                      reference's two, so pyramid values differ from a scalar evaluation of sift.cpp:323-326 in their
                      last bits (relative 1e-16); a tolerance mode like SG_F32, four times faster in the widest filters */
} sg_precision;

/* Same layout as the reference's struct Keypoint (sift/sift.h:75-102): 1120 bytes. */
typedef struct sg_keypoint {
    int octave;
    int interval;
    double offset_interval;
    int x;
    int y;
    double scale;
    double dx;
    double dy;
    double offset_x;
    double offset_y;
    double octave_scale;
    double ori;
    int descr_length;
    double descriptor[128];
    double val;
} sg_keypoint;

/* Compact feature for throughput mode: position/scale/orientation in the input image's frame
 * (Keypoint.dx, .dy, .scale, .ori) and the 128 integer-valued descriptor bins. 144 bytes. */
typedef struct sg_packed_feature {
    float x, y, scale, ori;
    uint8_t desc[128];
} sg_packed_feature;

/* The five counters DetectionLocalExtrema prints (sift/sift.cpp:984-986). */
typedef struct sg_counters {
    unsigned long long cc00, cc0, cc1, cc2, cc3;
} sg_counters;

/* Defaults are the macros of sift/sift.h:19-70. */
typedef struct sg_params {
    double sigma;      /* SIGMA 1.6; any value > 1 (the base blur is sqrt(sigma^2 - 1), sift.cpp:343): up to 1.68 every blur
                          runs in the unrolled fused kernels, above that the wide ones take the any-radius kernels */
    int intervals;     /* INTERVALS 3; only 3 is supported (the reference hard-wires the macro at
                          sift.cpp:870,886,1325,1613, so other values are inconsistent there) */
    int precision;     /* sg_precision */
    int nointerval;    /* 0: 3x3x3 extremum test (Sift); 1: in-layer 3x3 test (Sift_nointerval) */
    int max_width;     /* largest input image the context is sized for; the upsampled base plane must stay below
                          4 GB (about 134 megapixels of input with the double pyramid): SG_E_UNSUPPORTED beyond */
    int max_height;
    int max_batch;     /* images processed per sg_extract_batch* call */
    int max_features;  /* per-image feature capacity; 0 = derived from the image size */
    int device;        /* HIP device ordinal */
    int verbose;
    int keep_prequant; /* also keep the float descriptor before quantisation (debug/parity) */
} sg_params;

typedef struct sg_ctx sg_ctx;

void sg_default_params(sg_params *p);
const char *sg_strerror(int status);
const char *sg_last_error(const sg_ctx *ctx);
int sg_device_count(void);

int sg_create(const sg_params *p, sg_ctx **out);
int sg_destroy(sg_ctx *ctx);
int sg_sync(sg_ctx *ctx);
/* After a call has returned SG_E_CAPACITY for an internal buffer (3x3x3 extrema, keypoints or features of one image
 * exceeded what sg_params.max_features sized): the max_features a new context needs for that image.  The default
 * (max_features = 0) sizes for width*height/40 + 4096 features per image, about four times the density of natural
 * texture; the reference has no such limit, so its host mirror recreates the context and runs again. */
int sg_required_features(const sg_ctx *ctx, size_t *max_features);

/* ---- whole path ---------------------------------------------------------------------------- */

/* Sift(src, features, sigma, intervals) / Sift_nointerval (sift/sift.h:139,141; sift.cpp:1646-1770).
 * Host 8-bit image in (>=3 interleaved channels, b,g,r first), reference-layout records out, ordered
 * by ascending scale, exact ties in emission order (octave, interval, y, x, orientation bin). */
int sg_extract(sg_ctx *ctx, const uint8_t *px, int width, int height, int channels, size_t stride,
               sg_keypoint *out, size_t cap, size_t *n);

/* The same call in two steps, for callers that size their output from the count (the host mirror of Sift() appends
 * to a std::vector): sg_extract_begin runs the path and returns the number of features; sg_extract_fetch then writes
 * those records (it may be called more than once until the next sg_extract_begin).  sg_extract is one after the
 * other.  An internal-capacity overflow is reported by sg_extract_begin (SG_E_CAPACITY, see sg_required_features). */
int sg_extract_begin(sg_ctx *ctx, const uint8_t *px, int width, int height, int channels, size_t stride, size_t *n);
int sg_extract_fetch(sg_ctx *ctx, sg_keypoint *out, size_t cap, size_t *n);

/* A feature on its way to a sg_keypoint record: every field of the record, the descriptor as the 128 bytes it is
 * (the record's 128 doubles hold integers 0..255).  224 bytes instead of 1120: what crosses the bus. */
typedef struct sg_wire_feature {
    int32_t octave, interval, x, y;
    double offset_interval, scale, dx, dy, offset_x, offset_y, octave_scale, ori, val;
    double reserved;
    uint8_t desc[128];
} sg_wire_feature;

/* Several host images of one size through the path in one batch, for a host layer that serves Sift() calls of
 * several threads with one context (n_img <= sg_params.max_batch).  sg_extract_batch_begin uploads and runs the
 * path; counts[i] receives the number of features of image i.  `pinned` != 0 promises that every image lies in
 * page-locked memory (sg_host_alloc): the uploads are then direct DMA transfers; otherwise each image goes through
 * a page-locked buffer of the context.  sg_extract_batch_fetch_wire then writes the records of all images back to
 * back (image 0 first; image i starts at the sum of the counts before it) as sg_wire_feature, ordered per image
 * like sg_extract's; out should be page-locked too.  sg_expand_wire is the CPU step from those to reference-layout
 * records (any thread, no context).  Capacity overflow of an internal buffer: SG_E_CAPACITY from
 * sg_extract_batch_begin (see sg_required_features). */
int sg_extract_batch_begin(sg_ctx *ctx, const uint8_t *const *imgs, int n_img, int width, int height, int channels,
                           size_t stride, int pinned, uint32_t *counts);
int sg_extract_batch_fetch_wire(sg_ctx *ctx, sg_wire_feature *out, size_t cap_total);
/* The same download in pieces (what Sift() appends to the caller's vector, sift.cpp:1646-1707, on its way over the bus):
 * _begin packs the records and enqueues their transfer image after image in pieces of at most piece_records records
 * (same layout in `out` as above), *n_pieces receives how many; it returns without waiting.  _wait(k) returns once
 * piece k -- and with it every piece before it -- lies in `out`, so the caller can expand piece k while piece k + 1 is
 * still on the bus.  A new sg_extract_batch_begin (or any other call that runs the path) forgets the pieces. */
int sg_extract_batch_fetch_wire_begin(sg_ctx *ctx, sg_wire_feature *out, size_t cap_total, size_t piece_records,
                                      int *n_pieces);
int sg_extract_batch_fetch_wire_wait(sg_ctx *ctx, int piece);
void sg_expand_wire(const sg_wire_feature *in, size_t n, sg_keypoint *out);
/* Page-locked host memory for the two calls above (hipHostMalloc / hipHostFree). */
int sg_host_alloc(size_t bytes, void **ptr);
int sg_host_free(void *ptr);

/* Throughput form of the same path: n_img device-resident images of one size (image i starts at
 * d_imgs + i*img_stride), packed features written to d_out + i*cap_per_image, counts to d_counts[i].
 * Asynchronous on the context's stream; sg_sync() waits and reports capacity overflow. */
int sg_extract_batch_device(sg_ctx *ctx, const uint8_t *d_imgs, int n_img, int width, int height,
                            int channels, size_t stride, size_t img_stride,
                            sg_packed_feature *d_out, size_t cap_per_image, uint32_t *d_counts);

/* Host-pointer convenience form: uploads, runs sg_extract_batch_device, downloads. */
int sg_extract_batch(sg_ctx *ctx, const uint8_t *const *imgs, int n_img, int width, int height,
                     int channels, size_t stride, sg_packed_feature *out, size_t cap_per_image,
                     uint32_t *counts);

/* ---- stage entry points (host pointers; upload -> kernel -> download) ------------------------ */

/* ConvertToGray (sift.h:115, sift.cpp:20-51) */
int sg_convert_to_gray(sg_ctx *ctx, const uint8_t *px, int width, int height, int channels, size_t stride,
                       double *dst);
/* UpSample (sift.h:119, sift.cpp:111-175): dst is 2*height x 2*width */
int sg_upsample(sg_ctx *ctx, const double *src, int width, int height, double *dst);
/* DownSample (sift.h:117, sift.cpp:59-103): dst is (height/2) x (width/2) */
int sg_downsample(sg_ctx *ctx, const double *src, int width, int height, double *dst);
/* GaussianSmooth (sift.h:125, sift.cpp:323-326) = cv::GaussianBlur(src,dst,Size(0,0),sigma,sigma) */
int sg_gaussian_smooth(sg_ctx *ctx, const double *src, double *dst, int width, int height, double sigma);
/* CreateInitSmoothGray (sift.h:127, sift.cpp:335-346): dst is 2*height x 2*width */
int sg_create_init_smooth_gray(sg_ctx *ctx, const uint8_t *px, int width, int height, int channels,
                               size_t stride, double sigma, double *dst);
/* GaussianPyramid (sift.h:129, sift.cpp:357-407): levels[o*(intervals+3)+i] receives octave o, level i */
int sg_gaussian_pyramid(sg_ctx *ctx, const double *base, int width, int height, int octaves, int intervals,
                        double sigma, double *const *levels);
/* Sub (sift.h:131, sift.cpp:416-437): c = a - b */
int sg_sub(sg_ctx *ctx, const double *a, const double *b, double *c, int width, int height);
/* DogPyramid (sift.h:133, sift.cpp:447-461): dog[o*(intervals+2)+i-1] = gauss[o*(S+3)+i] - gauss[o*(S+3)+i-1] */
int sg_dog_pyramid(sg_ctx *ctx, const double *const *gauss, int width, int height, int octaves, int intervals,
                   double *const *dog);
/* DetectionLocalExtrema / _nointerval (sift.h:135,137; sift.cpp:920-1056) on a DoG pyramid whose octave 0
 * is height x width; records have octave, interval, x, y, offsets, dx, dy, val filled (sift.cpp:895-907,962). */
int sg_detect_local_extrema(sg_ctx *ctx, const double *const *dog, int width, int height, int octaves,
                            int intervals, int nointerval, sg_keypoint *out, size_t cap, size_t *n,
                            sg_counters *counters);
/* CalculateScale + HalfFeatures (sift.cpp:1065-1096), in place (host arithmetic on n records) */
int sg_calculate_scale(sg_keypoint *kp, size_t n, double sigma, int intervals);
int sg_half_features(sg_keypoint *kp, size_t n);
/* OrientationAssignment (sift.cpp:1317-1339): one output record per orientation peak */
int sg_orientation_assignment(sg_ctx *ctx, const sg_keypoint *extrema, size_t n_extrema,
                              const double *const *gauss, int width, int height, int octaves,
                              sg_keypoint *out, size_t cap, size_t *n);
/* DescriptorRepresentation (sift.cpp:1607-1631): fills descr_length and descriptor of each record.
 * Range: the histograms are 64-bit fixed-point sums that hold values below 2^14; pyramids the path makes itself
 * (pixels 0..1, window radius <= 42) stay below 300.  For planes and scales handed in here the call checks that
 * (largest - smallest pixel) * (largest window radius)^2 < 113000 and that every pixel is finite: SG_E_UNSUPPORTED /
 * SG_E_INVALID otherwise (the reference, which sums in double, has no such limit). */
int sg_descriptor_representation(sg_ctx *ctx, sg_keypoint *features, size_t n, const double *const *gauss,
                                 int width, int height, int octaves);
/* std::sort(features, FeatureCmp) (sift.cpp:1640-1643,1703): ascending scale, ties keep input order */
int sg_sort_features(sg_ctx *ctx, sg_keypoint *features, size_t n);

/* ---- inspection of the last sg_extract / batch call (parity tests, diagnostics) -------------- */

int sg_last_octaves(const sg_ctx *ctx);
int sg_level_dims(const sg_ctx *ctx, int octave, int *rows, int *cols);
/* bit o set: levels 1 and 2 of octave o of the last call were produced by one launch (timed under SG_ST_BLUR_PAIR) */
unsigned int sg_last_pair_octaves(const sg_ctx *ctx);
/* device bytes this context has allocated as scratch of sg_match / sg_match_packed_device (0 until the first such
 * call, grown on demand); sg_match_packed_device_noalloc never changes it */
size_t sg_match_scratch_bytes(const sg_ctx *ctx);
/* first octave of the last call whose levels all came from the one launch timed under SG_ST_PYR_TAIL (octaves small
 * enough for a workgroup's LDS); the octave count when there was no such launch */
int sg_last_tail_octave(const sg_ctx *ctx);
/* Gaussian level (octave, level 0..5) of image `image` of the last call, converted to double */
int sg_read_level(sg_ctx *ctx, int image, int octave, int level, double *dst);
int sg_read_counters(sg_ctx *ctx, int image, sg_counters *out);
/* Work done by the two histogram kernels in the last call, summed over its first n_img images: the number of
 * 64-sample trips of the descriptor drain loop and of the orientation window walk (one trip = one wave taking 64
 * window samples).  With the instruction counts of those loops this prices the kernels against the vector issue rate. */
int sg_read_work_counters(sg_ctx *ctx, int n_img, unsigned long long *descr_trips, unsigned long long *orient_trips);
/* keypoints after scale assignment and halving, in the reference's emission order */
int sg_read_keypoints(sg_ctx *ctx, int image, sg_keypoint *out, size_t cap, size_t *n);
/* features in emission order (before the scale sort) with descriptors; prequant (n x 128, may be NULL)
 * needs sg_params.keep_prequant */
int sg_read_features_unsorted(sg_ctx *ctx, int image, sg_keypoint *out, size_t cap, size_t *n, double *prequant);

/* ---- timing: HIP events recorded on the context's stream around each stage ------------------- */

typedef enum sg_stage {
    SG_ST_PREP = 0,       /* gray + upsample */
    SG_ST_BLUR_ROW0 = 1,  /* row-pass launches; +0 initial blur, +1..+5 pyramid levels 1..5 */
    SG_ST_BLUR_COL0 = 7,  /* column-pass launches, same sub-index */
    SG_ST_DOWNSAMPLE = 13,
    SG_ST_DETECT,         /* fused DoG + extrema scan */
    SG_ST_COMPACT,        /* ordered compaction launches */
    SG_ST_REFINE,
    SG_ST_ORIENT,
    SG_ST_DESCR,
    SG_ST_SORT,           /* rank + output packing */
    SG_ST_BLUR_PAIR,      /* pyramid levels 1 and 2 of an octave in one launch (not counted under the level slots) */
    SG_ST_PYR_TAIL,       /* every level of the smallest octaves in one launch (not counted under the level slots) */
    SG_ST_COUNT
} sg_stage;

int sg_set_profiling(sg_ctx *ctx, int on);
/* Accumulated milliseconds and launch counts per stage since the last reset; waits for the stream. */
int sg_get_stage_times(sg_ctx *ctx, double *ms /* SG_ST_COUNT */, unsigned long long *launches /* SG_ST_COUNT */);
int sg_reset_stage_times(sg_ctx *ctx);

/* ---- descriptor matching: the step after the path in the reference's demo --------------------- */

/* MatchPoint (sift/sift.h:105-110): indices into the two feature lists. */
typedef struct sg_match_point {
    int index_1, index_2;
} sg_match_point;

/* Brute-force matching with ratio test, the loop of the demo's main (sift/源.cpp:88-156): for every feature of
 * f1 the Euclidean distance over the 128 descriptor values to every feature of f2, the smallest (first index on
 * ties) and the second smallest; a match iff smallest / second < ratio (0.5 in the demo).  Distances are formed
 * per pair (the demo's `odest`, 源.cpp:97-110, is not reset between candidates; that carry-over is not
 * reproduced).  Descriptor values must be integers 0..255, as the path produces them (SG_E_INVALID otherwise).
 * Matches come out in index_1 order; SG_E_CAPACITY with *n = needed count when cap is too small. */
int sg_match(sg_ctx *ctx, const sg_keypoint *f1, size_t n1, const sg_keypoint *f2, size_t n2, double ratio,
             sg_match_point *out, size_t cap, size_t *n);

/* Device form on packed features (the output of sg_extract_batch_device): d_index2[i] = index of the match of
 * feature i of d_f1 in d_f2, or -1; d_dist2 (optional, 2 per feature) = smallest and second smallest squared
 * distance (0xffffffff where there is none).  Asynchronous on the context's stream; sg_sync waits. */
int sg_match_packed_device(sg_ctx *ctx, const sg_packed_feature *d_f1, uint32_t n1, const sg_packed_feature *d_f2,
                           uint32_t n2, double ratio, int32_t *d_index2, uint32_t *d_dist2);

/* Allocation-free form of sg_match_packed_device: same arguments, same results, but no hipMalloc/hipFree and no
 * allocating library call anywhere in it.  It has no counterpart in the reference; it exists for callers that
 * cannot allocate: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Its scratch
 * (signed-byte rows, norms and per-slice partial results, 14.7 MB) is a static device array that every GPU the
 * library runs on holds until the process ends, so the sets are limited to SG_MATCH_NOALLOC_MAX features each
 * (SG_E_CAPACITY beyond that).  All calls on one GPU share that array: calls that overlap on one GPU (two
 * streams, two contexts, two threads) race; serialise them.  Asynchronous on the context's stream. */
#define SG_MATCH_NOALLOC_MAX 32768
int sg_match_packed_device_noalloc(sg_ctx *ctx, const sg_packed_feature *d_f1, uint32_t n1,
                                   const sg_packed_feature *d_f2, uint32_t n2, double ratio, int32_t *d_index2,
                                   uint32_t *d_dist2);

/* ---- multi-GPU: the one exchange step ------------------------------------------------------------------------
 * The reference has no multi-GPU path (it has no parallelism at all); frames are independent (Sift() is a pure
 * function of one image, sift/sift.cpp:1646-1707), so a batch is sharded by frame, one rank per GPU, and the only
 * exchange is the all-gather of the ranks' packed feature sets after extraction.
 *
 * sg_allgather_features: every rank passes the output of its sg_extract_batch_device call (d_feats: n_img slots of
 * cap_per_image features, d_counts: n_img counts; the same n_img and cap_per_image on every rank) and receives in
 * d_all the features of all ranks' frames, rank after rank, frame after frame, without padding, in d_all_counts
 * ([n_ranks][n_img], device) every frame's count, and in rank_first (host, n_ranks + 1 entries) the index of each
 * rank's first feature in d_all (rank_first[n_ranks] = total).  SG_E_CAPACITY when the total exceeds cap_all.
 * `comm` is an RCCL communicator (ncclComm_t, passed as void* so that this header needs no rccl.h) created by the
 * caller or by sg_comm_create; NULL means a single rank (no RCCL call is made).  The counts are all-gathered and read
 * by the host (stream synchronisations); the features then move as one grouped set of ncclBroadcast calls (one per
 * rank's slot) or -- SIFTGPU_ALLGATHERV=p2p in the environment, read at every call, the same on every rank -- point to
 * point: every rank sends its slot to every peer and receives every peer's, one group of ncclSend / ncclRecv; over a
 * fully connected xGMI node a slot then crosses each link once.  Either way asynchronous on the context's stream
 * (sg_sync waits).  Call it on a context whose stream is not the one a next
 * batch is being extracted on if the exchange is to overlap that extraction.
 * Errors are decided by all ranks together where they can be: the ranks first exchange (cap_all, n_img,
 * cap_per_image); differing n_img or cap_per_image give SG_E_INVALID on every rank, and the capacity test uses the
 * smallest cap_all, so SG_E_CAPACITY is returned by every rank or by none and no rank is left waiting in a
 * collective.  Any other failure (SG_E_HIP: a HIP or RCCL error on one rank) cannot be agreed on: after it the
 * communicator must be destroyed, and the other ranks' calls may not return.
 * The path with more than one rank needs one GPU per rank; it has been exercised on multi-GPU hardware only through
 * bench.py's self-check (exchange_check), which every multi-rank run performs. */
#define SG_COMM_ID_BYTES 128
int sg_comm_unique_id(void *id /* SG_COMM_ID_BYTES, filled on one rank and handed to all */);
int sg_comm_create(sg_ctx *ctx, const void *id, int rank, int n_ranks, void **comm); /* on ctx's device */
int sg_comm_destroy(void *comm);
int sg_allgather_features(sg_ctx *ctx, void *comm, const sg_packed_feature *d_feats, const uint32_t *d_counts,
                          int n_img, size_t cap_per_image, sg_packed_feature *d_all, size_t cap_all,
                          uint32_t *d_all_counts, size_t *rank_first);
/* The local half of that step on its own: n_img padded slots -> frames back to back at d_out (room for the sum of
 * the counts; 16-byte aligned).  Asynchronous on the context's stream. */
int sg_compact_features(sg_ctx *ctx, const sg_packed_feature *d_feats, const uint32_t *d_counts, int n_img,
                        size_t cap_per_image, sg_packed_feature *d_out);

/* ---- device memory and synthetic input (so that callers need no HIP binding of their own) ---- */

int sg_malloc(sg_ctx *ctx, size_t bytes, void **d_ptr);
int sg_free(sg_ctx *ctx, void *d_ptr);
int sg_memcpy_h2d(sg_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int sg_memcpy_d2h(sg_ctx *ctx, void *dst, const void *d_src, size_t bytes);
/* 5-octave value-noise test image (b=g=r), generated on the device: height x width x 3 bytes */
int sg_synth_vnoise5(sg_ctx *ctx, uint8_t *d_dst, int width, int height, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
