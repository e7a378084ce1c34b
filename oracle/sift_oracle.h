/*
 * sift_oracle.h -- CPU oracle for the SIFT extraction path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, double-precision, single-threaded restatement of the algorithm in the
 * reference's sift/sift.cpp (Jia-Baos/SIFT).  It exists so that the GPU path can be
 * checked against it; it is never linked into, loaded by or called from the product
 * library.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors, and it cannot
 * be built in this image (it needs OpenCV, which is absent).  The one place where the
 * reference's arithmetic leaves its own sources is cv::GaussianBlur (OpenCV imgproc,
 * version not pinned by the reference); so_gaussian_blur() restates OpenCV's published
 * algorithm for CV_64F input (kernel size rule, truncated + renormalised taps,
 * BORDER_REFLECT_101, generic row filter followed by the symmetric column filter).
 *
 * Every function cites the reference lines it follows.
 */
#ifndef SIFT_ORACLE_H
#define SIFT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as the reference's `struct Keypoint` (sift/sift.h:75-102): 1120 bytes. */
typedef struct so_keypoint {
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
} so_keypoint;

/* Gaussian + DoG pyramids, flat like the reference's vector<Mat>:
 * gauss[o*(S+3)+i], dog[o*(S+2)+i]; every level of octave o is rows[o] x cols[o], contiguous. */
typedef struct so_pyramid {
    int octaves;
    int intervals;
    int *rows;
    int *cols;
    double **gauss;
    double **dog;
} so_pyramid;

/* Counters printed by the reference's DetectionLocalExtrema (sift.cpp:984-986). */
typedef struct so_counters {
    long cc00, cc0, cc1, cc2, cc3;
} so_counters;

/* Stage timings of one so_sift() call, seconds. */
typedef struct so_timings {
    double init, gauss, dog, detect, orient, descr, sort, total;
} so_timings;

size_t so_sizeof_keypoint(void);

/* ---- image preparation (sift.cpp:20-175) ---- */
void so_convert_to_gray(const uint8_t *src, int rows, int cols, int channels, size_t step, double *dst);
void so_upsample(const double *src, int rows, int cols, double *dst /* 2rows x 2cols */);
void so_downsample(const double *src, int rows, int cols, double *dst /* rows/2 x cols/2 */);

/* ---- blur: cv::GaussianBlur(src,dst,Size(0,0),sigma,sigma) on CV_64FC1 (sift.cpp:323-326) ---- */
int so_gaussian_ksize(double sigma);
void so_gaussian_kernel(double sigma, int n, double *k);
void so_gaussian_blur(const double *src, double *dst, int rows, int cols, double sigma);

/* ---- base image (sift.cpp:335-346) ---- */
void so_create_init_smooth_gray(const uint8_t *src, int rows, int cols, int channels, size_t step,
                                double sigma, double *dst /* 2rows x 2cols */);
int so_num_octaves(int base_rows, int base_cols); /* sift.cpp:1653 */
void so_pyramid_sigmas(double sigma, int intervals, double *sigmas /* intervals+3 */);

/* ---- pyramids (sift.cpp:357-461) ---- */
so_pyramid *so_build_pyramid(const double *base, int rows, int cols, int octaves, int intervals, double sigma);
void so_free_pyramid(so_pyramid *p);

/* ---- detection (sift.cpp:473-1056) ---- */
int so_inverse3d(const double *H, double *H_inv); /* sift.cpp:720-764 */
/* Appends to *out (realloc'd); returns the new count. nointerval selects the 2-D extremum test. */
size_t so_detect(const so_pyramid *p, int nointerval, so_keypoint **out, size_t count, so_counters *ctr);

/* ---- scale, orientation, descriptors, ordering (sift.cpp:1065-1643) ---- */
void so_calculate_scale(so_keypoint *kp, size_t n, double sigma, int intervals);
void so_half_features(so_keypoint *kp, size_t n);
void so_orientation_histogram(const double *gauss, int rows, int cols, int x, int y, int radius,
                              double sigma, double *hist /* 36, raw (unsmoothed) */);
size_t so_orientation_assignment(const so_keypoint *extrema, size_t n, const so_pyramid *p,
                                 so_keypoint **features, size_t count);
/* float descriptor before quantisation (after the second normalisation), may be NULL */
void so_descriptor_one(so_keypoint *f, const so_pyramid *p, double *prequant /* 128 or NULL */);
void so_descriptors(so_keypoint *f, size_t n, const so_pyramid *p);
/* ascending scale; ties keep emission order (the reference's std::sort is unstable on ties,
 * sift.cpp:1640-1643,1703, so any tie order is a valid reference order) */
void so_sort_features(so_keypoint *f, size_t n);

/* ---- drivers (sift.cpp:1646-1770) ---- */
/* Returns a malloc'd array (caller frees with so_free) of *n features. */
so_keypoint *so_sift(const uint8_t *src, int rows, int cols, int channels, size_t step,
                     double sigma, int intervals, int nointerval,
                     size_t *n, size_t *n_keypoints, so_counters *ctr, so_timings *tm);
/* Same, but also hands back the unsorted features and the pyramid for per-stage checks. */
so_keypoint *so_sift_debug(const uint8_t *src, int rows, int cols, int channels, size_t step,
                           double sigma, int intervals, int nointerval,
                           size_t *n, so_keypoint **keypoints, size_t *n_keypoints,
                           so_pyramid **pyr, so_counters *ctr);
void so_free(void *p);

/* Runs so_sift on `n_images` images (each rows x cols x 3, tightly packed) with `threads`
 * worker threads, one image per thread at a time. Returns elapsed wall seconds; counts[i] = features. */
double so_sift_batch(const uint8_t *imgs, int n_images, int rows, int cols, int threads, size_t *counts);

/* ---- text feature format (sift.cpp:1886-1931) ---- */
int so_write_features(const so_keypoint *f, size_t n, const char *path);

/* ---- brute-force descriptor matching with ratio test (demo main, sift/源.cpp:88-156) ----
 * For every feature of f1: Euclidean distance to every feature of f2 over the 128 descriptor values, smallest
 * (first index on ties: strict <) and second smallest value, match iff smallest / second < ratio (0.5 in the demo).
 * The demo's distance variable is not reset between candidates (源.cpp:97,107,110: `odest` carries the previous
 * square root into the next sum); accumulate_bug = 0 resets it per pair (the intended metric, what the GPU path
 * computes), accumulate_bug = 1 reproduces the demo literally.  Returns the number of matches written to
 * index_1 / index_2 (each with room for n1 entries), in index_1 order.  best2 (optional, n1 x 2) receives the
 * smallest and second smallest squared distances of the reset form. */
size_t so_match(const so_keypoint *f1, size_t n1, const so_keypoint *f2, size_t n2, double ratio, int accumulate_bug,
                int *index_1, int *index_2, double *best2);

/* ---- synthetic input: 5-octave value noise, stateless hash; b=g=r ---- */
void so_vnoise5(uint8_t *dst /* rows x cols x 3 */, int rows, int cols, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
