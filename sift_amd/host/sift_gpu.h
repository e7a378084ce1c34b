// sift_gpu.h -- host C++ interface of the MI355X SIFT path: the entry points of the reference's
// sift/sift.h (Jia-Baos/SIFT), same names, argument meaning and record layout, backed by libsiftgpu.
//
// A program written against the reference includes this header instead of its sift.h and links
// libsift_host.so + libsiftgpu.so; nothing else changes for the extraction path
//   Sift / Sift_nointerval and the stage functions ConvertToGray ... DetectionLocalExtrema,
//   plus CalculateScale, HalfFeatures, OrientationAssignment, DescriptorRepresentation, FeatureCmp
//   (defined with external linkage in the reference's sift.cpp:1065-1643 though absent from its header)
//   and the text feature files (write_features / read_features, sift.cpp:1886-1931).
//   and the drawing and pyramid-dump helpers (sift.cpp:1772-1884; host code, no GPU involved).
// The reference's off-path helpers (the 8-bit legacy blurs and testInverse3D, sift.cpp:183-314, 1934-1960) are
// declared at the end as plain CPU functions, so that a program that names them still links.
//
// Like the reference these functions return void and append to the vectors they are given; a failed
// GPU call leaves the outputs untouched and is reported by SiftLastStatus() / SiftLastError().
#pragma once

#include <string>
#include <vector>

#ifdef SIFT_WITH_OPENCV
#include <opencv2/core/core.hpp>
#else
#include "cvmini.h"
#endif

using cv::Mat; // the reference's header has `using namespace cv` at global scope (sift.h:9-10)

// ---- parameters: the macros of sift/sift.h:16-70, same names and values -------------------------
typedef double pixel_t;               // pyramid sample type
#define INIT_SIGMA 0.5                // assumed blur of the input image
#define SIGMA 1.6                     // blur of octave 0, level 0
#define INTERVALS 3                   // sampled intervals per octave
#define RATIO 10                      // principal-curvature ratio of the edge test
#define MAX_INTERPOLATION_STEPS 5     // sub-pixel refinement moves before a candidate is dropped
#define DXTHRESHOLD 0.03              // contrast threshold on |D(x)|
#define ORI_HIST_BINS 36              // orientation histogram bins
#define ORI_SIGMA_TIMES 1.5           // orientation window sigma = this * octave_scale
#define ORI_WINDOW_RADIUS 3.0 * ORI_SIGMA_TIMES // orientation window radius factor (unparenthesised as in the reference)
#define ORI_SMOOTH_TIMES 2            // [1/4 1/2 1/4] passes over the orientation histogram
#define ORI_PEAK_RATIO 0.8            // peaks above this share of the maximum spawn a feature
#define FEATURE_ELEMENT_LENGTH 128    // descriptor length
#define DESCR_HIST_BINS 8             // orientation bins per descriptor cell
#define IMG_BORDER 5                  // border not scanned for extrema
#define DESCR_WINDOW_WIDTH 4          // descriptor cells per side
#define DESCR_SCALE_ADJUST 3          // cell width = this * octave_scale
#define DESCR_MAG_THR 0.2             // clamp of the normalised descriptor
#define INT_DESCR_FCTR 512.0          // float -> integer descriptor factor

// ---- records: layout of sift/sift.h:75-110 (Keypoint is 1120 bytes) ---------------------------------
struct Keypoint {
    int octave;             // octave of the DoG layer the point was found in
    int interval;           // layer inside the octave after refinement
    double offset_interval; // sub-layer offset
    int x;                  // column in that octave's image
    int y;                  // row in that octave's image
    double scale;           // sigma * 2^(octave + (interval+offset)/S), in input-image pixels
    double dx;              // x in input-image pixels
    double dy;              // y in input-image pixels
    double offset_x;        // sub-pixel offsets
    double offset_y;
    double octave_scale;    // sigma * 2^((interval+offset)/S): scale inside the octave
    double ori;             // orientation, radians in [-pi, pi)
    int descr_length;       // 128 once described
    double descriptor[FEATURE_ELEMENT_LENGTH]; // integer-valued 0..255
    double val;             // DoG value at the refined position
};

struct MatchPoint {
    int index_1; // feature index in image 1
    int index_2; // nearest feature in image 2
};

// ---- image preparation and pyramids (sift.h:115-133) -------------------------------------------------
void ConvertToGray(const Mat &src, Mat &dst);  // CV_8UC3+ (b,g,r) -> CV_64FC1 in [0,1]
void DownSample(const Mat &src, Mat &dst);     // every second row and column
void UpSample(const Mat &src, Mat &dst);       // 2x with the reference's edge replication
void GaussianSmooth(const Mat &src, Mat &dst, double sigma);
void CreateInitSmoothGray(const Mat &src, Mat &dst, double sigma = SIGMA);
void GaussianPyramid(const Mat &src, std::vector<Mat> &gauss_pyr, int octaves, int intervals = INTERVALS,
                     double sigma = SIGMA);
void Sub(const Mat &a, const Mat &b, Mat &c);
void DogPyramid(const std::vector<Mat> &gauss_pyr, std::vector<Mat> &dog_pyr, int octaves, int intervals = INTERVALS);

// ---- detection (sift.h:135-137) ------------------------------------------------------------------------
void DetectionLocalExtrema(const std::vector<Mat> &dog_pyr, std::vector<Keypoint> &extrema, int octaves,
                           int intervals = INTERVALS);
void DetectionLocalExtrema_nointerval(const std::vector<Mat> &dog_pyr, std::vector<Keypoint> &extrema, int octaves,
                                      int intervals = INTERVALS);

// ---- scale, orientation, descriptors, ordering (sift.cpp:1065-1643) --------------------------------------
void CalculateScale(std::vector<Keypoint> &features, double sigma = SIGMA, int intervals = INTERVALS);
void HalfFeatures(std::vector<Keypoint> &features);
void OrientationAssignment(std::vector<Keypoint> &extrema, std::vector<Keypoint> &features,
                           const std::vector<Mat> &gauss_pyr);
void DescriptorRepresentation(std::vector<Keypoint> &features, const std::vector<Mat> &gauss_pyr, int bins, int width);
bool FeatureCmp(Keypoint &f1, Keypoint &f2);

// ---- drivers (sift.h:139-141) ----------------------------------------------------------------------------
void Sift(const Mat &src, std::vector<Keypoint> &features, double sigma = SIGMA, int intervals = INTERVALS);
void Sift_nointerval(const Mat &src, std::vector<Keypoint> &features, double sigma = SIGMA, int intervals = INTERVALS);

// ---- text feature files (sift.h:155-157) -------------------------------------------------------------------
void read_features(std::vector<Keypoint> &features, const char *file);
void write_features(const std::vector<Keypoint> &features, const char *file);

// ---- matching: the loop of the reference's demo (sift/源.cpp:88-156) as a function --------------------------
// For every feature of features1 the nearest and second nearest feature of features2 by Euclidean distance over
// the descriptor; a MatchPoint is appended when nearest / second nearest < ratio (0.5 in the demo).  Distances
// are formed per pair (the demo's distance variable carries over between candidates, 源.cpp:97-110; that is not
// reproduced).  Output in index_1 order.
void MatchFeatures(const std::vector<Keypoint> &features1, const std::vector<Keypoint> &features2,
                   std::vector<MatchPoint> &matches, double ratio = 0.5);

// ---- drawing and pyramid dumps (sift.h:143-153, sift.cpp:1772-1884); host code -----------------------------------
// Drawing works on CV_8UC3 images.  Without OpenCV the lines are 8-connected Bresenham lines and the circles midpoint
// circles of thickness 1, clipped to the image (OpenCV's own rasteriser may choose other pixels on a tie), and
// writecv64f / write_pyr write binary PGM files (`<dir><i>.pgm`) where the reference writes JPEG through cv::imwrite.
void DrawSiftFeature(Mat &src, Keypoint &feat, cv::Scalar color); // arrow of length 5*scale along ori (sift.cpp:1772-1806)
void DrawSiftFeatures(Mat &src, std::vector<Keypoint> &features); // green arrows (sift.cpp:1808-1815)
void DrawKeyPoints(Mat &src, std::vector<Keypoint> &keypoints);   // blue circles of radius 3 at (dx, dy) (sift.cpp:1817-1827)
void DrawKeyPointsRight(Mat &src, std::vector<Keypoint> &keypoints, cv::Size size); // ... shifted right by size.width
void cv64f_to_cv8U(const Mat &src, Mat &dst);          // min-max normalised to 0..255, rounded (sift.cpp:1843-1847)
void writecv64f(const char *filename, const Mat &mat); // cv64f_to_cv8U, then the image file (sift.cpp:1858-1863)
void write_pyr(const std::vector<Mat> &pyr, const char *dir); // every level as <dir><i>.pgm (sift.cpp:1865-1873)
const char *GetFileName(const char *dir, int i);               // "<dir><i>.jpg" (sift.h:149, sift.cpp:1849-1855); per-thread buffer
void display_pyr(const std::vector<Mat> &pyr, std::string dir); // prints the dump names (sift.h:153, sift.cpp:1875-1884)

// ---- off-path helpers of the reference, kept so that its callers link (host CPU code) ---------------------------
void GaussianTemplateSmooth(const Mat &src, Mat &dst);          // fixed 7x7 template on 8-bit images (sift.h:121, sift.cpp:183-236)
void GaussianSmooth2D(const Mat &src, Mat &dst, double sigma);  // direct 2-D Gaussian on 8-bit gray (sift.h:123, sift.cpp:245-314)
void testInverse3D();                                           // prints H * inverse(H) for one 3x3 (sift.h:159, sift.cpp:1934-1960)

// ---- additions of this implementation -----------------------------------------------------------------------
int SiftLastStatus();            // sg_status of the most recent call on this thread (0 = ok)
const char *SiftLastError();     // text for it
void SiftSetDevice(int device);  // HIP device used by later calls on this thread (default 0)
void SiftSetVerbose(bool on);    // print the reference's progress lines (sift.cpp:1654-1706); default off
void SiftReleaseContext();       // free this thread's GPU context
// Sift()/Sift_nointerval() calls of all threads share one context per device that runs up to max_batch images per
// batch (whatever is waiting when the context falls idle; nobody waits for a batch to fill).  0 or 1: every thread
// drives a context of its own (the default; the environment variable SIFT_GPU_BATCH sets another default).  There are
// two such contexts per device, taken in turn (SIFT_GPU_LANES = 1..4 sets another number), so that one batch's
// transfers overlap the other's kernels; a context whose batch size does not fit the device is created smaller.
// Call it while no Sift() call is running: it frees the shared contexts.
void SiftSetBatching(int max_batch);
// Arithmetic of the Gaussian pyramid in later Sift()/Sift_nointerval() calls of every thread (the reference has one
// arithmetic: double, one rounding per operation).  SIFT_EXACT (the default): exactly that, results bit-identical to
// the reference's algorithm.  SIFT_FUSED: double, with fused multiply-adds in the Gaussian filters -- one rounding per
// filter tap instead of two, values move by 1e-16 relative, a frame's features come out identical or nearly so (on
// 1080p test frames: all of 178 000 features identical; on small random images 99.7 % of the features, 98.8 % of the
// descriptors) and extraction runs 5 % faster.  SIFT_SINGLE: a single-precision pyramid (features agree with the exact
// ones to the extent single precision allows; fastest).  The environment variable SIFT_GPU_PRECISION = exact | fused |
// single sets another default.  Call it while no Sift() call is running: contexts are rebuilt at their next use.
enum SiftPrecision { SIFT_EXACT = 0, SIFT_SINGLE = 1, SIFT_FUSED = 2 };
void SiftSetPrecision(int precision);

// ---- a batch of frames over the GPUs of one node ---------------------------------------------------------------------
// The reference extracts one image per call (sift/源.cpp:21-22 calls Sift() once per image) and a frame's features
// depend on its pixels alone, so a batch shards by frame: GPU g of G takes frames [g * ceil(F/G), (g+1) * ceil(F/G)),
// one host thread per GPU, no communication during extraction; then the GPUs all-gather their packed feature sets
// (RCCL all-gather-v over xGMI: afterwards every GPU holds every frame's features, in frame order) and the host
// receives them from GPU 0, ordered per frame as Sift() orders them, in the packed form of the throughput path: dx,
// dy, scale, ori as floats and the 128 descriptor values as bytes.
struct SiftPackedFeature {
    float x, y, scale, ori;        // Keypoint.dx, .dy, .scale, .ori
    unsigned char descriptor[128]; // Keypoint.descriptor (integers 0..255)
};
struct SiftBatchFeatures {
    std::vector<SiftPackedFeature> features; // all frames' features back to back, frame after frame
    std::vector<size_t> first;               // frame f's features: features[first[f]] .. features[first[f + 1] - 1]
    size_t count(size_t f) const { return first[f + 1] - first[f]; }
    const SiftPackedFeature *frame(size_t f) const { return features.data() + first[f]; }
};
struct SiftMultiGpuOptions {
    int n_gpus = 0;                 // 0: every GPU of the node
    int batch = 32;                 // frames a GPU extracts per launch sequence (a 1080p frame takes 0.7 GB of pyramid)
    double sigma = SIGMA;
    bool nointerval = false;        // Sift_nointerval's detector
    int precision = 0;              // SIFT_EXACT, SIFT_FUSED or SIFT_SINGLE (see SiftSetPrecision)
    int max_features_per_frame = 0; // 0: width * height / 40 + 4096; raised (and the shards run again) on overflow
    bool rehearsal = false;         // ranks share the GPUs present (rank r on GPU r mod G) and their sets meet in host
                                    // memory instead of RCCL, which refuses two ranks on one GPU: for trying the
                                    // sharding on a machine with fewer GPUs than ranks
};
// Returns 0 (SG_OK) or a negative sg_status; *error (optional) receives the text.  One call runs at a time (others
// wait); the GPUs' contexts, buffers and RCCL communicators are kept for the next call of the same shape until
// SiftMultiGpuRelease() frees them.
int SiftBatchMultiGpu(const std::vector<Mat> &frames, SiftBatchFeatures &result,
                      const SiftMultiGpuOptions &options = SiftMultiGpuOptions(), std::string *error = nullptr);
void SiftMultiGpuRelease();
