// kernels_describe.hip -- orientation assignment, 4x4x8 descriptors, scale ordering and output packing
// for gfx950.  All histogram arithmetic is double precision whatever the pyramid's storage type.
#include "sg_internal.h"
#include "sg_dd.h"
#include "sg_math.h"

#include <stddef.h>
#include <stdlib.h>
#include <type_traits>

namespace sg {

#define SG_PI 3.1415926535897932384626433832795 /* CV_PI */

// CalcGradMagOri (reference sift/sift.cpp:1109-1136): differences without the 1/2 factor.
template <typename T>
__device__ __forceinline__ bool grad_at(const T *g, int rows, int cols, int pitch, int x, int y, double &dx, double &dy)
{
    if (x > 0 && x < cols - 1 && y > 0 && y < rows - 1) {
        const T *p = g + (size_t)y * pitch + x;
        dx = (double)p[1] - (double)p[-1];
        dy = (double)p[pitch] - (double)p[-pitch];
        return true;
    }
    return false;
}

__device__ __forceinline__ void lds_add(double *p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Histogram add in 64-bit fixed point.  The LDS pipe of a compute unit takes a 64-lane ds_add_u64 in half the time
// of a ds_add_f64, and the histogram adds are a sizeable part of what the descriptor kernel makes a SIMD issue.  A bin holds an
// unsigned count of units of 2^-50.  The two factors of a contribution carry 2^-512 each on top of that unit, so their
// product a*b (below 4 in real terms) falls into the subnormal range, where a double's bit pattern IS its value in
// units of 2^-1074: one multiply rounds the contribution to a whole unit (to nearest, ties to even) and leaves the
// integer as the result's bit pattern, with nothing to mask off.  (Double-precision subnormals run at full speed and
// are never flushed on this hardware.)  Rounding to 2^-50 is finer than the rounding of a double sum of the same
// size, and integer adds commute, so a bin does not depend on the order in which its adds arrive.
constexpr double kFixScale = 1125899906842624.0; // 2^50: units per 1.0
// a*b = x * 2^-1024 puts x * 2^50 units into the bit pattern (2^50 * 2^-1074).  The magnitude shares carry 2^-462 (2^-231
// from each of the two factors of the Gaussian weight), the orientation fractions 2^-562: both stay normal numbers.
constexpr double kFixWeightHalf = 0x1p-231;
constexpr double kFixWeight = 0x1p-462;
constexpr double kFixFrac = 0x1p-562;
static_assert(kFixWeightHalf * kFixWeightHalf == kFixWeight && kFixWeight * kFixFrac == 0x1p-1024, "the scales meet at 2^-1024");

// LDS addresses as 32-bit integers (what the ds_ instructions take), for address arithmetic the compiler is not to
// rearrange: a * b + c with 24-bit factors is one full-rate instruction.
__device__ __forceinline__ unsigned int lds_address(void *p)
{
    return (unsigned int)(size_t)(__attribute__((address_space(3))) void *)p;
}
__device__ __forceinline__ unsigned int mad_u24(unsigned int a, unsigned int b_uniform, unsigned int c)
{
    unsigned int d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c));
    return d;
}
// Two histogram adds (the two orientation bins of one cell) for the lanes of `mask`, which is a wave mask in scalar
// registers, computed well before: the execution mask is set from it around the two adds and put back.  Setting the
// execution mask from scalar registers costs nothing, whereas a vector compare whose result goes straight into the
// execution mask, with vector work behind it -- what an `if` around the adds compiles to -- holds up the whole SIMD,
// other waves included, for about the time of three double-precision instructions (measured: 6.7 ns against
// 2.2 ns per v_fma_f64 at five waves per SIMD).  The products are formed for every lane.
template <int OFFSET>
__device__ __forceinline__ void lds_add_fixed_pair_masked(unsigned int address, double a, double b0, double b1,
                                                          unsigned long long mask)
{
    const double p0 = a * b0, p1 = a * b1; // bit patterns = whole units (see above)
#ifdef SG_X_NOATOM // timing experiment: the products are formed, the adds are not issued
    asm volatile("" ::"v"(p0), "v"(p1), "v"(address), "s"(mask));
#else
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %1\n\t"
#ifdef SG_X_ADD32 // timing experiment: 32-bit adds
                 "ds_add_u32 %2, %3 offset:%5\n\t"
                 "ds_add_u32 %2, %4 offset:%6\n\t"
#else
                 "ds_add_u64 %2, %3 offset:%5\n\t"
                 "ds_add_u64 %2, %4 offset:%6\n\t"
#endif
                 "s_mov_b64 exec, %0"
                 : "=&s"(saved)
                 : "s"(mask), "v"(address), "v"(p0), "v"(p1), "n"(OFFSET), "n"(OFFSET + 8)
                 : "memory");
#endif
}

// A double moved between lanes as a data-parallel-primitive operand (two 32-bit moves); a lane without a source takes 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_zero_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

// ---- orientation ---------------------------------------------------------------------------------------
// One wave (a 64-thread workgroup) per keypoint.  CalculateOrientationHistogram (sift.cpp:1149-1199): the
// lanes walk the (2r+1)^2 window and add mag*weight into 36-bin LDS histograms with double atomics.  The
// wave keeps four copies (lane & 3, at a stride that puts them in different banks) so that neighbouring
// lanes, whose gradients point the same way, do not serialise on one bin; copies are added in a fixed order,
// and one wave's adds to one copy resolve in a fixed order, so sums are reproducible run to run.  The four
// neighbour pixels of a lane's next sample are fetched while the current one is processed.  The Gaussian
// weight exp(-(i^2+j^2)/(2 s^2)) is taken as E[|i|]*E[|j|] from a per-keypoint table of
// E[k] = exp(-k^2/(2 s^2)).  Two passes of the circular [1/4 1/2 1/4] filter (sift.cpp:1209-1221), the
// maximum (sift.cpp:1231-1240) and the peak test with parabolic refinement (sift.cpp:1280-1307) run on
// lanes 0..35.  BIG (chosen on the host from the largest radius the parameters allow) adds the path for
// windows of radius 64 and more, whose weights do not fit the table.
// Kept out of line: it runs for a handful of samples per frame and must not cost the window walk any registers.
__device__ __noinline__ double atan2_correctly_rounded(double y, double x) { return sg_atan2_cr(y, x); }

constexpr int kOriCopies = 4;
constexpr int kOriCopyStride = 66; // doubles; 64 + 2, so that the copies start 4 banks apart
constexpr int kOriTab = 64;

// Same-wave LDS hand-off: LDS operations of one wave complete in order, so waiting for the wave's own
// outstanding LDS operations makes earlier stores and atomics visible to its later loads (other lanes included).
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

#ifndef SG_ORI_WPE
#define SG_ORI_WPE 5
#endif
template <typename T, bool BIG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SG_ORI_WPE, SG_ORI_WPE))) void
orient_k(OctaveTab tab, const KpRec *__restrict__ kp, size_t kp_img, const unsigned int *__restrict__ kp_src,
         size_t src_img, const unsigned int *__restrict__ kp_count, double *__restrict__ ori_tmp, size_t ori_img,
         unsigned long long *__restrict__ peak_words, size_t pw_img, unsigned int cap,
         unsigned long long *__restrict__ trip_slots)
{
    __shared__ double hist[kOriCopies * kOriCopyStride];
    __shared__ double E[kOriTab];
    __shared__ double2 centres[5];
    const int lane = threadIdx.x;
    double *hs = hist;
    double *hc = hist + (lane & (kOriCopies - 1)) * kOriCopyStride;
    const int b = blockIdx.y;
    const unsigned int n = min(kp_count[b], cap);
    const double PI2 = 2.0 * SG_PI;
    sg_atan_centres_fill(centres, lane, true);

    // The slot index two keypoints ahead and the record one keypoint ahead are loaded while the current
    // keypoint is processed.
    const unsigned int jstep = gridDim.x;
    const unsigned int j_first = blockIdx.x;
    const unsigned int *ks = kp_src + (size_t)b * src_img;
    const KpRec *kpb = kp + (size_t)b * kp_img;
    KpRec rec_next;
    unsigned int slot_next2 = 0;
    if (j_first < n)
        rec_next = kpb[ks[j_first]];
    if (j_first + jstep < n)
        slot_next2 = ks[j_first + jstep];
    unsigned int trips = 0; // 64-sample trips of the window walk, for pricing the kernel against the issue rate
    for (unsigned int j = j_first; j < n; j += jstep) {
        const KpRec r = rec_next;
        if (j + jstep < n)
            rec_next = kpb[slot_next2];
        if (j + 2 * jstep < n)
            slot_next2 = ks[j + 2 * jstep];
        const int radius = (int)rint(3.0 * 1.5 * r.oscale); // ORI_WINDOW_RADIUS * octave_scale
        const double sig = 1.5 * r.oscale;                   // ORI_SIGMA_TIMES * octave_scale
        const double econs = -1.0 / (2.0 * sig * sig);
#pragma unroll
        for (int q = 0; q < kOriCopies; q++)
            hist[q * kOriCopyStride + lane] = 0.0;
        E[lane] = exp((lane * lane) * econs);
        wave_lds_fence();
        {
            const int o = r.octave;
            const int rows = tab.rows[o], cols = tab.cols[o], pitch = tab.pitch[o];
            const char *gb = reinterpret_cast<const char *>((const T *)tab.lv[o][r.interval] + (size_t)b * tab.img_stride[o]);
            const unsigned int pitch_b = (unsigned int)pitch * (unsigned int)sizeof(T);
            const int W = 2 * radius + 1, total = W * W;
            trips += (unsigned int)((total + 63) >> 6);
            const bool use_tab = !BIG || radius < kOriTab;
            // sample s -> (jj = s / W - radius: y offset; i = s % W - radius: x offset), so that neighbouring lanes
            // read neighbouring pixels (the reference walks x in its outer loop; the sums do not depend on it
            // beyond rounding)
            // A window that lies inside the plane with a pixel to spare on every side (nearly all do) has a gradient
            // at every sample: its walk needs neither the validity test nor the redirection of invalid samples.
            const bool interior = r.x - radius > 0 && r.x + radius < cols - 1 && r.y - radius > 0 && r.y + radius < rows - 1;
            auto walk = [&](auto interior_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
                int qj = lane / W, qi = lane - qj * W;
                const int dq = 64 / W, dr = 64 - dq * W;
                int i_n = 0, j_n = 0;
                bool ok_n = false;
                T nl = 0, nr = 0, nu = 0, nd = 0;
                auto fetch = [&]() {
                    i_n = qi - radius;
                    j_n = qj - radius;
                    const int x = r.x + i_n, y = r.y + j_n;
                    int xx = x, yy = y;
                    if (!INTERIOR) {
                        ok_n = x > 0 && x < cols - 1 && y > 0 && y < rows - 1; // CalcGradMagOri's validity (sift.cpp:1115)
                        // the loads of an invalid sample are redirected to the nearest pixel that has four neighbours
                        xx = min(max(x, 1), cols - 2);
                        yy = min(max(y, 1), rows - 2);
                    }
                    const unsigned int off = (unsigned int)yy * pitch_b + (unsigned int)xx * (unsigned int)sizeof(T);
                    const char *p8 = gb + (size_t)off;
                    nl = *reinterpret_cast<const T *>(p8 - sizeof(T));
                    nr = *reinterpret_cast<const T *>(p8 + sizeof(T));
                    nu = *reinterpret_cast<const T *>(p8 - (size_t)pitch_b);
                    nd = *reinterpret_cast<const T *>(p8 + (size_t)pitch_b);
                    qj += dq;
                    qi += dr;
                    if (qi >= W) {
                        qi -= W;
                        qj++;
                    }
                };
                if (lane < total)
                    fetch();
                for (int s = lane; s < total; s += 64) {
                    const int i = i_n, jj = j_n;
                    const bool ok = INTERIOR || ok_n;
                    const double dx = (double)nr - (double)nl;
                    const double dy = (double)nd - (double)nu;
                    if (s + 64 < total)
                        fetch();
                    if (ok) {
                        double mag = sg_sqrt(dx * dx + dy * dy);
                        double ori = sg_atan2(dy, dx, centres);
                        double weight;
                        if (use_tab)
                            weight = E[abs(i)] * E[abs(jj)];
                        else
                            weight = exp((i * i + jj * jj) * econs);
                        // bin = cvRound(36 (pi - ori) / (2 pi)) (sift.cpp:1190).  The quotient is first taken as a
                        // product with 1/(2 pi), a few ulp from the division; the two can round to different integers
                        // only next to a half-integer, and there the division itself is evaluated.
                        // (The 1-ulp square root and arctangent stay here although the sums would tolerate less: on a
                        // mirror-symmetric image neighbouring histogram bins tie exactly, and which side a peak's
                        // parabola offset falls to then hangs on the last bits.)
                        const double num = kOriBins * (SG_PI - ori);
                        double quo = num * (1.0 / PI2);
                        const bool near_half = fabs(fabs(quo - rint(quo)) - 0.5) < 1e-9;
                        if (__ballot(near_half)) { // wave-uniform, so that this is skipped and not merely masked
                            // Next to a bin boundary (gradients along a diagonal, which symmetric structure produces
                            // to the last bit or an ulp off it) the bin hangs on the last bit of atan2: there the angle
                            // is evaluated correctly rounded, as the C library does it but for one argument in two
                            // thousand, and the quotient by the reference's own division.
                            if (near_half)
                                quo = (kOriBins * (SG_PI - atan2_correctly_rounded(dy, dx))) / PI2;
                        }
                        int bin = (int)rint(quo);
                        bin = bin < kOriBins ? bin : 0;
                        lds_add(&hc[bin], mag * weight);
                    }
                }
            };
            if (interior)
                walk(std::true_type{});
            else
                walk(std::false_type{});
        }
        wave_lds_fence();
        // The 36 bins sit in lanes 0..35 from here on.  A bin's neighbours around the ring come from the lanes beside it
        // by whole-wave shifts (data-parallel-primitive moves), the two ends from each other through scalar registers: the
        // two smoothing passes, the maximum and the peak test then need no LDS round trips (they took a dozen).
        const bool binlane = lane < kOriBins;
        double c = 0.0;
        if (binlane)
            c = ((hs[lane] + hs[kOriCopyStride + lane]) + hs[2 * kOriCopyStride + lane]) + hs[3 * kOriCopyStride + lane];
        auto lane_value = [&](double v, int l) {
            return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
        };
        auto ring_prev = [&](double v) { // the value of the bin below (bin 35 for bin 0)
            const double below = dpp_or_zero_f64<0x138, 0xf>(v); // wave_shr:1
            const double wrap = lane_value(v, kOriBins - 1);
            return lane == 0 ? wrap : below;
        };
        auto ring_next = [&](double v) { // the value of the bin above (bin 0 for bin 35)
            const double above = dpp_or_zero_f64<0x130, 0xf>(v); // wave_shl:1
            const double wrap = lane_value(v, 0);
            return lane == kOriBins - 1 ? wrap : above;
        };
#pragma unroll
        for (int pass = 0; pass < 2; pass++) { // ORI_SMOOTH_TIMES
            const double prev = ring_prev(c), next = ring_next(c);
            c = 0.25 * prev + 0.5 * c + 0.25 * next;
        }
        // the largest bin: a running maximum towards lane 63 (0 stands in where a lane has no source: bins are >= 0)
        double mx = binlane ? c : 0.0;
        {
            double t;
            t = dpp_or_zero_f64<0x111, 0xf>(mx); mx = t > mx ? t : mx; // row_shr:1
            t = dpp_or_zero_f64<0x112, 0xf>(mx); mx = t > mx ? t : mx;
            t = dpp_or_zero_f64<0x114, 0xf>(mx); mx = t > mx ? t : mx;
            t = dpp_or_zero_f64<0x118, 0xf>(mx); mx = t > mx ? t : mx;
            t = dpp_or_zero_f64<0x142, 0xa>(mx); mx = t > mx ? t : mx; // row_bcast:15
            t = dpp_or_zero_f64<0x143, 0xc>(mx); mx = t > mx ? t : mx; // row_bcast:31
            mx = lane_value(mx, 63);
        }
        bool peak = false;
        {
            const double l = ring_prev(c), rr = ring_next(c);
            const double thr = mx * 0.8; // ORI_PEAK_RATIO
            if (binlane && c > l && c > rr && c >= thr) {
                peak = true;
                double bin = lane + (0.5 * (l - rr) / (l - 2.0 * c + rr));
                bin = (bin < 0) ? (bin + kOriBins) : (bin >= kOriBins ? (bin - kOriBins) : bin);
                ori_tmp[(size_t)b * ori_img + (size_t)j * kOriBins + lane] = ((PI2 * bin) / kOriBins) - SG_PI;
            }
        }
        // (the peaks are counted per chunk of keypoints by a launch of their own, launch_chunk_popc: counting here
        // would be one atomic add per keypoint on a word that a thousand keypoints share)
        unsigned long long m = __ballot(peak);
        if (lane == 0)
            peak_words[(size_t)b * pw_img + j] = m;
        wave_lds_fence(); // the histograms are cleared for the next keypoint
    }
    if (lane == 0 && trips)
        atomicAdd(&trip_slots[(size_t)b * kTripWordsPerImage + (size_t)(kTripSlots + blockIdx.x % kTripSlots) * kTripStride],
                  (unsigned long long)trips);
}

// One-wave workgroups per image for the kernels that walk a list of keypoints or features with a grid-stride loop.
// Every workgroup pays a start-up (two dependent loads before the first record is there, tables, constants) and only
// a wave that takes several entries can load the next record while it works on the current one, so a launch holds a
// fixed number of waves in all -- waves_per_cu for every compute unit, a few rounds of the resident waves, enough to
// even out the differences between windows -- and at least min_per_image per image, whatever the list lengths.  (With one wave per entry the orientation
// kernel ran 30 % longer.)  SIFTGPU_ORI_WAVES / SIFTGPU_DESCR_WAVES (read once per process) override the total, for experiments.
static unsigned int feature_blocks(unsigned int cap, int B, unsigned int waves_per_cu, unsigned int min_per_image,
                                   int total_override)
{
    unsigned int total = (unsigned int)env().n_cu * waves_per_cu;
    if (total_override > 0)
        total = (unsigned int)total_override;
    unsigned int blocks = total / (unsigned int)(B > 0 ? B : 1);
    if (blocks < min_per_image && total_override <= 0) // large batches: a wave still takes no more than a few dozen entries of an image
        blocks = min_per_image;
    if (blocks == 0)
        blocks = 1;
    return blocks < cap ? blocks : cap;
}

void launch_orient(const OctaveTab &tab, const KpRec *kp, size_t kp_img_stride, const unsigned int *kp_src,
                   size_t src_img_stride, const unsigned int *kp_count, double *ori_tmp, size_t ori_img_stride,
                   unsigned long long *peak_words, size_t pw_img_stride, unsigned int *chunk_sums,
                   size_t chunks_img_stride, int n_chunks, unsigned int cap, int max_radius, int B, int prec,
                   unsigned long long *trip_slots, hipStream_t s)
{
    // Short lists (a frame alone, a small batch): one resident generation of waves (20 per compute unit) -- a keypoint
    // list that fills the chip only a few times over is finished sooner by waves that stay and take several keypoints
    // each than by waves that come and go; longer lists gain from more waves, which even out the differences between
    // windows.  The list lengths are on the device; the capacity (about three times a natural image's count) stands in.
    // (Measured per batch size at 1080p and for an 8K frame alone.)
    const unsigned long long room = (unsigned long long)cap * (unsigned int)(B > 0 ? B : 1);
    unsigned int blocks = feature_blocks(cap, B, room <= 500000ull ? 20 : (room <= 1000000ull ? 40 : 64), 256,
                                         env().ori_waves);
    if (blocks == 0)
        return;
    dim3 g(blocks, B);
    const bool big = max_radius >= kOriTab;
#define SG_ORI_LAUNCH(TT, BB)                                                                                         \
    orient_k<TT, BB><<<g, 64, 0, s>>>(tab, kp, kp_img_stride, kp_src, src_img_stride, kp_count, ori_tmp, ori_img_stride, \
                                      peak_words, pw_img_stride, cap, trip_slots)
    if (prec == SG_F64) {
        if (big)
            SG_ORI_LAUNCH(double, true);
        else
            SG_ORI_LAUNCH(double, false);
    } else {
        if (big)
            SG_ORI_LAUNCH(float, true);
        else
            SG_ORI_LAUNCH(float, false);
    }
#undef SG_ORI_LAUNCH
    // peaks per chunk of keypoints (one word per keypoint; words beyond the list are stale and not counted)
    launch_chunk_popc(peak_words, pw_img_stride, (unsigned int)pw_img_stride, kp_count, 1, chunk_sums, chunks_img_stride,
                      n_chunks, B, s);
}

// ---- descriptors -----------------------------------------------------------------------------------------
// One wave (a 64-thread workgroup) per feature, nothing shared between waves and no barriers.
// CalculateDescrHist + InterpHistEntry (sift.cpp:1352-1535).  The samples that can fall into the rotated
// 4x4 grid form, in every window row, one interval of columns; the lanes compute those intervals (one row per
// lane, a superset by a 1e-6 px margin, clipped to the pixels that have a gradient), the wave then writes the
// intervals' (row, column) pairs into a queue, and takes the queued samples 64 at a time: the reference's
// acceptance test in double, gradient, and the trilinear spread into 4x4x8 histograms with LDS atomics in 64-bit
// fixed point (lds_add_fixed_pair_masked; six copies of the histograms, lane % 6, at a stride that puts them in different banks).
// So the expensive part runs with full lanes and the window's corners outside the grid are never visited.
// Against the reference's per-sample formulas three things are algebraically regrouped (results agree to
// ~1e-15 relative): the rotated coordinates use cos/shw and sin/shw in fused multiply-adds, the window weight
// exp(-(rx^2+ry^2)/8), with rx^2+ry^2 = (i^2+j^2)/shw^2, is E[|i|]*E[|j|] from a per-feature table
// E[k] = exp(-k^2/(8 shw^2)), and the orientation coordinate is formed directly in bin units and not wrapped
// into [0, 8) (the bin index wraps).  HistToDescriptor (sift.cpp:1567-1597): normalise, clamp at 0.2, normalise, min(255,(int)(512 v)).

constexpr int kDescQueue = 768;  // queue entries (samples) per round: a multiple of 128 (full double trips)
constexpr int kDescQueuePad = 128; // entries behind the list that the look-ahead of the last two trips reads (copies of the last entry)
// Histogram copies per wave, and waves per SIMD.  Adds of one instruction that meet in one bin are carried out one
// after the other, and neighbouring lanes -- neighbouring pixels -- meet often: with one copy the kernel takes 81 ms
// per 256 frames, with two 54, with four 46.  Four waves per SIMD run as fast as five (the loop is bound by the
// instructions a SIMD can issue, not by latencies) and leave each wave 10 KB of LDS: room for six copies, 2 % faster.
#ifndef SG_DESCR_COPIES
#define SG_DESCR_COPIES 6
#endif
#ifndef SG_DESCR_WPE
#define SG_DESCR_WPE 4
#endif
constexpr int kDescCopies = SG_DESCR_COPIES;
constexpr int kDescCellBins = 9; // a cell's eight orientation bins and a ninth for the upper share of bin 7 (added to bin 0 at the end)
constexpr int kDescCopyStride = 16 * kDescCellBins + 2; // entries; + 2, so that the copies start 4 banks apart
constexpr int kDescTab = 256;      // weight table entries of the wide-window variant (one per window row/column)
constexpr int kDescTabSmall = 128; // ... of the common variant (every radius below 64)
constexpr int kDescPackRadius = 127; // largest radius whose (row, column) pair fits the 16-bit queue entry

// Inclusive sum over the 64 lanes with data-parallel-primitive moves (no LDS): shifts by 1, 2, 4, 8 inside the rows of
// 16 lanes, then lane 15 of rows 0 and 2 into rows 1 and 3, then lane 31 into the upper half.  A lane with no source
// takes 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned int dpp_or_zero(unsigned int v)
{
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ unsigned int wave_scan_add(unsigned int v)
{
    v += dpp_or_zero<0x111, 0xf>(v); // row_shr:1
    v += dpp_or_zero<0x112, 0xf>(v);
    v += dpp_or_zero<0x114, 0xf>(v);
    v += dpp_or_zero<0x118, 0xf>(v);
    v += dpp_or_zero<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3
    return v;
}

// Sum over the 64 lanes, the same value in every lane: partial sums move between lanes as data-parallel-primitive
// operands (an inclusive scan as in wave_scan_add: lane 63 ends with the total), which lane 63 then hands to the wave
// through scalar registers.  No LDS round trips (six dependent ones per sum with shuffles).
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_or_zero_f64<0x111, 0xf>(v); // row_shr:1
    v += dpp_or_zero_f64<0x112, 0xf>(v);
    v += dpp_or_zero_f64<0x114, 0xf>(v);
    v += dpp_or_zero_f64<0x118, 0xf>(v);
    v += dpp_or_zero_f64<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero_f64<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Columns [lo, lo+cnt) (as offsets jj from the feature) of window row i that can pass the acceptance test
// |c*jj - s*i| < T and |s*jj + c*i| < T, T = 2.5*shw + margin, and whose pixel has a gradient.
__device__ __forceinline__ void descr_row_span(int i, double c, double s, double inv_c, double inv_s, bool c_ok,
                                               bool s_ok, double T, int radius, int x0, int y0, int rows, int cols,
                                               int &lo, int &cnt)
{
    double lo_d = -(double)radius, hi_d = (double)radius;
    const double b1 = -s * i, b2 = c * i;
    bool empty = false;
    if (c_ok) {
        const double p = (-T - b1) * inv_c, q = (T - b1) * inv_c;
        lo_d = fmax(lo_d, fmin(p, q));
        hi_d = fmin(hi_d, fmax(p, q));
    } else {
        empty = !(fabs(b1) < T);
    }
    if (s_ok) {
        const double p = (-T - b2) * inv_s, q = (T - b2) * inv_s;
        lo_d = fmax(lo_d, fmin(p, q));
        hi_d = fmin(hi_d, fmax(p, q));
    } else {
        empty = empty || !(fabs(b2) < T);
    }
    const int y = y0 + i;
    empty = empty || !(y > 0 && y < rows - 1) || !(lo_d <= hi_d);
    int l = (int)ceil(lo_d), h = (int)floor(hi_d);
    l = max(l, 1 - x0);
    h = min(h, cols - 2 - x0);
    lo = l;
    cnt = empty ? 0 : max(h - l + 1, 0);
}

// What the wave keeps in LDS.  The histograms come last: bin addresses are formed from cell indices (r0, c0) >= -1, five
// cells below the first bin at the lowest, and the arrays in front keep those addresses above zero.
template <bool BIG>
struct DescrLds {
    double2 lut[kSinLutEntries];                 // direction table (sg_angle_sin_lut)
    double etab[BIG ? kDescTab : kDescTabSmall]; // etab[k] = E[|k - radius|], k = row or column in the window
    unsigned short queue[kDescQueue + (BIG ? 2 : 1) * kDescQueuePad]; // (the wide-window variant: 256, for the overrun of its longest row)
    unsigned long long hist[kDescCopies * kDescCopyStride]; // units of 2^-50 (lds_add_fixed_pair_masked)
};
static_assert(offsetof(DescrLds<false>, hist) % 16 == 0 && offsetof(DescrLds<false>, hist) >= 5 * kDescCellBins * 8,
              "16-byte stores clear the histograms; addresses stay positive");
static_assert(4 * SG_DESCR_WPE * sizeof(DescrLds<false>) <= 160 * 1024, "the waves of a compute unit share 160 KB");

template <typename T, bool BIG>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(BIG ? 4 : SG_DESCR_WPE, BIG ? 4 : SG_DESCR_WPE))) void // (the wide tables leave room for fewer)
descr_k(OctaveTab tab, const FeatRec *__restrict__ frec, size_t frec_img, const unsigned int *__restrict__ feat_count,
        uint8_t *__restrict__ desc, size_t desc_img, double *__restrict__ prequant, size_t pq_img, unsigned int cap,
        unsigned long long *__restrict__ trip_slots)
{
    __shared__ __align__(16) DescrLds<BIG> lds;
    const int lane = threadIdx.x;
    unsigned long long *hist = lds.hist;
    double *etab = lds.etab;
    unsigned short *queue = lds.queue;
    const int b = blockIdx.y;
    const unsigned int n = min(feat_count[b], cap);
    const FeatRec *fr = frec + (size_t)b * frec_img;
    sg_sin_lut_fill(lds.lut, lane);

    // the next feature's record is loaded while the current one is processed
    FeatRec nxt;
    if (blockIdx.x < n)
        nxt = fr[blockIdx.x];
    unsigned int trips = 0; // 64-sample trips of the drain loop, for pricing the kernel against the issue rate
    for (unsigned int f = blockIdx.x; f < n; f += gridDim.x) {
        const FeatRec r = nxt;
        if (f + gridDim.x < n)
            nxt = fr[f + gridDim.x];
        const double ori = r.ori;
        const int o = r.octave;
        const int rows = tab.rows[o], cols = tab.cols[o], pitch = tab.pitch[o];
        const double cos_ori = r.cos_ori, sin_ori = r.sin_ori; // cos(ori), sin(ori) (sift.cpp:1474-1475)
        const double shw = 3 * r.oscale;                       // DESCR_SCALE_ADJUST * octave_scale
        const double inv_shw = 1.0 / shw;
        const int radius = (int)((shw * sqrt(2.0) * (4 + 1)) / 2.0); // truncating cast (sift.cpp:1488)
        const int W = 2 * radius + 1;
        const double wcons = -0.125 * inv_shw * inv_shw; // -1/(2*(0.5*4)^2) / shw^2
        // BIG is chosen on the host from the largest radius the parameters allow (64 or more): only then can a
        // window be too wide for the 16-bit queue entries and the 128-entry weight table -- and for planes whose rows
        // are 16 MB and longer: the queued path forms sample offsets with a 24-bit multiply.
        const unsigned int pitch_b = (unsigned int)pitch * (unsigned int)sizeof(T);
        const bool packed = !BIG || (radius <= kDescPackRadius && pitch_b < (1u << 24));
        // Sample addresses are offsets from the window's first sample (which may lie outside the image: only samples
        // with four neighbours inside it are listed).
        const T *g0 = (const T *)tab.lv[o][r.interval] + (size_t)b * tab.img_stride[o] +
                      ((ptrdiff_t)(r.y - radius) * pitch + (r.x - radius));
        const char *gb = reinterpret_cast<const char *>(g0);
        const char *gb_up = reinterpret_cast<const char *>(g0 - pitch), *gb_dn = reinterpret_cast<const char *>(g0 + pitch);
        const double obin_c = (SG_PI - ori) * (4.0 / SG_PI) + 8.0; // + 8: see the drain loop

        {
            double2 zero2;
            zero2.x = 0.0;
            zero2.y = 0.0;
            double2 *h2 = reinterpret_cast<double2 *>(hist);
#pragma unroll
            for (int q = 0; q < (kDescCopies * kDescCopyStride / 2 + 63) / 64; q++)
                if (lane + 64 * q < kDescCopies * kDescCopyStride / 2)
                    h2[lane + 64 * q] = zero2;
        }
        static_assert((kDescCopies * kDescCopyStride) % 2 == 0, "cleared in pairs");
        {
            const double e0 = exp((lane * lane) * wcons) * kFixWeightHalf; // both factors of a weight carry 2^-231
            if (lane <= radius && radius + lane < (BIG ? kDescTab : kDescTabSmall)) {
                etab[radius + lane] = e0;
                etab[radius - lane] = e0;
            }
            if (BIG && radius >= 64) {
                const int k2 = lane + 64;
                const double e1 = exp((k2 * k2) * wcons) * kFixWeightHalf;
                if (k2 <= radius && radius + k2 < kDescTab) {
                    etab[radius + k2] = e1;
                    etab[radius - k2] = e1;
                }
            }
        }
        static_assert(2 * kDescPackRadius + 1 <= kDescTab && 2 * (kDescTabSmall / 2 - 1) + 1 <= kDescTabSmall,
                      "the table covers every packed window");

        const double cs = cos_ori * inv_shw, ss = sin_ori * inv_shw;
        const double Tm = 2.5 * shw + 1e-6;
        const bool c_ok = fabs(cos_ori) > 1e-9, s_ok = fabs(sin_ori) > 1e-9;
        const double inv_c = c_ok ? 1.0 / cos_ori : 0.0, inv_s = s_ok ? 1.0 / sin_ori : 0.0;
        // Row spans: lane L holds rows L, L+64, L+128, L+192 (sets 0..3; two sets when every radius < 64), each as
        // (listed samples in the rows before it) << 16 | (its listed samples) << 8 | (its first listed column): the
        // samples of a window are numbered row by row, at most 255 * 255 of them.
        constexpr int kSets = BIG ? 4 : 2;
        unsigned int span[kSets];
        unsigned int n_total = 0;
        if (packed) {
#pragma unroll
            for (int k = 0; k < kSets; k++) {
                const int row = lane + 64 * k;
                span[k] = 0u;
                if (64 * k < W) { // (uniform)
                    int lo = 0, cnt = 0;
                    if (row < W)
                        descr_row_span(row - radius, cos_ori, sin_ori, inv_c, inv_s, c_ok, s_ok, Tm, radius, r.x, r.y, rows,
                                       cols, lo, cnt);
                    const unsigned int n = cnt > 0 ? (unsigned int)cnt : 0u;
                    const unsigned int incl = wave_scan_add(n);
                    span[k] = ((n_total + incl - n) << 16) | (n << 8) | ((unsigned int)(lo + radius) & 255u);
                    n_total += (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
                }
            }
        }
        // the lane's histogram copy, as the byte address of the bin (cell (-1, -1), orientation 0)
        const unsigned int hbytes = lds_address(hist + (lane % kDescCopies) * kDescCopyStride) - 5u * kDescCellBins * 8u;
        const unsigned int etab_at = lds_address(etab), lut_at = lds_address(lds.lut);
        wave_lds_fence();

        const unsigned int queue_lane = lds_address(queue + lane);
        int row = 0;    // wide windows: the row whose samples are taken next
        int q_base = 0; // listed windows: the number of the round's first sample
        while (packed ? q_base < (int)n_total : row < W) {
            // ---- fill: the next kDescQueue samples of the window (listed), or one row's span computed on the spot ----
            int count = 0, big_lo = 0;
            const int big_row = row;
            if (packed) {
                count = min((int)n_total - q_base, kDescQueue);
                // The rows that have samples in this round, one after the other: a row's k-th sample goes to slot
                // (samples before the row) + k - q_base.  Whole waves store: what lands behind a row's last sample is
                // overwritten by the next row or by the padding below (the slots behind the queue take the overrun of
                // the last row), and a store under a lane condition would cost more than the store (see
                // lds_add_fixed_pair_masked).  The loop runs on the scalar unit, which the 16 waves of a compute unit
                // share: its body is kept to a dozen scalar instructions and has no branch but the rare one for rows of
                // more than 64 samples.
#ifndef SG_X_NOFILL
#pragma unroll
                for (int k = 0; k < kSets; k++) {
                    if (64 * k < W) { // (uniform)
                        const unsigned int before = span[k] >> 16, n = (span[k] >> 8) & 255u;
                        // rows of this set that end before the round, and rows that begin before its end (the prefix
                        // sums ascend with the lane, so both are counts of a lane prefix)
                        const int r_first = __builtin_popcountll(__builtin_amdgcn_ballot_w64(before + n <= (unsigned int)q_base));
                        const int r_end = min(__builtin_popcountll(__builtin_amdgcn_ballot_w64(before < (unsigned int)(q_base + kDescQueue))),
                                              min(64, W - 64 * k));
                        // every lane works out its own row's part -- the first slot (as a byte offset) and the first
                        // entry -- so that the loop over the rows only has to hand them to the wave
                        const int pos = (int)before - q_base; // slot of the row's first sample (negative: it began in the round before)
                        const int k0 = max(-pos, 0);
                        const unsigned int row_at = 2u * (unsigned int)max(pos, 0);
                        const unsigned int row_first = ((((unsigned int)(64 * k + lane)) << 8) | (span[k] & 255u)) + (unsigned int)k0;
                        // (four rows per pass, so that their broadcasts and stores overlap; the last pass repeats its last row)
                        for (int rl = r_first; rl < r_end; rl += 4) {
                            unsigned int at[4], first[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int rj = min(rl + j, r_end - 1);
                                at[j] = (unsigned int)__builtin_amdgcn_readlane((int)row_at, rj);
                                first[j] = (unsigned int)__builtin_amdgcn_readlane((int)row_first, rj);
                            }
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                *(__attribute__((address_space(3))) unsigned short *)(size_t)(queue_lane + at[j]) =
                                    (unsigned short)(first[j] + (unsigned int)lane);
                        }
                        // rows with more than 64 samples left (few, and only in wide windows): their further stores, which
                        // come after the stores of the rows behind them and therefore end with the row
                        unsigned long long more = __builtin_amdgcn_ballot_w64((int)n - k0 > 64 && lane >= r_first && lane < r_end);
                        while (more) {
                            const int rl = __builtin_ctzll(more);
                            more &= more - 1;
                            const unsigned int at = queue_lane + (unsigned int)__builtin_amdgcn_readlane((int)row_at, rl);
                            const unsigned int first = (unsigned int)__builtin_amdgcn_readlane((int)row_first, rl) + (unsigned int)lane;
                            const int left = __builtin_amdgcn_readlane((int)n - k0, rl);
#pragma unroll
                            for (int u = 1; u < (BIG ? 4 : 2); u++)
                                if (lane + 64 * u < left)
                                    *(__attribute__((address_space(3))) unsigned short *)(size_t)(at + 128u * u) =
                                        (unsigned short)(first + 64u * u);
                        }
                    }
                }
#endif
                wave_lds_fence();
                // the slots up to the end of the last double trip, and the two trips of look-ahead behind it, repeat the last entry
                const unsigned int last_val = queue[count - 1];
                const int padded = ((count + 127) & ~127) + kDescQueuePad;
#pragma unroll
                for (int u = 0; u < 4; u++) // (lanes beyond the end write its last slot again)
                    queue[min(count + lane + 64 * u, padded - 1)] = (unsigned short)last_val;
                q_base += kDescQueue;
                wave_lds_fence();
            } else if (BIG) {
                descr_row_span(row - radius, cos_ori, sin_ori, inv_c, inv_s, c_ok, s_ok, Tm, radius, r.x, r.y, rows, cols,
                               big_lo, count);
                big_lo += radius;
                row++;
            }
            // ---- drain: double arithmetic on the listed samples ----
            // A trip takes 64 listed samples.  Its instruction stream is straight-line up to the histogram adds, and
            // everything it reads from LDS -- the two weight factors, the direction table entry, and the list entry
            // that the trip after the next will work on -- is requested before the arithmetic that does not need it, so
            // that a trip waits for LDS once, and by then the adds of the trip before have drained (LDS operations of a
            // wave complete in order: a read requested right behind eight adds waits for all of them).  The four neighbour pixels of the next
            // trip's samples are loaded at the top of the trip from the list entry read one trip earlier; the slots
            // behind the list repeat its last entry, so every fetch is issued unconditionally and the wait in front of
            // a slot's first use leaves exactly the younger slot's four loads in flight (vmcnt(4)).
            struct Slot {
                unsigned int rc; // (row << 8) | column inside the window (packed), or the column alone (wide windows)
                T nl, nr, nu, nd;
            };
            const int last_entry = count - 1;
            auto entry_of = [&](int e0) -> unsigned int { // list entries e0 .. e0+63
                if (packed)
                    return queue[e0 + lane];
                return (unsigned int)(big_lo + min(e0 + lane, last_entry));
            };
            auto issue_loads = [&](unsigned int rc, Slot &sl) {
                sl.rc = rc;
                unsigned int off;
                if (packed)
                    off = mad_u24(rc >> 8, pitch_b, (rc & 255u) * (unsigned int)sizeof(T)); // (pitch_b < 2^24 on this path)
                else
                    off = (unsigned int)big_row * pitch_b + rc * (unsigned int)sizeof(T);
#ifdef SG_X_NOLOAD // timing experiment: no pixel loads
                sl.nl = (T)off;
                sl.nr = (T)(off + 3u);
                sl.nu = (T)(off ^ 5u);
                sl.nd = (T)(off + 9u);
#else
                const char *p8 = gb + (size_t)off;
                sl.nl = *reinterpret_cast<const T *>(p8 - sizeof(T));
                sl.nr = *reinterpret_cast<const T *>(p8 + sizeof(T));
                sl.nu = *reinterpret_cast<const T *>(gb_up + (size_t)off);
                sl.nd = *reinterpret_cast<const T *>(gb_dn + (size_t)off);
#endif
            };
            // one trip: works on `cur`; starts the loads of `nxt` from rc_nxt; reads the list entries e_ahead.. into rc_ahead
            auto trip = [&](const Slot &cur, bool listed, Slot &nxt, unsigned int rc_nxt, int e_ahead, unsigned int &rc_ahead) {
                issue_loads(rc_nxt, nxt);
                const unsigned int sr = packed ? cur.rc >> 8 : (unsigned int)big_row, sc = packed ? cur.rc & 255u : cur.rc;
                double e_r = 0.0, e_c = 0.0;
#ifdef SG_X_NOLDSREAD // timing experiment: no LDS reads for the weights and the table entry
                e_r = (double)sr;
                e_c = (double)sc;
#else
                if (!BIG || packed) {
                    e_r = *(const __attribute__((address_space(3))) double *)(size_t)(etab_at + sr * 8u);
                    e_c = *(const __attribute__((address_space(3))) double *)(size_t)(etab_at + sc * 8u);
                }
#endif
                rc_ahead = entry_of(e_ahead);
                // (1) what needs neither the pixels nor LDS: the sample's place in the grid, shifted by one cell
                // (xs = xbin + 1 in (0, 5) for an accepted sample), so that truncation is the floor and the cell index
                // is never negative
                const int i = (int)sr - radius, jj = (int)sc - radius;
                const double xs = __builtin_fma(cs, (double)jj, __builtin_fma(-ss, (double)i, 2.5));
                const double ys = __builtin_fma(ss, (double)jj, __builtin_fma(cs, (double)i, 2.5));
                // 0 < v < 5 on the high word alone: 5.0 is 0x40140000:00000000, so v >= 5 (and every NaN) has a high
                // word of 0x40140000 or more, a negative v (and -0) one of 0x80000000 or more, +0 has 0, which the
                // subtraction wraps; a positive v below 2^-1042 (high word 0 as well) cannot come out of these sums of
                // products of numbers of ordinary size.
                const unsigned int hx = (unsigned int)((unsigned long long)__double_as_longlong(xs) >> 32);
                const unsigned int hy = (unsigned int)((unsigned long long)__double_as_longlong(ys) >> 32);
                // The sample adds to cell row r1 - 1 if 1 <= ys < 5 and to row r1 if 0 < ys < 4, and likewise in the
                // columns (the cells that exist around an accepted sample: 0 < xs, ys < 5); each of the four tests is
                // one unsigned compare of the shifted high word, and the four wave masks stay in scalar registers
                // until the adds.
                const unsigned long long m_listed = __builtin_amdgcn_ballot_w64(listed);
                const unsigned long long m_r0 = __builtin_amdgcn_ballot_w64(hy - 0x3FF00000u < 0x00240000u) & m_listed;
                const unsigned long long m_r1 = __builtin_amdgcn_ballot_w64(hy - 1u < 0x400FFFFFu) & m_listed;
                const unsigned long long m_c0 = __builtin_amdgcn_ballot_w64(hx - 0x3FF00000u < 0x00240000u);
                const unsigned long long m_c1 = __builtin_amdgcn_ballot_w64(hx - 1u < 0x400FFFFFu);
                const int c1 = (int)xs, r1 = (int)ys;
                const double d_r = __builtin_amdgcn_fract(ys), d_c = __builtin_amdgcn_fract(xs);
                __builtin_amdgcn_sched_barrier(0);
                // (2) the pixels: gradient, the seed of 1/magnitude, and from it the direction table entry to read
                const double dx = (double)cur.nr - (double)cur.nl;
                const double dy = (double)cur.nd - (double)cur.nu;
                const double m2 = __builtin_fma(dx, dx, dy * dy);
                const double y = __builtin_amdgcn_rsq(__builtin_fmax(m2, 1e-300));
                const double ax = fabs(dx), ay = fabs(dy);
                const double mx = __builtin_fmax(ax, ay), mn = __builtin_fmin(ax, ay);
                double k_dbl;
                const unsigned int lut_off = sg_sin_lut_index(mn, y, k_dbl);
                double2 ck; // {cos, angle} of the centre
#ifdef SG_X_NOLDSREAD
                ck.x = (double)lut_off;
                ck.y = k_dbl;
#else
                ck.x = *(const __attribute__((address_space(3))) double *)(size_t)(lut_at + lut_off);
                ck.y = *(const __attribute__((address_space(3))) double *)(size_t)(lut_at + lut_off + 8u);
#endif
                // (3) while that read is under way: the square root by one Newton-Raphson step on the seed (sg_math.h; the seed and
                // the step's residual also serve the direction)
                const double g = m2 * y, hh = 0.5 * y;
                const double res = __builtin_fma(-hh, g, 0.5);
                const double grad_mag = __builtin_fma(g, res, g);
                const double mxk = mx * k_dbl;
                __builtin_amdgcn_sched_barrier(0);
                // (4) everything that was read from LDS is used from here on.
                // obin = ((pi - atan2(dy,dx)) - ori) * 8/(2 pi) = obin_c - atan2(dy,dx) in units of pi/4; the
                // histogram index below wraps by itself, so the angle is not brought into [0, 2 pi) first
                // (obin_c carries +8, which keeps the coordinate positive and leaves the index unchanged)
                double q = sg_angle_sin_lut(mn, mxk, y, res, ck);
                q = ay > ax ? 2.0 - q : q;
                q = dx < 0.0 ? 4.0 - q : q;
                const double obin = obin_c - __builtin_copysign(q, dy);
                double weight;
                if (BIG && !packed)
                    weight = exp((i * i + jj * jj) * wcons) * kFixWeight;
                else
                    weight = e_r * e_c;
                const double mag = grad_mag * weight;
                // InterpHistEntry: cell indices r1 = r0 + 1, c1 = c0 + 1 in 0..4, fractions by v_fract_f64
                const int o0 = (int)obin;
                const double d_o = __builtin_amdgcn_fract(obin);
                // the eight bins lie at fixed distances from the (r0, c0) cell's bin o0; the bin above o0 is the
                // next entry (the ninth entry of a cell stands for bin 0)
                unsigned int hb = (((unsigned int)o0 & 7u) << 3) + hbytes;
                hb = mad_u24((unsigned int)c1, kDescCellBins * 8u, hb);
                hb = mad_u24((unsigned int)r1, 4u * kDescCellBins * 8u, hb);
                const double w_o1 = d_o * kFixFrac, w_o0 = kFixFrac - w_o1; // (1 - d_o) and d_o, scaled (exactly)
                // mag split over the two rows, each row over the two columns: the far share is a product, the near
                // share the remainder
                const double v_r1 = mag * d_r, v_r0 = mag - v_r1;
                const double v_01 = v_r0 * d_c, v_00 = v_r0 - v_01;
                const double v_11 = v_r1 * d_c, v_10 = v_r1 - v_11;
                lds_add_fixed_pair_masked<0>(hb, v_00, w_o0, w_o1, m_r0 & m_c0);
                lds_add_fixed_pair_masked<kDescCellBins * 8>(hb, v_01, w_o0, w_o1, m_r0 & m_c1);
                lds_add_fixed_pair_masked<4 * kDescCellBins * 8>(hb, v_10, w_o0, w_o1, m_r1 & m_c0);
                lds_add_fixed_pair_masked<5 * kDescCellBins * 8>(hb, v_11, w_o0, w_o1, m_r1 & m_c1);
            };
            trips += (unsigned int)((count + 63) >> 6);
#ifdef SG_X_NODRAIN // timing experiment: everything but the trips
            if (count < 0) {
#else
            if (count > 0) {
#endif
                Slot s0, s1;
                unsigned int rc_a = entry_of(0), rc_b = entry_of(64);
                // (both entries are in registers before the loop is entered: a read still pending at its head would
                // put a wait for LDS there, which inside the loop would wait for the histogram adds of the trip before)
                asm volatile("" : "+v"(rc_a), "+v"(rc_b));
                issue_loads(rc_a, s0);
                for (int base = 0; base < count; base += 128) { // count is the wave's: a uniform loop
                    const int left = count - base;
                    trip(s0, lane < left, s1, rc_b, base + 128, rc_a);
                    trip(s1, lane < left - 64, s0, rc_a, base + 192, rc_b);
                }
            }
            wave_lds_fence();
        }

        // lane L finishes bins 2L and 2L+1: orientations 2(L & 3) and the next of cell L >> 2
        unsigned long long u0 = 0, u1 = 0;
        {
            const int cell = (lane >> 2) * kDescCellBins, ob = (lane & 3) * 2;
#pragma unroll
            for (int q2 = 0; q2 < kDescCopies; q2++) {
                u0 += hist[q2 * kDescCopyStride + cell + ob];
                u1 += hist[q2 * kDescCopyStride + cell + ob + 1];
                const unsigned long long wrapped = hist[q2 * kDescCopyStride + cell + 8];
                u0 += ob == 0 ? wrapped : 0ull;
            }
        }
        // whole units to double (one rounding: the high word's 2^32 multiples and the low word are both exact)
        double v0 = __builtin_fma((double)(unsigned int)(u0 >> 32), 4294967296.0, (double)(unsigned int)u0) * (1.0 / kFixScale);
        double v1 = __builtin_fma((double)(unsigned int)(u1 >> 32), 4294967296.0, (double)(unsigned int)u1) * (1.0 / kFixScale);
        double len_inv = 1.0 / sqrt(wave_sum(v0 * v0 + v1 * v1));
        v0 *= len_inv;
        v1 *= len_inv;
        v0 = v0 > 0.2 ? 0.2 : v0; // DESCR_MAG_THR
        v1 = v1 > 0.2 ? 0.2 : v1;
        len_inv = 1.0 / sqrt(wave_sum(v0 * v0 + v1 * v1));
        v0 *= len_inv;
        v1 *= len_inv;
        if (prequant) {
            double2 pq;
            pq.x = v0;
            pq.y = v1;
            *reinterpret_cast<double2 *>(prequant + (size_t)b * pq_img + (size_t)f * 128 + 2 * lane) = pq;
        }
        int q0 = 0, q1 = 0;
        if (v0 == v0) { // an all-zero window gives 0/0; keep such a descriptor at zero
            q0 = (int)(512.0 * v0); // INT_DESCR_FCTR, truncating
            q0 = q0 < 255 ? q0 : 255;
        }
        if (v1 == v1) {
            q1 = (int)(512.0 * v1);
            q1 = q1 < 255 ? q1 : 255;
        }
        *reinterpret_cast<unsigned short *>(desc + (size_t)b * desc_img + (size_t)f * 128 + 2 * lane) =
            (unsigned short)(q0 | (q1 << 8));
        wave_lds_fence(); // the histograms are cleared for the next feature
    }
    if (lane == 0 && trips)
        atomicAdd(&trip_slots[(size_t)b * kTripWordsPerImage + (size_t)(blockIdx.x % kTripSlots) * kTripStride],
                  (unsigned long long)trips);
}

void launch_descr(const OctaveTab &tab, const FeatRec *frec, size_t frec_img_stride, const unsigned int *feat_count,
                  uint8_t *desc, size_t desc_img_stride, double *prequant, size_t pq_img_stride, unsigned int cap,
                  int max_radius, int B, int prec, unsigned long long *trip_slots, hipStream_t s)
{
    unsigned int waves = feature_blocks(cap, B, 384, 384, env().descr_waves);
    // about four features per wave at least (the capacity is about two and a half times a natural image's count): with
    // fewer, the set-up of a wave and the gaps between generations of waves cost more than the finer balance gains
    // -- and whole generations (16 waves per compute unit are resident): a last generation of a few waves costs as much
    // time as a full one
    if (env().descr_waves <= 0 && waves > cap / 10) {
        const unsigned int gen = 4u * SG_DESCR_WPE * (unsigned int)env().n_cu, nb = (unsigned int)(B > 0 ? B : 1);
        unsigned int total = cap / 10 * nb;
        total = total > gen ? total - total % gen : gen; // at least the one generation that costs nothing extra
        const unsigned int per_image = total / nb > 384u ? total / nb : 384u;
        waves = per_image < waves ? per_image : waves;
    }
    if (waves == 0)
        return;
    dim3 g(waves, B);
    bool big = 2 * max_radius + 1 > kDescTabSmall; // a window row or column beyond the small table
    for (int o = 0; o < tab.octaves; o++) // ... or a row of 16 MB
        big = big || (size_t)tab.pitch[o] * (prec == SG_F64 ? 8u : 4u) >= ((size_t)1 << 24);
#define SG_DESCR_LAUNCH(TT, BB)                                                                                       \
    descr_k<TT, BB><<<g, 64, 0, s>>>(tab, frec, frec_img_stride, feat_count, desc, desc_img_stride, prequant,         \
                                     pq_img_stride, cap, trip_slots)
    if (prec == SG_F64) {
        if (big)
            SG_DESCR_LAUNCH(double, true);
        else
            SG_DESCR_LAUNCH(double, false);
    } else {
        if (big)
            SG_DESCR_LAUNCH(float, true);
        else
            SG_DESCR_LAUNCH(float, false);
    }
#undef SG_DESCR_LAUNCH
}

// One thread per feature: follows feat_src -> kp_src -> kp / ori_tmp once and leaves a compact record.
__global__ __launch_bounds__(256) void feat_gather_k(const KpRec *kp, size_t kp_img, const unsigned int *kp_src,
                                                     size_t src_img, const unsigned int *feat_src, size_t fsrc_img,
                                                     const unsigned int *feat_count, const double *ori_tmp,
                                                     size_t ori_img, FeatRec *frec, size_t frec_img, double *fscale,
                                                     size_t fs_img, unsigned int cap)
{
    const int b = blockIdx.y;
    const unsigned int n = min(feat_count[b], cap);
    for (unsigned int f = blockIdx.x * 256u + threadIdx.x; f < n; f += gridDim.x * 256u) {
        const unsigned int fs = feat_src[(size_t)b * fsrc_img + f];
        const unsigned int j = fs >> 6, pbin = fs & 63;
        const unsigned int slot = kp_src[(size_t)b * src_img + j];
        const KpRec r = kp[(size_t)b * kp_img + slot];
        FeatRec o;
        o.x = r.x; o.y = r.y; o.octave = r.octave; o.interval = r.interval;
        o.oscale = r.oscale; o.scale = r.scale;
        o.ori = ori_tmp[(size_t)b * ori_img + (size_t)j * kOriBins + pbin];
        o.cos_ori = cos(o.ori);
        o.sin_ori = sin(o.ori);
        o.kp_slot = slot; o.pad = 0;
        frec[(size_t)b * frec_img + f] = o;
        fscale[(size_t)b * fs_img + f] = r.scale;
    }
}

void launch_feat_gather(const KpRec *kp, size_t kp_img_stride, const unsigned int *kp_src, size_t src_img_stride,
                        const unsigned int *feat_src, size_t fsrc_img_stride, const unsigned int *feat_count,
                        const double *ori_tmp, size_t ori_img_stride, FeatRec *frec, size_t frec_img_stride,
                        double *fscale, size_t fs_img_stride, unsigned int cap, int B, hipStream_t s)
{
    if (cap == 0)
        return;
    unsigned int blocks = (cap + 255) / 256;
    if (blocks > 512)
        blocks = 512;
    dim3 g(blocks, B);
    feat_gather_k<<<g, 256, 0, s>>>(kp, kp_img_stride, kp_src, src_img_stride, feat_src, fsrc_img_stride, feat_count,
                                    ori_tmp, ori_img_stride, frec, frec_img_stride, fscale, fs_img_stride, cap);
}

// ---- ordering: stable rank by scale -----------------------------------------------------------------------
// rank(i) = #{ j : scale_j < scale_i  or  (scale_j == scale_i and j < i) } -- the position of feature i in a
// stable ascending sort by scale (reference sift/sift.cpp:1640-1643,1703 sorts by scale alone).
// Scales are positive doubles, so their bit patterns order like their values; the top bits (exponent and 10
// mantissa bits, clamped) give a monotone bucket index.  Features are counted per bucket, the counts are
// scanned, indices are scattered into per-bucket lists (in any order), and each feature then counts the
// smaller keys inside its own bucket only: rank = bucket offset + that count.

__device__ __forceinline__ unsigned int rank_bucket(double s)
{
    long long k = (__double_as_longlong(s) >> 42) - (0x3FE0000000000000ll >> 42); // 0.5 -> 0
    k = k < 0 ? 0 : (k >= kRankBuckets ? kRankBuckets - 1 : k);
    return (unsigned int)k;
}

__global__ __launch_bounds__(256) void rank_hist_k(const double *fscale, size_t fs_img, const unsigned int *feat_count,
                                                   unsigned int *bcount, size_t bc_img, unsigned int cap)
{
    const int b = blockIdx.y;
    const unsigned int n = min(feat_count[b], cap);
    for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
        atomicAdd(&bcount[(size_t)b * bc_img + rank_bucket(fscale[(size_t)b * fs_img + i])], 1u);
}

__global__ __launch_bounds__(256) void rank_scatter_k(const double *fscale, size_t fs_img,
                                                      const unsigned int *feat_count, const unsigned int *boff,
                                                      unsigned int *bcursor, size_t bc_img, RankEntry *blist,
                                                      size_t bl_img, unsigned int cap)
{
    const int b = blockIdx.y;
    const unsigned int n = min(feat_count[b], cap);
    for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const double sc = fscale[(size_t)b * fs_img + i];
        unsigned int k = rank_bucket(sc);
        unsigned int pos = boff[(size_t)b * bc_img + k] + atomicAdd(&bcursor[(size_t)b * bc_img + k], 1u);
        RankEntry e;
        e.scale = sc;
        e.index = i;
        e.pad = 0;
        blist[(size_t)b * bl_img + pos] = e;
    }
}

__global__ __launch_bounds__(256) void rank_count_k(const double *fscale, size_t fs_img, const unsigned int *feat_count,
                                                    const unsigned int *boff, size_t bc_img,
                                                    const RankEntry *blist, size_t bl_img, unsigned int *rank,
                                                    size_t rank_img, unsigned int cap)
{
    const int b = blockIdx.y;
    const unsigned int n = min(feat_count[b], cap);
    const double *fsb = fscale + (size_t)b * fs_img;
    const RankEntry *bl = blist + (size_t)b * bl_img;
    for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const double my = fsb[i];
        const unsigned int k = rank_bucket(my);
        const unsigned int start = boff[(size_t)b * bc_img + k];
        const unsigned int end = k + 1 < (unsigned int)kRankBuckets ? boff[(size_t)b * bc_img + k + 1] : n;
        unsigned int cnt = 0;
        for (unsigned int p = start; p < end; p++) {
            const RankEntry e = bl[p];
            cnt += (e.scale < my || (e.scale == my && e.index < i)) ? 1u : 0u;
        }
        rank[(size_t)b * rank_img + i] = start + cnt;
    }
}

void launch_rank(const double *fscale, size_t fs_img_stride, const unsigned int *feat_count, unsigned int *rank,
                 size_t rank_img_stride, unsigned int *bwork, RankEntry *blist, unsigned int *scratch_totals,
                 unsigned int *overflow_dummy, unsigned int cap, int B, hipStream_t s)
{
    if (cap == 0)
        return;
    // bwork: per image [2][kRankBuckets]: counts (scanned in place to offsets), then scatter cursors
    const size_t bc_img = 2 * (size_t)kRankBuckets;
    (void)hipMemsetAsync(bwork, 0, sizeof(unsigned int) * bc_img * B, s);
    // at most 256 workgroups per image in a batch; a frame alone (an 8K frame has 360 000 features, and equal scales by
    // the thousand in one bucket) may take 4096, so that its in-bucket counting is spread over the chip
    const unsigned int most = std::max(256u, 4096u / (unsigned int)(B > 0 ? B : 1));
    unsigned int blocks = (cap + 255) / 256;
    if (blocks > most)
        blocks = most;
    dim3 g(blocks, B);
    rank_hist_k<<<g, 256, 0, s>>>(fscale, fs_img_stride, feat_count, bwork, bc_img, cap);
    launch_scan(bwork, bc_img, kRankBuckets, scratch_totals, 0xffffffffu, overflow_dummy, B, s);
    rank_scatter_k<<<g, 256, 0, s>>>(fscale, fs_img_stride, feat_count, bwork, bwork + kRankBuckets, bc_img, blist,
                                     cap, cap);
    rank_count_k<<<g, 256, 0, s>>>(fscale, fs_img_stride, feat_count, bwork, bc_img, blist, cap, rank, rank_img_stride,
                                   cap);
}

// ---- output packing ------------------------------------------------------------------------------------------
// Position = rank (sorted) or emission index.  MODE 0: packed records, 16 lanes per feature; MODE 1: sg_keypoint
// records and MODE 2: packed records at an unaligned address, one wave per feature.
template <int MODE>
__global__ __launch_bounds__(256) void pack_k(const KpRec *kp, size_t kp_img, const unsigned int *kp_src, size_t src_img,
                                              const unsigned int *feat_src, size_t fsrc_img,
                                              const unsigned int *feat_count, const double *ori_tmp, size_t ori_img,
                                              const uint8_t *desc, size_t desc_img, const unsigned int *rank,
                                              size_t rank_img, void *out, size_t out_img, unsigned int out_cap,
                                              unsigned int *out_counts, unsigned int *overflow, int sorted,
                                              unsigned int cap)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const unsigned int n = min(feat_count[b], cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (out_counts)
            out_counts[b] = n < out_cap ? n : out_cap;
        if (n > out_cap)
            atomicMax(overflow, n);
    }
    if (MODE == 3) {
        // as MODE 0, with the full-precision header: lanes 0..7 copy the descriptor, lane 8 writes the 96 header bytes
        const int sub = threadIdx.x & 15;
        for (unsigned int f = blockIdx.x * 16u + (threadIdx.x >> 4); f < n; f += gridDim.x * 16u) {
            const unsigned int pos = sorted ? rank[(size_t)b * rank_img + f] : f;
            if (pos >= out_cap)
                continue;
            WireRec *o = reinterpret_cast<WireRec *>(out) + (size_t)b * out_img + pos;
            if (sub < 8) {
                const uint4 v = reinterpret_cast<const uint4 *>(desc + (size_t)b * desc_img + (size_t)f * 128)[sub];
                reinterpret_cast<uint4 *>(o->desc)[sub] = v;
            } else if (sub == 8) {
                const unsigned int fs = feat_src[(size_t)b * fsrc_img + f];
                const unsigned int j = fs >> 6, pbin = fs & 63;
                const KpRec *r = kp + (size_t)b * kp_img + kp_src[(size_t)b * src_img + j];
                o->octave = r->octave; o->interval = r->interval; o->x = r->x; o->y = r->y;
                o->offi = r->offi; o->scale = r->scale; o->dx = r->dx; o->dy = r->dy;
                o->offx = r->offx; o->offy = r->offy; o->oscale = r->oscale;
                o->ori = ori_tmp[(size_t)b * ori_img + (size_t)j * kOriBins + pbin];
                o->val = r->val; o->pad = 0.0;
            }
        }
        return;
    }
    if (MODE == 0) {
        // 16 lanes per feature, so that a wave follows four index chains at a time: lanes 0..7 of a group copy the
        // descriptor in 16-byte pieces, lane 8 writes the 16-byte header
        const int sub = threadIdx.x & 15;
        for (unsigned int f = blockIdx.x * 16u + (threadIdx.x >> 4); f < n; f += gridDim.x * 16u) {
            const unsigned int pos = sorted ? rank[(size_t)b * rank_img + f] : f;
            if (pos >= out_cap)
                continue;
            sg_packed_feature *o = reinterpret_cast<sg_packed_feature *>(out) + (size_t)b * out_img + pos;
            static_assert(sizeof(sg_packed_feature) == 144 && offsetof(sg_packed_feature, desc) == 16,
                          "16-byte header, then the descriptor");
            if (sub < 8) {
                const uint4 v = reinterpret_cast<const uint4 *>(desc + (size_t)b * desc_img + (size_t)f * 128)[sub];
                reinterpret_cast<uint4 *>(o->desc)[sub] = v;
            } else if (sub == 8) {
                const unsigned int fs = feat_src[(size_t)b * fsrc_img + f];
                const unsigned int j = fs >> 6, pbin = fs & 63;
                const KpRec *r = kp + (size_t)b * kp_img + kp_src[(size_t)b * src_img + j];
                const double ori = ori_tmp[(size_t)b * ori_img + (size_t)j * kOriBins + pbin];
                float4 h;
                h.x = (float)r->dx; h.y = (float)r->dy; h.z = (float)r->scale; h.w = (float)ori;
                *reinterpret_cast<float4 *>(o) = h;
            }
        }
        return;
    }
    for (unsigned int f = blockIdx.x * 4u + (threadIdx.x >> 6); f < n; f += gridDim.x * 4u) {
        const unsigned int pos = sorted ? rank[(size_t)b * rank_img + f] : f;
        if (pos >= out_cap)
            continue;
        const unsigned int fs = feat_src[(size_t)b * fsrc_img + f];
        const unsigned int j = fs >> 6, pbin = fs & 63;
        const KpRec r = kp[(size_t)b * kp_img + kp_src[(size_t)b * src_img + j]];
        const double ori = ori_tmp[(size_t)b * ori_img + (size_t)j * kOriBins + pbin];
        const uint8_t *d = desc + (size_t)b * desc_img + (size_t)f * 128;
        if (MODE != 1) { // MODE 2: packed records at an address that is not a multiple of 16
            sg_packed_feature *o = reinterpret_cast<sg_packed_feature *>(out) + (size_t)b * out_img + pos;
            if (lane < 32)
                reinterpret_cast<uint32_t *>(o->desc)[lane] = reinterpret_cast<const uint32_t *>(d)[lane];
            else if (lane == 32) {
                o->x = (float)r.dx; o->y = (float)r.dy; o->scale = (float)r.scale; o->ori = (float)ori;
            }
        } else {
            sg_keypoint *o = reinterpret_cast<sg_keypoint *>(out) + (size_t)b * out_img + pos;
            o->descriptor[lane] = (double)d[lane];
            o->descriptor[lane + 64] = (double)d[lane + 64];
            if (lane == 0) {
                o->octave = r.octave; o->interval = r.interval; o->offset_interval = r.offi;
                o->x = r.x; o->y = r.y; o->scale = r.scale; o->dx = r.dx; o->dy = r.dy;
                o->offset_x = r.offx; o->offset_y = r.offy; o->octave_scale = r.oscale; o->ori = ori;
                // descr_length and the 4 padding bytes behind it in one store, so that records are byte-reproducible
                *reinterpret_cast<unsigned long long *>(&o->descr_length) = 128ull; o->val = r.val;
            }
        }
    }
}

void launch_pack(const KpRec *kp, size_t kp_img_stride, const unsigned int *kp_src, size_t src_img_stride,
                 const unsigned int *feat_src, size_t fsrc_img_stride, const unsigned int *feat_count,
                 const double *ori_tmp, size_t ori_img_stride, const uint8_t *desc, size_t desc_img_stride,
                 const unsigned int *rank, size_t rank_img_stride, void *out, size_t out_img_stride,
                 unsigned int out_cap, unsigned int *out_counts, unsigned int *overflow, int mode, int sorted,
                 unsigned int cap, int B, hipStream_t s)
{
    unsigned int blocks = (mode == 0 || mode == 3) ? (cap + 15) / 16 : (cap + 3) / 4;
    if (blocks > 2048)
        blocks = 2048;
    if (blocks == 0)
        blocks = 1;
    dim3 g(blocks, B);
    if (mode == 3)
        pack_k<3><<<g, 256, 0, s>>>(kp, kp_img_stride, kp_src, src_img_stride, feat_src, fsrc_img_stride, feat_count,
                                    ori_tmp, ori_img_stride, desc, desc_img_stride, rank, rank_img_stride, out,
                                    out_img_stride, out_cap, out_counts, overflow, sorted, cap);
    else if (mode == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)desc % 16 == 0 && desc_img_stride % 16 == 0)
        pack_k<0><<<g, 256, 0, s>>>(kp, kp_img_stride, kp_src, src_img_stride, feat_src, fsrc_img_stride, feat_count,
                                    ori_tmp, ori_img_stride, desc, desc_img_stride, rank, rank_img_stride, out,
                                    out_img_stride, out_cap, out_counts, overflow, sorted, cap);
    else if (mode == 0)
        pack_k<2><<<g, 256, 0, s>>>(kp, kp_img_stride, kp_src, src_img_stride, feat_src, fsrc_img_stride, feat_count,
                                    ori_tmp, ori_img_stride, desc, desc_img_stride, rank, rank_img_stride, out,
                                    out_img_stride, out_cap, out_counts, overflow, sorted, cap);
    else
        pack_k<1><<<g, 256, 0, s>>>(kp, kp_img_stride, kp_src, src_img_stride, feat_src, fsrc_img_stride, feat_count,
                                    ori_tmp, ori_img_stride, desc, desc_img_stride, rank, rank_img_stride, out,
                                    out_img_stride, out_cap, out_counts, overflow, sorted, cap);
}

} // namespace sg
