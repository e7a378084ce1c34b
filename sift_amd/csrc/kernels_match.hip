// kernels_match.hip -- brute-force descriptor matching with ratio test for gfx950.
//
// The step after the extraction path in the reference's demo (sift/源.cpp:88-156): for every feature of
// image 1 the Euclidean distances to all features of image 2 over the 128 descriptor values, the smallest
// (first index on ties) and the second smallest, and a match when smallest / second < ratio.
//
// Descriptor values are integers 0..255 (HistToDescriptor, sift/sift.cpp:1567-1597), so the squared
// distance is an exact integer:  |a-b|^2 = |a'|^2 + |b'|^2 - 2 a'.b'  with a' = a - 128, b' = b - 128 as
// signed bytes (the shift cancels in the difference).  The dot products are one dense N x M x 128
// contraction and run on the matrix cores (v_mfma_i32_32x32x32_i8, exact in int32); the running
// smallest / second smallest per feature is kept by the lanes that own the output column.
#include "sg_internal.h"

namespace sg {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kMatchPadNorm = 1 << 26; // squared norm given to the padding rows: never the smallest (real ones are below 2^23)
constexpr int kMatchKeyShift = 4;      // a candidate's key = (e << 4) | its accumulator register index

// ---- preparation: descriptors -> signed-byte rows and squared norms ------------------------------------
// 32 lanes per feature, four values per lane.  MODE 0: sg_packed_feature (128 u8 at byte 16, 144-byte
// records); MODE 1: sg_keypoint (128 doubles at byte 88, 1120-byte records).  Rows n .. n_pad-1 are zero
// with the padding norm.  A value that is not an integer in 0..255 sets *bad.
template <int MODE>
__global__ __launch_bounds__(256) void match_prep_k(const unsigned char *feats, unsigned int n, unsigned int n_pad,
                                                    signed char *rows, int *norms, unsigned int *bad)
{
    const unsigned int f = blockIdx.x * 8u + (threadIdx.x >> 5);
    const int part = threadIdx.x & 31;
    if (f >= n_pad)
        return;
    int v[4] = {128, 128, 128, 128}; // shifted value 0 for the padding rows
    bool ok = true;
    if (f < n) {
        if (MODE == 0) {
            const unsigned char *p = feats + (size_t)f * 144 + 16 + part * 4;
#pragma unroll
            for (int q = 0; q < 4; q++)
                v[q] = p[q];
        } else {
            const double *p = reinterpret_cast<const double *>(feats + (size_t)f * 1120 + 88) + part * 4;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const double d = p[q];
                ok = ok && d >= 0.0 && d <= 255.0 && d == floor(d);
                v[q] = ok ? (int)d : 128;
            }
        }
    }
    int sq = 0;
    unsigned int packed = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int s = v[q] - 128;
        sq += s * s;
        packed |= (unsigned int)(s & 0xff) << (8 * q);
    }
    reinterpret_cast<unsigned int *>(rows + (size_t)f * 128)[part] = packed;
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1)
        sq += __shfl_xor(sq, d, 32);
    if (part == 0)
        norms[f] = f < n ? sq : kMatchPadNorm;
    if (!ok)
        atomicOr(bad, 1u);
}

// ---- matching ------------------------------------------------------------------------------------------
// One workgroup of kMatchWaves waves per kMatchGroups x 32 features of image 1.  They are the B operands (columns) of the MFMAs,
// loaded once; the features of image 2 stream through as the A operand, 32 at a time (wave w takes blocks
// w, w + kMatchWaves, ...), four K-steps of 32 bytes each.  In the 32x32 result lane l holds column l&31 -- one feature of
// image 1 -- and 16 rows (features of image 2): row = (reg&3) + 8*(reg>>2) + 4*(l>>5).  Every lane keeps a
// running (smallest, its index, second smallest) of  e = |b'|^2 - 2 a'.b'  (|a'|^2 is the same for every
// candidate and is added at the end); a lane meets its rows in ascending index order, so a strict < keeps the
// first index, and the partial results of a column (two lane halves per wave) are merged by
// (value, index) at the end.  The ratio test is evaluated in double exactly as the demo does:
// sqrt(d1) / sqrt(d2) < ratio, with the second distance infinite when image 2 has one feature.
constexpr int kMatchWaves = 8;
#ifndef SG_MATCH_GROUPS
#define SG_MATCH_GROUPS 4
#endif
constexpr int kMatchGroups = SG_MATCH_GROUPS; // groups of 32 image-1 features per workgroup
static_assert(kMatchGroups <= kMatchWaves, "one wave finishes each group");

struct MatchBest {
    int m1, idx, m2;
};

// median of three signed integers, one instruction
__device__ __forceinline__ int med3_i32(int a, int b, int c)
{
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// a * b + c with a and b taken as signed 24-bit integers, one instruction
__device__ __forceinline__ int mad_i24(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ MatchBest match_merge(const MatchBest &a, const MatchBest &b)
{
    const bool first = a.m1 < b.m1 || (a.m1 == b.m1 && a.idx < b.idx);
    MatchBest r;
    r.m1 = first ? a.m1 : b.m1;
    r.idx = first ? a.idx : b.idx;
    r.m2 = first ? min(a.m2, b.m1) : min(b.m2, a.m1);
    return r;
}

// The ratio test in double exactly as the demo evaluates it (sqrt(d1) / sqrt(d2) < ratio, the second distance
// infinite when image 2 has one feature), from e = d - |a'|^2 of the nearest and second nearest feature.
__device__ __forceinline__ void match_finish(const MatchBest &b, int ci, unsigned int n2, double ratio, unsigned int i,
                                             int *__restrict__ index2, unsigned int *__restrict__ dist2)
{
    int res = -1;
    unsigned int d1 = 0xffffffffu, d2 = 0xffffffffu;
    if (n2 >= 1) {
        d1 = (unsigned int)(b.m1 + ci);
        const double min_value = sqrt((double)d1);
        double pen_value = 1.7976931348623157e308; // DBL_MAX, the demo's initial value (源.cpp:102)
        if (n2 >= 2) {
            d2 = (unsigned int)(b.m2 + ci);
            pen_value = sqrt((double)d2);
        }
        if (min_value / pen_value < ratio)
            res = b.idx;
    }
    index2[i] = res;
    if (dist2) {
        dist2[2 * (size_t)i] = d1;
        dist2[2 * (size_t)i + 1] = d2;
    }
}

__global__ __launch_bounds__(64 * kMatchWaves) void match_k(const signed char *__restrict__ rows1,
                                                            const int *__restrict__ norms1, unsigned int n1,
                                                            unsigned int n1_pad, const signed char *__restrict__ rows2,
                                                            const int *__restrict__ norms2, unsigned int n2,
                                                            unsigned int n2_pad, double ratio, int *__restrict__ index2,
                                                            unsigned int *__restrict__ dist2,
                                                            MatchBest *__restrict__ partials, unsigned int blocks_per_slice)
{
    __shared__ MatchBest part[kMatchWaves][kMatchGroups][32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    // kMatchGroups groups of 32 image-1 features per workgroup: every block of image-2 features a wave loads serves
    // all of them (the image-2 matrix is streamed n1 / (32 * kMatchGroups) times in all)
    v4i bfrag[kMatchGroups][4];
    unsigned int i_of[kMatchGroups];
#pragma unroll
    for (int g = 0; g < kMatchGroups; g++) {
        i_of[g] = min((blockIdx.x * kMatchGroups + g) * 32u + r, n1_pad - 1); // groups past the end repeat the last row
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
            bfrag[g][ks] = *reinterpret_cast<const v4i *>(rows1 + (size_t)i_of[g] * 128 + ks * 32 + h * 16);
    }
    // Per group the lane keeps the two smallest keys seen (k1 <= k2), key = (e << 4) | register index: one min and one
    // median-of-three per candidate.  A lane meets its rows in ascending order and the register index ascends with the
    // row inside a block, so inside a block the smaller key is also the first index on ties; at the end of a block
    // the index bits of k1 are cleared (after the block and register of a new k1 have been noted), so that an equal
    // distance in a later block compares as not smaller.
    int k1[kMatchGroups], k2[kMatchGroups], blk[kMatchGroups], reg[kMatchGroups];
#pragma unroll
    for (int g = 0; g < kMatchGroups; g++) {
        k1[g] = 0x7fffffff;
        k2[g] = 0x7fffffff;
        blk[g] = 0;
        reg[g] = 0;
    }
    constexpr unsigned int kStep = 32u * kMatchWaves;
    v4i afrag[4];
    // gridDim.y slices of image 2, blocks_per_slice blocks of 32 features each (a multiple of kMatchWaves)
    const unsigned int j_end = min(n2_pad, 32u * blocks_per_slice * (blockIdx.y + 1));
    unsigned int jb = 32u * (blocks_per_slice * blockIdx.y + wave);
    if (jb < j_end) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
            afrag[ks] = *reinterpret_cast<const v4i *>(rows2 + (size_t)(jb + r) * 128 + ks * 32 + h * 16);
    }
    for (; jb < j_end; jb += kStep) {
        v16i acc[kMatchGroups];
#pragma unroll
        for (int g = 0; g < kMatchGroups; g++) {
#pragma unroll
            for (int q = 0; q < 16; q++)
                acc[g][q] = 0;
#pragma unroll
            for (int ks = 0; ks < 4; ks++)
                acc[g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[ks], bfrag[g][ks], acc[g], 0, 0, 0);
        }
        // this wave's next block of image-2 features while the vector ALU folds the products in
        if (jb + kStep < j_end) {
#pragma unroll
            for (int ks = 0; ks < 4; ks++)
                afrag[ks] = *reinterpret_cast<const v4i *>(rows2 + (size_t)(jb + kStep + r) * 128 + ks * 32 + h * 16);
        }
        int before[kMatchGroups];
#pragma unroll
        for (int g = 0; g < kMatchGroups; g++)
            before[g] = k1[g];
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            const int4 cj = *reinterpret_cast<const int4 *>(norms2 + jb + 8 * q4 + 4 * h);
            const int c[4] = {cj.x, cj.y, cj.z, cj.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // key = ((c - 2 dot) << 4) | register = ((c << 4) | register) - 32 dot; |dot| < 2^22 fits the 24-bit multiply
                const int ckey = (c[q] << kMatchKeyShift) | (4 * q4 + q);
#pragma unroll
                for (int g = 0; g < kMatchGroups; g++) {
                    const int key = mad_i24(acc[g][4 * q4 + q], -(2 << kMatchKeyShift), ckey);
                    k2[g] = med3_i32(k1[g], k2[g], key);
                    k1[g] = min(k1[g], key);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < kMatchGroups; g++) {
            const bool fresh = k1[g] != before[g];
            blk[g] = fresh ? (int)jb : blk[g];
            reg[g] = fresh ? (k1[g] & ((1 << kMatchKeyShift) - 1)) : reg[g];
            k1[g] &= ~((1 << kMatchKeyShift) - 1);
        }
    }
    // per lane: smallest e, its row index, second smallest e; then merge the two halves of each column, then the waves
    MatchBest best[kMatchGroups];
#pragma unroll
    for (int g = 0; g < kMatchGroups; g++) {
        best[g].m1 = k1[g] >> kMatchKeyShift;
        best[g].m2 = k2[g] >> kMatchKeyShift;
        best[g].idx = blk[g] + (reg[g] & 3) + 8 * (reg[g] >> 2) + 4 * h;
        MatchBest other;
        other.m1 = __shfl_xor(best[g].m1, 32);
        other.m2 = __shfl_xor(best[g].m2, 32);
        other.idx = __shfl_xor(best[g].idx, 32);
        best[g] = match_merge(best[g], other);
        if (h == 0)
            part[wave][g][r] = best[g];
    }
    __syncthreads();
    if (wave < kMatchGroups && h == 0) {
        const int g = wave; // wave g finishes group g
        const unsigned int i = (blockIdx.x * kMatchGroups + g) * 32u + r;
        if (i < n1) {
            MatchBest b = part[0][g][r];
#pragma unroll
            for (int w = 1; w < kMatchWaves; w++)
                b = match_merge(b, part[w][g][r]);
            if (gridDim.y == 1)
                match_finish(b, norms1[i], n2, ratio, i, index2, dist2);
            else
                partials[(size_t)i * gridDim.y + blockIdx.y] = b;
        }
    }
}

// merges the slices' partial results of one image-1 feature (ascending slices = ascending indices) and finishes it
__global__ __launch_bounds__(256) void match_finish_k(const MatchBest *__restrict__ partials, unsigned int slices,
                                                      const int *__restrict__ norms1, unsigned int n1, unsigned int n2,
                                                      double ratio, int *__restrict__ index2,
                                                      unsigned int *__restrict__ dist2)
{
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n1)
        return;
    MatchBest b = partials[(size_t)i * slices];
    for (unsigned int sl = 1; sl < slices; sl++)
        b = match_merge(b, partials[(size_t)i * slices + sl]);
    match_finish(b, norms1[i], n2, ratio, i, index2, dist2);
}

void launch_match_prep(const void *feats, int mode, unsigned int n, unsigned int n_pad, signed char *rows, int *norms,
                       unsigned int *bad, hipStream_t s)
{
    if (n_pad == 0)
        return;
    const unsigned int blocks = (n_pad + 7) / 8;
    if (mode == 0)
        match_prep_k<0><<<blocks, 256, 0, s>>>((const unsigned char *)feats, n, n_pad, rows, norms, bad);
    else
        match_prep_k<1><<<blocks, 256, 0, s>>>((const unsigned char *)feats, n, n_pad, rows, norms, bad);
}

// Slices of image 2 a launch is cut into, so that it has enough workgroups for the chip (one workgroup covers
// 32 * kMatchGroups features of image 1 and, unsliced, all of image 2).
unsigned int match_slices(unsigned int n1, unsigned int n2)
{
    const unsigned int wgs = ((n1 + 31) / 32 + kMatchGroups - 1) / kMatchGroups;
    const unsigned int blocks2 = (n2 + 31) / 32;
    unsigned int s = wgs ? (768 + wgs - 1) / wgs : 1;          // about three workgroups per compute unit
    const unsigned int most = blocks2 / (2 * kMatchWaves);      // at least two rounds of blocks per wave and slice
    s = s < most ? s : most;
    return s < 1 ? 1 : (s > kMatchMaxSlices ? kMatchMaxSlices : s);
}

void launch_match(const signed char *rows1, const int *norms1, unsigned int n1, const signed char *rows2,
                  const int *norms2, unsigned int n2, double ratio, int *index2, unsigned int *dist2, void *partials,
                  unsigned int slices, hipStream_t s)
{
    if (n1 == 0)
        return;
    const unsigned int n2_pad = (n2 + 31) / 32 * 32;
    const unsigned int n1_pad = (n1 + 31) / 32 * 32;
    const unsigned int groups = n1_pad / 32;
    if (slices < 1 || !partials)
        slices = 1;
    unsigned int bps = (n2_pad / 32 + slices - 1) / slices;
    bps = (bps + kMatchWaves - 1) / kMatchWaves * kMatchWaves;
    if (bps == 0)
        bps = kMatchWaves;
    dim3 g((groups + kMatchGroups - 1) / kMatchGroups, slices);
    match_k<<<g, 64 * kMatchWaves, 0, s>>>(rows1, norms1, n1, n1_pad, rows2, norms2, n2, n2_pad, ratio, index2, dist2,
                                           (MatchBest *)partials, bps);
    if (slices > 1)
        match_finish_k<<<(n1 + 255) / 256, 256, 0, s>>>((const MatchBest *)partials, slices, norms1, n1, n2, ratio,
                                                        index2, dist2);
}

} // namespace sg
