// kernels_detect.hip -- fused DoG + local-extrema scan with wavefront-ballot compaction, ordered
// compaction primitives (scan, expand), and sub-pixel refinement + contrast + edge tests.  gfx950 only.
//
// Ordering: the scan writes one 64-bit ballot word per (layer, row, 64-pixel segment); words are laid
// out in the reference's raster order (octave, layer, y, x), so expanding the set bits in word order
// reproduces the emission order of the reference's nested loops (sift/sift.cpp:925-938) without a sort.
#include "sg_internal.h"
#include "sg_dd.h"
#include "sg_lds_dma.h"
#include <stdlib.h>

#ifndef SG_DET_RS
#define SG_DET_RS 16
#endif
#ifndef SG_DET_D
#define SG_DET_D 3
#endif
#ifndef SG_DET_XCD
#define SG_DET_XCD 1
#endif

namespace sg {

// DoG level accessor: either a stored DoG pyramid or the difference of two Gaussian levels computed
// in the pyramid's own type (exactly what a stored DoG level of that type would hold).
template <typename T, bool FROM_DOG>
struct DogAt {
    __device__ static __forceinline__ T at(const OctaveTab &tab, int o, int l, size_t off)
    {
        if (FROM_DOG)
            return ((const T *)tab.lv[o][l])[off];
        return ((const T *)tab.lv[o][l + 1])[off] - ((const T *)tab.lv[o][l])[off];
    }
};

// max / min of finite pyramid differences, one instruction each.  (Written with the fmax / fmin builtins the compiler
// first canonicalises every operand it cannot prove to be a quiet number -- a v_max_f64 x, x per loaded value and per
// loop-carried extreme, half as many instructions again as the scan itself needs.)
__device__ __forceinline__ double vmax(double a, double b)
{
    double d;
    asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ double vmin(double a, double b)
{
    double d;
    asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float vmax(float a, float b)
{
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float vmin(float a, float b)
{
    float d;
    asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// Value of the neighbouring lane: UP = true gives lane i the value of lane i-1 (lane 0 gets `edge`), UP = false the
// value of lane i+1 (lane 63 gets `edge`).  One DPP move per dword.
template <bool UP>
__device__ __forceinline__ int wave_shift_dword(int v, int edge)
{
    return UP ? __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xf, 0xf, false)  // wave_shr:1
              : __builtin_amdgcn_update_dpp(edge, v, 0x130, 0xf, 0xf, false); // wave_shl:1
}
template <bool UP>
__device__ __forceinline__ double wave_shift(double v, double edge)
{
    return __hiloint2double(wave_shift_dword<UP>(__double2hiint(v), __double2hiint(edge)),
                            wave_shift_dword<UP>(__double2loint(v), __double2loint(edge)));
}
template <bool UP>
__device__ __forceinline__ float wave_shift(float v, float edge)
{
    return __int_as_float(wave_shift_dword<UP>(__float_as_int(v), __float_as_int(edge)));
}

// ---- fused DoG + extrema scan, rows loaded into registers ------------------------------------------------
// The form for planes the staged kernel below does not take (level rows not 16-byte aligned, or a plane of 2 GB or
// more); SIFTGPU_DETECT_DIRECT=1 in the environment selects it everywhere.
// One wave owns a 64-pixel segment of RS consecutive rows and all three scanned layers.  Each lane
// keeps, for the 5 DoG levels, the extremes of a 3-row x 3-column window as running column/row partial
// maxima and minima and slides them down the rows.
// isExtremum (reference sift/sift.cpp:473-526): val > 0 -> no neighbour (27 incl. centre) greater;
// otherwise no neighbour smaller.  Pre-threshold |val| > thresh is strict (sift.cpp:945).
template <typename T, bool FROM_DOG, bool NOINT, int RS>
__global__ __launch_bounds__(256) void detect_direct_k(OctaveTab tab, int o, unsigned long long *words,
                                                       size_t words_img, unsigned int *chunk_sums, size_t chunks_img,
                                                       unsigned long long *counters, double thresh)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rows = tab.rows[o], cols = tab.cols[o], pitch = tab.pitch[o];
    const int b = blockIdx.z;
    // the four waves of a workgroup scan four neighbouring 64-pixel segments of the same rows, so that the lines a
    // wave shares with its neighbour segment are streamed at about the same time
    const int k = blockIdx.x * 4 + wave;
    const int x = kBorder + 64 * k + lane;
    const bool x_ok = x < cols - kBorder;
    const int xc = min(x, cols - 1); // column for loads (the border columns are real pixels); results masked by x_ok
    const int y_begin = kBorder + blockIdx.y * RS;
    const int y_end = k < tab.wpr[o] ? min(y_begin + RS, rows - kBorder) : y_begin; // no rows for a segment past the row end
    __shared__ unsigned int blk_ctr[3];
    if (threadIdx.x < 3)
        blk_ctr[threadIdx.x] = 0;
    __syncthreads();
    const size_t img = (size_t)b * tab.img_stride[o];
    const int nrows = tab.nrows[o], wpr = tab.wpr[o];
    unsigned long long *wbase = words + (size_t)b * words_img + tab.wbase[o];
    unsigned int *csum = chunk_sums + (size_t)b * chunks_img;

    // Per DoG level the lane keeps the maximum and minimum over columns x-1..x+1 of the current row (rmx1, rmn1)
    // and of the current and previous rows together (rmx01, rmn01); with the next row's column extremes the
    // 3x3 extremes of a level cost two operations each per row.  cen1 / cen2: the centre values of the three
    // scanned layers in the current and next row.
    T rmx01[kDogLevels], rmn01[kDogLevels], rmx1[kDogLevels], rmn1[kDogLevels], rmx2[kDogLevels], rmn2[kDogLevels];
    T cen1[3], cen2[3];
    // A lane loads its own column only; the DoG values of columns x-1 and x+1 come from the neighbouring lanes by
    // wave shifts, and lanes 0 and 63 load the one column beyond the segment that no lane holds.
    const int xe = lane == 0 ? xc - 1 : min(xc + 1, cols - 1); // used by lanes 0 and 63
    const bool edge_lane = lane == 0 || lane == 63;
    auto load_row = [&](int y) { // fills rmx2 / rmn2 / cen2 from row y
        const size_t row = img + (size_t)y * pitch;
        T dc[kDogLevels], de[kDogLevels]; // DoG of the lane's column; of the edge column (lanes 0, 63)
        if (FROM_DOG) {
#pragma unroll
            for (int l = 0; l < kDogLevels; l++) {
                const T *p = (const T *)tab.lv[o][l] + row;
                dc[l] = p[xc];
                de[l] = edge_lane ? p[xe] : (T)0;
            }
        } else {
            T gc[kLevels], ge[kLevels];
#pragma unroll
            for (int l = 0; l < kLevels; l++) {
                const T *p = (const T *)tab.lv[o][l] + row;
                gc[l] = p[xc];
                ge[l] = edge_lane ? p[xe] : (T)0;
            }
#pragma unroll
            for (int l = 0; l < kDogLevels; l++) {
                dc[l] = gc[l + 1] - gc[l];
                de[l] = ge[l + 1] - ge[l];
            }
        }
#pragma unroll
        for (int l = 0; l < kDogLevels; l++) {
            const T left = wave_shift<true>(dc[l], de[l]);   // column x-1: lane-1's value, lane 0 keeps its edge value
            const T right = wave_shift<false>(dc[l], de[l]); // column x+1: lane+1's value, lane 63 keeps its edge value
            rmx2[l] = vmax(vmax(left, dc[l]), right);
            rmn2[l] = vmin(vmin(left, dc[l]), right);
        }
#pragma unroll
        for (int l = 0; l < 3; l++)
            cen2[l] = dc[l + 1];
    };

    unsigned int n00 = 0, n0 = 0, n1 = 0;
    if (y_begin < y_end) {
        load_row(y_begin - 1);
#pragma unroll
        for (int l = 0; l < kDogLevels; l++) {
            rmx1[l] = rmx2[l];
            rmn1[l] = rmn2[l];
        }
        load_row(y_begin);
#pragma unroll
        for (int l = 0; l < kDogLevels; l++) {
            rmx01[l] = vmax(rmx1[l], rmx2[l]);
            rmn01[l] = vmin(rmn1[l], rmn2[l]);
            rmx1[l] = rmx2[l];
            rmn1[l] = rmn2[l];
        }
#pragma unroll
        for (int l = 0; l < 3; l++)
            cen1[l] = cen2[l];
    }
    for (int y = y_begin; y < y_end; y++) {
        load_row(y + 1);
        // per-level 3x3 max and min
        T mx[kDogLevels], mn[kDogLevels];
#pragma unroll
        for (int l = 0; l < kDogLevels; l++) {
            mx[l] = vmax(rmx01[l], rmx2[l]);
            mn[l] = vmin(rmn01[l], rmn2[l]);
        }
        n00 += __popcll(__ballot(x_ok));
#pragma unroll
        for (int layer = 1; layer <= 3; layer++) {
            T val = cen1[layer - 1];
            bool pre = x_ok && (fabs((double)val) > thresh);
            T hi = mx[layer], lo = mn[layer];
            if (!NOINT) {
                hi = vmax(vmax(hi, mx[layer - 1]), mx[layer + 1]);
                lo = vmin(vmin(lo, mn[layer - 1]), mn[layer + 1]);
            }
            bool ext = pre && (val > (T)0 ? !(val < hi) : !(val > lo));
            unsigned long long mpre = __ballot(pre);
            unsigned long long mext = __ballot(ext);
            n0 += __popcll(mpre);
            n1 += __popcll(mext);
            if (lane == 0) {
                unsigned int widx = ((unsigned int)(layer - 1) * nrows + (y - kBorder)) * wpr + k;
                wbase[widx] = mext;
                if (mext)
                    atomicAdd(&csum[(tab.wbase[o] + widx) / kChunkWords], (unsigned int)__popcll(mext));
            }
        }
        // slide down one row
#pragma unroll
        for (int l = 0; l < kDogLevels; l++) {
            rmx01[l] = vmax(rmx1[l], rmx2[l]);
            rmn01[l] = vmin(rmn1[l], rmn2[l]);
            rmx1[l] = rmx2[l];
            rmn1[l] = rmn2[l];
        }
#pragma unroll
        for (int l = 0; l < 3; l++)
            cen1[l] = cen2[l];
    }
    // counters: one LDS add per wave, then one global add per block and counter
    if (lane == 0 && n00) {
        atomicAdd(&blk_ctr[0], n00 * 3u); // one count per layer
        atomicAdd(&blk_ctr[1], n0);
        atomicAdd(&blk_ctr[2], n1);
    }
    __syncthreads();
    if (threadIdx.x < 3 && blk_ctr[threadIdx.x])
        atomicAdd(&counters[(size_t)b * 8 + threadIdx.x], (unsigned long long)blk_ctr[threadIdx.x]);
}

// ---- the same scan with its rows staged through LDS by the memory pipeline (the form normally used) --------
// detect_direct_k keeps one row of each level in flight per wave and no more: loaded values need registers.  Here every
// wave owns a ring of D row slots in LDS and fills it with loads that write LDS directly (global_load_lds_dwordx4:
// no register destination, lane i of the instruction writes 16 bytes at the slot's base + 16 i), D - 1 rows ahead of
// the row it is working on, so that the bytes a compute unit has in flight no longer depend on its register file.
// A row slot holds, per level, the 66 columns x0-1 .. x0+64 of the wave's 64-pixel segment (x0 - 1 = 4 + 64 k is a
// multiple of four, so every 16-byte piece is aligned); a lane reads its three columns from there.
// Ordering: a wave reads a slot only after its own s_waitcnt vmcnt(n) has retired the loads that fill it (loads
// retire in order; stores in between only make the count conservative), and refills a slot only after the
// s_waitcnt lgkmcnt(0) that returned its reads.  No other wave touches the ring.
template <typename T, bool FROM_DOG, int D>
struct DetectRing {
    static constexpr int NL = FROM_DOG ? kDogLevels : kLevels; // planes read per row
    static constexpr int EPC = 16 / (int)sizeof(T);            // elements per 16-byte piece
    static constexpr int NCH = (66 + EPC - 1) / EPC;           // pieces per level and row
    static constexpr int LSTRIDE = NCH * 16;
    static constexpr int ROWSLOT = NL * LSTRIDE;
    static constexpr int WAVE_BYTES = D * ROWSLOT;
};

// One launch scans a run of octaves [o_begin, o_end): tiles are numbered octave after octave, first[o] is the first
// tile of octave o (first[o_end] = the launch's tile count).  The small octaves, a few tiles each, then cost no launch
// of their own: a frame processed alone spends its time on dependent launches, not on work.
struct DetectTiles {
    unsigned int first[kMaxOctaves + 1];
    int o_begin, o_end;
};

template <typename T, bool FROM_DOG, bool NOINT, int RS, int D>
__global__ __launch_bounds__(256) void detect_k(OctaveTab tab, DetectTiles dt, unsigned long long *words, size_t words_img,
                                                unsigned int *chunk_sums, size_t chunks_img,
                                                unsigned long long *counters, double thresh, unsigned int n_img)
{
    using R = DetectRing<T, FROM_DOG, D>;
    constexpr int NL = R::NL;
    __shared__ __align__(16) unsigned char ring[4 * R::WAVE_BYTES];
    __shared__ unsigned int blk_ctr[3];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // Workgroups are handed to the eight XCDs in turn, and each XCD has its own L2.  The launch is one-dimensional
    // and workgroup i works on tile (i % 8) * (n / 8) + i / 8, so that an XCD walks a contiguous run of tiles: the
    // rows two vertically adjacent tiles share are then read by the same XCD, not long apart.
    const unsigned int per_xcd = gridDim.x >> 3;
    unsigned int tile = SG_DET_XCD ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    if (tile >= dt.first[dt.o_end])
        return;
    int o = dt.o_begin;
    while (o + 1 < dt.o_end && tile >= dt.first[o + 1]) // uniform over the workgroup
        o++;
    tile -= dt.first[o];
    const int rows = tab.rows[o], cols = tab.cols[o], pitch = tab.pitch[o];
    const int nrows = tab.nrows[o], wpr = tab.wpr[o];
    const unsigned int gx = (wpr + 3) / 4, gy = (nrows + RS - 1) / RS;
    (void)n_img;
    const int b = tile / (gx * gy);
    const unsigned int in_img = tile - b * (gx * gy);
    const int by = in_img / gx, bx = in_img - by * gx;
    const int k = bx * 4 + wave;
    const int x = kBorder + 64 * k + lane;
    const bool x_ok = x < cols - kBorder;
    const int y_begin = kBorder + by * RS;
    const int y_end = k < wpr ? min(y_begin + RS, rows - kBorder) : y_begin;
    if (threadIdx.x < 3)
        blk_ctr[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long *wbase = words + (size_t)b * words_img + tab.wbase[o];
    unsigned int *csum = chunk_sums + (size_t)b * chunks_img;

    unsigned int n00 = 0, n0 = 0, n1 = 0;
    if (y_begin < y_end) {
        const int n_stage = y_end - y_begin + 2; // rows y_begin-1 .. y_end
        unsigned long long base[NL];
#pragma unroll
        for (int l = 0; l < NL; l++)
            base[l] = (unsigned long long)tab.lv[o][l] + ((unsigned long long)b * tab.img_stride[o]) * sizeof(T);
        // A piece that would run past the row end starts at the last whole piece instead: it then holds no column the
        // scan needs (those end at cols - kBorder), and the read stays inside the row.
        const int c0 = min(kBorder - 1 + 64 * k + R::EPC * lane, cols - R::EPC);
        unsigned int voff = ((unsigned int)(y_begin - 1) * (unsigned int)pitch + (unsigned int)c0) * (unsigned int)sizeof(T);
        const unsigned int vstep = (unsigned int)pitch * (unsigned int)sizeof(T);
        const unsigned int lds_wave = (unsigned int)(uintptr_t)ring + (unsigned int)wave * R::WAVE_BYTES;
        const unsigned char *ring_wave = ring + wave * R::WAVE_BYTES;
        auto stage_row = [&](int slot) { // next row of all levels into the slot
            if (lane < R::NCH) {
#pragma unroll
                for (int l = 0; l < NL; l++)
                    load16_to_lds(lds_wave + slot * R::ROWSLOT + l * R::LSTRIDE, base[l], voff);
            }
            voff += vstep;
        };

        T rmx01[kDogLevels], rmn01[kDogLevels], rmx1[kDogLevels], rmn1[kDogLevels], rmx2[kDogLevels], rmn2[kDogLevels];
        T cen1[3], cen2[3];
        int staged = 0, taken = 0, slot = 0;
        // takes the next staged row: fills rmx2 / rmn2 / cen2 from it, then restages its slot
        auto take_row = [&]() {
            wait_groups_ahead<NL, D - 1>(staged - taken - 1);
            const T *p = reinterpret_cast<const T *>(ring_wave + slot * R::ROWSLOT) + lane;
            T g[NL][3];
#pragma unroll
            for (int l = 0; l < NL; l++) {
#pragma unroll
                for (int q = 0; q < 3; q++)
                    g[l][q] = p[l * (R::LSTRIDE / (int)sizeof(T)) + q];
            }
            wait_lgkm0(); // the slot's reads have returned: it may be refilled
            taken++;
            if (staged < n_stage) {
                stage_row(slot);
                staged++;
            }
            slot = slot + 1 == D ? 0 : slot + 1;
#pragma unroll
            for (int l = 0; l < kDogLevels; l++) {
                T d[3];
#pragma unroll
                for (int q = 0; q < 3; q++)
                    d[q] = FROM_DOG ? g[l][q] : g[l + 1 < NL ? l + 1 : l][q] - g[l][q];
                rmx2[l] = vmax(vmax(d[0], d[1]), d[2]);
                rmn2[l] = vmin(vmin(d[0], d[1]), d[2]);
                if (l >= 1 && l <= 3)
                    cen2[l - 1] = d[1];
            }
        };
        {
            int fill_slot = 0;
            for (; staged < min(D, n_stage); staged++)
                stage_row(fill_slot++);
        }
        take_row(); // row y_begin - 1
#pragma unroll
        for (int l = 0; l < kDogLevels; l++) {
            rmx1[l] = rmx2[l];
            rmn1[l] = rmn2[l];
        }
        take_row(); // row y_begin
#pragma unroll
        for (int l = 0; l < kDogLevels; l++) {
            rmx01[l] = vmax(rmx1[l], rmx2[l]);
            rmn01[l] = vmin(rmn1[l], rmn2[l]);
            rmx1[l] = rmx2[l];
            rmn1[l] = rmn2[l];
        }
#pragma unroll
        for (int l = 0; l < 3; l++)
            cen1[l] = cen2[l];
        for (int y = y_begin; y < y_end; y++) {
            take_row(); // row y + 1
            T mx[kDogLevels], mn[kDogLevels];
#pragma unroll
            for (int l = 0; l < kDogLevels; l++) {
                mx[l] = vmax(rmx01[l], rmx2[l]);
                mn[l] = vmin(rmn01[l], rmn2[l]);
            }
            n00 += __popcll(__ballot(x_ok));
#pragma unroll
            for (int layer = 1; layer <= 3; layer++) {
                T val = cen1[layer - 1];
                bool pre = x_ok && (fabs((double)val) > thresh);
                T hi = mx[layer], lo = mn[layer];
                if (!NOINT) {
                    hi = vmax(vmax(hi, mx[layer - 1]), mx[layer + 1]);
                    lo = vmin(vmin(lo, mn[layer - 1]), mn[layer + 1]);
                }
                bool ext = pre && (val > (T)0 ? !(val < hi) : !(val > lo));
                unsigned long long mpre = __ballot(pre);
                unsigned long long mext = __ballot(ext);
                n0 += __popcll(mpre);
                n1 += __popcll(mext);
                if (lane == 0) {
                    unsigned int widx = ((unsigned int)(layer - 1) * nrows + (y - kBorder)) * wpr + k;
                    wbase[widx] = mext;
                    if (mext)
                        atomicAdd(&csum[(tab.wbase[o] + widx) / kChunkWords], (unsigned int)__popcll(mext));
                }
            }
#pragma unroll
            for (int l = 0; l < kDogLevels; l++) {
                rmx01[l] = vmax(rmx1[l], rmx2[l]);
                rmn01[l] = vmin(rmn1[l], rmn2[l]);
                rmx1[l] = rmx2[l];
                rmn1[l] = rmn2[l];
            }
#pragma unroll
            for (int l = 0; l < 3; l++)
                cen1[l] = cen2[l];
        }
    }
    if (lane == 0 && n00) {
        atomicAdd(&blk_ctr[0], n00 * 3u);
        atomicAdd(&blk_ctr[1], n0);
        atomicAdd(&blk_ctr[2], n1);
    }
    __syncthreads();
    if (threadIdx.x < 3 && blk_ctr[threadIdx.x])
        atomicAdd(&counters[(size_t)b * 8 + threadIdx.x], (unsigned long long)blk_ctr[threadIdx.x]);
}

// true when the staged scan applies: 16-byte aligned level rows and row offsets that fit 32 bits
static bool detect_staged_ok(const OctaveTab &tab, int o, int prec)
{
    const size_t esz = prec == SG_F64 ? 8 : 4;
    const int nl = tab.from_dog ? kDogLevels : kLevels;
    if (env().detect_direct)
        return false;
    if ((size_t)tab.pitch[o] * esz % 16 || (size_t)tab.img_stride[o] * esz % 16)
        return false;
    if ((size_t)tab.rows[o] * tab.pitch[o] * esz >= (1ull << 31) || tab.cols[o] < 16)
        return false;
    for (int l = 0; l < nl; l++)
        if ((uintptr_t)tab.lv[o][l] % 16)
            return false;
    return true;
}

static void launch_detect_direct(const OctaveTab &tab, int o, int B, unsigned long long *words, size_t words_img_stride,
                                 unsigned int *chunk_sums, size_t chunks_img_stride, unsigned long long *counters,
                                 double thresh, bool fd, bool ni, int prec, hipStream_t s)
{
    constexpr int RS = 16;
    dim3 g((tab.wpr[o] + 3) / 4, (tab.nrows[o] + RS - 1) / RS, B);
#define SG_DET(T, FD, NI)                                                                                      \
    detect_direct_k<T, FD, NI, RS><<<g, 256, 0, s>>>(tab, o, words, words_img_stride, chunk_sums, chunks_img_stride, \
                                              counters, thresh)
    if (prec == SG_F64) {
        if (fd) { if (ni) SG_DET(double, true, true); else SG_DET(double, true, false); }
        else { if (ni) SG_DET(double, false, true); else SG_DET(double, false, false); }
    } else {
        if (fd) { if (ni) SG_DET(float, true, true); else SG_DET(float, true, false); }
        else { if (ni) SG_DET(float, false, true); else SG_DET(float, false, false); }
    }
#undef SG_DET
}

// Scans the octaves [o_from, o_to) of the table (all of them by default) that have pixels inside the border.  Consecutive octaves the staged kernel can take
// share one launch; an octave it cannot take (rows not 16-byte aligned, planes of 2 GB and more) gets the direct kernel.
// Returns the number of launches.
int launch_detect(const OctaveTab &tab, int B, unsigned long long *words, size_t words_img_stride,
                  unsigned int *chunk_sums, size_t chunks_img_stride, unsigned long long *counters,
                  double thresh, int nointerval, int prec, hipStream_t s, int o_from, int o_to)
{
    const bool fd = tab.from_dog != 0, ni = nointerval != 0;
    constexpr int RS = SG_DET_RS, D = SG_DET_D;
    int launches = 0;
    int o = o_from < 0 ? 0 : o_from;
    const int o_stop = o_to < 0 || o_to > tab.octaves ? tab.octaves : o_to;
    while (o < o_stop) {
        if (tab.nrows[o] <= 0 || tab.wpr[o] <= 0) {
            o++;
            continue;
        }
        if (!detect_staged_ok(tab, o, prec)) {
            launch_detect_direct(tab, o, B, words, words_img_stride, chunk_sums, chunks_img_stride, counters, thresh, fd,
                                 ni, prec, s);
            launches++;
            o++;
            continue;
        }
        DetectTiles dt;
        for (unsigned int &f : dt.first)
            f = 0;
        dt.o_begin = o;
        unsigned long long tiles = 0;
        while (o < o_stop && tab.nrows[o] > 0 && tab.wpr[o] > 0 && detect_staged_ok(tab, o, prec)) {
            const unsigned long long t = (unsigned long long)((tab.wpr[o] + 3) / 4) * ((tab.nrows[o] + RS - 1) / RS) * B;
            if (tiles + t > 0xfffffff0ull)
                break; // (never with this library's own allocations) the next launch takes it
            dt.first[o] = (unsigned int)tiles;
            tiles += t;
            o++;
        }
        dt.o_end = o;
        dt.first[o] = (unsigned int)tiles;
        const unsigned int g = (unsigned int)((tiles + 7) / 8 * 8);
#define SG_DET(T, FD, NI)                                                                                              \
    detect_k<T, FD, NI, RS, D><<<g, 256, 0, s>>>(tab, dt, words, words_img_stride, chunk_sums, chunks_img_stride,      \
                                                  counters, thresh, (unsigned int)B)
        if (prec == SG_F64) {
            if (fd) { if (ni) SG_DET(double, true, true); else SG_DET(double, true, false); }
            else { if (ni) SG_DET(double, false, true); else SG_DET(double, false, false); }
        } else {
            if (fd) { if (ni) SG_DET(float, true, true); else SG_DET(float, true, false); }
            else { if (ni) SG_DET(float, false, true); else SG_DET(float, false, false); }
        }
#undef SG_DET
        launches++;
    }
    return launches;
}

// ---- ordered compaction: exclusive scan of per-chunk popcounts, then bit expansion ---------------------

__device__ __forceinline__ unsigned int wave_incl_scan(unsigned int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned int t = __shfl_up(v, d);
        if (lane >= d)
            v += t;
    }
    return v;
}

// One block per image: in-place exclusive scan of n values, total to totals[b].
__global__ __launch_bounds__(1024) void scan_k(unsigned int *sums, size_t img_stride, int n, unsigned int *totals,
                                               unsigned int cap, unsigned int *overflow)
{
    __shared__ unsigned int wsum[16];
    __shared__ unsigned int carry_s;
    unsigned int *a = sums + (size_t)blockIdx.x * img_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        carry_s = 0;
    __syncthreads();
    if (n >= 4096 && n % 4096 == 0 && n <= 64 * 1024) {
        // Long inputs (the 16 384 scale buckets of the ordering step): every thread takes n / 1024 consecutive values,
        // so the block synchronises once instead of four times per 1024 values.
        const int per = n >> 10; // 4 .. 64, a multiple of 4
        unsigned int *mine = a + (size_t)threadIdx.x * per;
        unsigned int sum = 0;
        for (int k = 0; k < per; k += 4) {
            const uint4 v = *reinterpret_cast<const uint4 *>(mine + k);
            sum += v.x + v.y + v.z + v.w;
        }
        const unsigned int inc = wave_incl_scan(sum, lane);
        if (lane == 63)
            wsum[wave] = inc;
        __syncthreads();
        unsigned int run = inc - sum;
        for (int w = 0; w < wave; w++)
            run += wsum[w];
        for (int k = 0; k < per; k += 4) {
            uint4 v = *reinterpret_cast<const uint4 *>(mine + k);
            uint4 o;
            o.x = run;
            o.y = o.x + v.x;
            o.z = o.y + v.y;
            o.w = o.z + v.z;
            run = o.w + v.w;
            *reinterpret_cast<uint4 *>(mine + k) = o;
        }
        if (threadIdx.x == 1023) {
            totals[blockIdx.x] = run < cap ? run : cap;
            if (run > cap)
                atomicMax(overflow, run);
        }
        return;
    }
    for (int base = 0; base < n; base += 1024) {
        int i = base + threadIdx.x;
        unsigned int v = i < n ? a[i] : 0;
        unsigned int inc = wave_incl_scan(v, lane);
        if (lane == 63)
            wsum[wave] = inc;
        __syncthreads();
        unsigned int woff = 0;
        for (int w = 0; w < wave; w++)
            woff += wsum[w];
        unsigned int carry = carry_s;
        if (i < n)
            a[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023)
            carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        unsigned int t = carry_s;
        totals[blockIdx.x] = t < cap ? t : cap;
        if (t > cap)
            atomicMax(overflow, t);
    }
}

void launch_scan(unsigned int *chunk_sums, size_t chunks_img_stride, int n_chunks, unsigned int *totals,
                 unsigned int cap, unsigned int *overflow, int B, hipStream_t s)
{
    scan_k<<<B, 1024, 0, s>>>(chunk_sums, chunks_img_stride, n_chunks, totals, cap, overflow);
}

// One block per chunk of kChunkWords words (256 threads x 4 consecutive words).
// dyn_items (optional): per-image item count, items_per_word items to a word; words at or beyond
// ceil(items/items_per_word) hold stale bits from earlier calls and are ignored.
__global__ __launch_bounds__(256) void expand_k(const unsigned long long *words, size_t words_img, unsigned int n_words,
                                                const unsigned int *dyn_items, unsigned int items_per_word,
                                                const unsigned int *chunk_offs,
                                                size_t chunks_img, unsigned int *out, size_t out_img, unsigned int cap)
{
    __shared__ unsigned int wsum[4];
    const int b = blockIdx.y;
    const unsigned int chunk = blockIdx.x;
    const unsigned long long *w = words + (size_t)b * words_img;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (dyn_items) {
        unsigned int nw = (dyn_items[b] + items_per_word - 1u) / items_per_word;
        n_words = nw < n_words ? nw : n_words;
    }
    if (chunk * kChunkWords >= n_words)
        return;
    unsigned int w0 = chunk * kChunkWords + threadIdx.x * 4;
    unsigned long long m[4];
    unsigned int cnt = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        m[q] = (w0 + q < n_words) ? w[w0 + q] : 0ull;
        cnt += __popcll(m[q]);
    }
    unsigned int inc = wave_incl_scan(cnt, lane);
    if (lane == 63)
        wsum[wave] = inc;
    __syncthreads();
    unsigned int pos = chunk_offs[(size_t)b * chunks_img + chunk] + inc - cnt;
    for (int q = 0; q < wave; q++)
        pos += wsum[q];
    unsigned int *o = out + (size_t)b * out_img;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        unsigned long long mm = m[q];
        while (mm) {
            int bit = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            if (pos < cap)
                o[pos] = (w0 + q) * 64u + (unsigned int)bit;
            pos++;
        }
    }
}

// The same expansion for a bitmap in which most bits are set (the survivors of the refinement: one bit per candidate,
// nearly half of them set).  There a short list sits in a few hundred words, which expand_k would hand to one or two
// workgroups whose threads each walk their own words bit by bit (a frame alone: 20 us in a single workgroup).  Here a
// wave takes 64 consecutive words of a chunk, and the lanes are the bits: word k's set bits are stored by the lanes that
// hold them, at the word's offset plus the number of set bits below the lane's.  Same output order.
__global__ __launch_bounds__(64) void expand_dense_k(const unsigned long long *words, size_t words_img, unsigned int n_words,
                                                     const unsigned int *dyn_items, unsigned int items_per_word,
                                                     const unsigned int *chunk_offs, size_t chunks_img, unsigned int *out,
                                                     size_t out_img, unsigned int cap)
{
    const int b = blockIdx.y;
    const unsigned int chunk = blockIdx.x / (kChunkWords / 64), part = blockIdx.x % (kChunkWords / 64);
    const unsigned long long *w = words + (size_t)b * words_img;
    const unsigned int lane = threadIdx.x;
    if (dyn_items) {
        unsigned int nw = (dyn_items[b] + items_per_word - 1u) / items_per_word;
        n_words = nw < n_words ? nw : n_words;
    }
    const unsigned int w_first = chunk * kChunkWords + part * 64u;
    if (w_first >= n_words)
        return;
    // set bits of the chunk's words in front of this wave's
    unsigned int before = 0;
    for (unsigned int g = 0; g < part; g++)
        before += (unsigned int)__popcll(w[chunk * kChunkWords + g * 64u + lane]); // (all below w_first < n_words)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        before += __shfl_xor(before, d);
    const unsigned long long mine = w_first + lane < n_words ? w[w_first + lane] : 0ull;
    const unsigned int cnt = (unsigned int)__popcll(mine);
    const unsigned int start = chunk_offs[(size_t)b * chunks_img + chunk] + before + wave_incl_scan(cnt, (int)lane) - cnt;
    unsigned int *o = out + (size_t)b * out_img;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int k = 0; k < 64; k++) {
        const unsigned long long m = __shfl(mine, k);
        if (m == 0)
            continue; // wave-uniform
        const unsigned int p = __shfl(start, k) + (unsigned int)__popcll(m & below);
        if (((m >> lane) & 1ull) && p < cap)
            o[p] = (w_first + (unsigned int)k) * 64u + lane;
    }
}

void launch_expand(const unsigned long long *words, size_t words_img_stride, unsigned int n_words,
                   const unsigned int *dyn_items, unsigned int items_per_word, const unsigned int *chunk_offs,
                   size_t chunks_img_stride, unsigned int *out,
                   size_t out_img_stride, unsigned int cap, int B, hipStream_t s, bool dense)
{
    if (n_words == 0)
        return;
    const unsigned int chunks = (n_words + kChunkWords - 1) / kChunkWords;
    if (dense) {
        dim3 g(chunks * (kChunkWords / 64), B);
        expand_dense_k<<<g, 64, 0, s>>>(words, words_img_stride, n_words, dyn_items, items_per_word, chunk_offs,
                                        chunks_img_stride, out, out_img_stride, cap);
        return;
    }
    dim3 g(chunks, B);
    expand_k<<<g, 256, 0, s>>>(words, words_img_stride, n_words, dyn_items, items_per_word, chunk_offs, chunks_img_stride, out, out_img_stride, cap);
}

// Per-chunk popcounts of a bitmap whose producer does not count (orient_k: one word per keypoint).  The producer's own
// count would be one atomic add per keypoint on a handful of words -- all keypoints of a chunk of 1024 add to the
// same word -- and adds to one word are served one after the other by the memory side (about 90 per microsecond): a
// frame alone, whose 18 000 keypoints share 18 words, spent two thirds of the orientation kernel's time in that queue.
// Words at or beyond ceil(items / items_per_word) hold stale bits and are not counted (as in expand_k).
__global__ __launch_bounds__(256) void chunk_popc_k(const unsigned long long *words, size_t words_img, unsigned int n_words,
                                                    const unsigned int *dyn_items, unsigned int items_per_word,
                                                    unsigned int *chunk_sums, size_t chunks_img)
{
    __shared__ unsigned int wsum[4];
    const int b = blockIdx.y;
    const unsigned int chunk = blockIdx.x;
    const unsigned long long *w = words + (size_t)b * words_img;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (dyn_items) {
        unsigned int nw = (dyn_items[b] + items_per_word - 1u) / items_per_word;
        n_words = nw < n_words ? nw : n_words;
    }
    const unsigned int w0 = chunk * kChunkWords + threadIdx.x * 4;
    unsigned int cnt = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (w0 + q < n_words)
            cnt += __popcll(w[w0 + q]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        cnt += __shfl_xor(cnt, d);
    if (lane == 0)
        wsum[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0)
        chunk_sums[(size_t)b * chunks_img + chunk] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

void launch_chunk_popc(const unsigned long long *words, size_t words_img_stride, unsigned int n_words,
                       const unsigned int *dyn_items, unsigned int items_per_word, unsigned int *chunk_sums,
                       size_t chunks_img_stride, int n_chunks, int B, hipStream_t s)
{
    if (n_chunks < 1)
        return;
    dim3 g((unsigned int)n_chunks, B);
    chunk_popc_k<<<g, 256, 0, s>>>(words, words_img_stride, n_words, dyn_items, items_per_word, chunk_sums, chunks_img_stride);
}

// ---- refinement ------------------------------------------------------------------------------------------
// InterploationExtremum, GetOffsetX, Hessian3D, DerivativeOf3D, Inverse3D, GetFabsDx, passEdgeResponse
// (reference sift/sift.cpp:600-910), then CalculateScale and HalfFeatures (sift.cpp:1065-1096).
// One thread per candidate, double arithmetic in the reference's operation order.

template <typename T, bool FROM_DOG>
struct Dog3 {
    const OctaveTab &tab;
    int o;
    size_t img;
    int pitch;
    __device__ __forceinline__ double at(int l, int x, int y) const
    {
        return (double)DogAt<T, FROM_DOG>::at(tab, o, l, img + (size_t)y * pitch + x);
    }
};

__device__ __forceinline__ bool inverse3d(const double *H, double *Hi)
{
#define HA(i, j) (H[(i) * 3 + (j)])
    double A = HA(0, 0) * HA(1, 1) * HA(2, 2) + HA(0, 1) * HA(1, 2) * HA(2, 0) + HA(0, 2) * HA(1, 0) * HA(2, 1) -
               HA(0, 0) * HA(1, 2) * HA(2, 1) - HA(0, 1) * HA(1, 0) * HA(2, 2) - HA(0, 2) * HA(1, 1) * HA(2, 0);
    if (fabs(A) < 1e-10)
        return false;
    Hi[0] = HA(1, 1) * HA(2, 2) - HA(2, 1) * HA(1, 2);
    Hi[1] = -(HA(0, 1) * HA(2, 2) - HA(2, 1) * HA(0, 2));
    Hi[2] = HA(0, 1) * HA(1, 2) - HA(0, 2) * HA(1, 1);
    Hi[3] = HA(1, 2) * HA(2, 0) - HA(2, 2) * HA(1, 0);
    Hi[4] = -(HA(0, 2) * HA(2, 0) - HA(0, 0) * HA(2, 2));
    Hi[5] = HA(0, 2) * HA(1, 0) - HA(0, 0) * HA(1, 2);
    Hi[6] = HA(1, 0) * HA(2, 1) - HA(1, 1) * HA(2, 0);
    Hi[7] = -(HA(0, 0) * HA(2, 1) - HA(0, 1) * HA(2, 0));
    Hi[8] = HA(0, 0) * HA(1, 1) - HA(0, 1) * HA(1, 0);
#pragma unroll
    for (int i = 0; i < 9; i++)
        Hi[i] /= A;
#undef HA
    return true;
}

template <typename D>
__device__ __forceinline__ void derivative3d(const D &g, int l, int x, int y, double *dx)
{
    dx[0] = (g.at(l, x + 1, y) - g.at(l, x - 1, y)) / 2.0;
    dx[1] = (g.at(l, x, y + 1) - g.at(l, x, y - 1)) / 2.0;
    dx[2] = (g.at(l + 1, x, y) - g.at(l - 1, x, y)) / 2.0;
}

template <typename D>
__device__ __forceinline__ void offset3d(const D &g, int l, int x, int y, double *off)
{
    double val = g.at(l, x, y);
    double Dxx = g.at(l, x + 1, y) + g.at(l, x - 1, y) - 2 * val;
    double Dyy = g.at(l, x, y + 1) + g.at(l, x, y - 1) - 2 * val;
    double Dss = g.at(l + 1, x, y) + g.at(l - 1, x, y) - 2 * val;
    double Dxy = (g.at(l, x + 1, y + 1) + g.at(l, x - 1, y - 1) - g.at(l, x + 1, y - 1) - g.at(l, x - 1, y + 1)) / 4.0;
    double Dxs = (g.at(l + 1, x + 1, y) + g.at(l - 1, x - 1, y) - g.at(l - 1, x + 1, y) - g.at(l + 1, x - 1, y)) / 4.0;
    double Dys = (g.at(l + 1, x, y + 1) + g.at(l - 1, x, y - 1) - g.at(l + 1, x, y - 1) - g.at(l - 1, x, y + 1)) / 4.0;
    double H[9] = {Dxx, Dxy, Dxs, Dxy, Dyy, Dys, Dxs, Dys, Dss};
    double Hi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // stays zero when the Hessian is singular (sift.cpp:778-780)
    inverse3d(H, Hi);
    double dx[3];
    derivative3d(g, l, x, y, dx);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < 3; j++)
            a += Hi[i * 3 + j] * dx[j];
        off[i] = -a;
    }
}

// The 3x3x3 block of DoG values around (level l, x, y), held in registers.  Every quantity of a refinement step --
// the Hessian, the gradient, the contrast, and the edge test when the level did not move by two -- reads inside this
// block, so a candidate loads it once per position (36 Gaussian pixels: four levels of 3x3) instead of once per use.
// Values are formed exactly as Dog3::at forms them (difference in the pyramid's type, then to double).
template <typename T, bool FROM_DOG>
struct DogCube {
    double v[3][3][3]; // [level - l + 1][y' - y + 1][x' - x + 1]
    __device__ __forceinline__ void load(const OctaveTab &tab, int o, size_t img, int pitch, int l, int x, int y)
    {
        constexpr int NG = FROM_DOG ? 3 : 4;
        T g[NG][3][3];
#pragma unroll
        for (int li = 0; li < NG; li++) {
            const T *p = (const T *)tab.lv[o][l - 1 + li] + img + (size_t)(y - 1) * pitch + (x - 1);
#pragma unroll
            for (int yi = 0; yi < 3; yi++)
#pragma unroll
                for (int xi = 0; xi < 3; xi++)
                    g[li][yi][xi] = p[yi * pitch + xi];
        }
#pragma unroll
        for (int li = 0; li < 3; li++)
#pragma unroll
            for (int yi = 0; yi < 3; yi++)
#pragma unroll
                for (int xi = 0; xi < 3; xi++)
                    v[li][yi][xi] = FROM_DOG ? (double)g[li][yi][xi] : (double)(T)(g[li + 1 < NG ? li + 1 : li][yi][xi] - g[li][yi][xi]);
    }
    // value at (l + dl, x + dx, y + dy), offsets in -1..1 known at compile time
    __device__ __forceinline__ double at(int dl, int dx, int dy) const { return v[dl + 1][dy + 1][dx + 1]; }
};

// derivative3d / offset3d on the block's centre: the same expressions in the same order
template <typename C>
__device__ __forceinline__ void derivative3d_c(const C &g, double *dx)
{
    dx[0] = (g.at(0, 1, 0) - g.at(0, -1, 0)) / 2.0;
    dx[1] = (g.at(0, 0, 1) - g.at(0, 0, -1)) / 2.0;
    dx[2] = (g.at(1, 0, 0) - g.at(-1, 0, 0)) / 2.0;
}

template <typename C>
__device__ __forceinline__ void offset3d_c(const C &g, double *off)
{
    double val = g.at(0, 0, 0);
    double Dxx = g.at(0, 1, 0) + g.at(0, -1, 0) - 2 * val;
    double Dyy = g.at(0, 0, 1) + g.at(0, 0, -1) - 2 * val;
    double Dss = g.at(1, 0, 0) + g.at(-1, 0, 0) - 2 * val;
    double Dxy = (g.at(0, 1, 1) + g.at(0, -1, -1) - g.at(0, 1, -1) - g.at(0, -1, 1)) / 4.0;
    double Dxs = (g.at(1, 1, 0) + g.at(-1, -1, 0) - g.at(-1, 1, 0) - g.at(1, -1, 0)) / 4.0;
    double Dys = (g.at(1, 0, 1) + g.at(-1, 0, -1) - g.at(1, 0, -1) - g.at(-1, 0, 1)) / 4.0;
    double H[9] = {Dxx, Dxy, Dxs, Dxy, Dyy, Dys, Dxs, Dys, Dss};
    double Hi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // stays zero when the Hessian is singular (sift.cpp:778-780)
    inverse3d(H, Hi);
    double dx[3];
    derivative3d_c(g, dx);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < 3; j++)
            a += Hi[i * 3 + j] * dx[j];
        off[i] = -a;
    }
}

template <typename T, bool FROM_DOG>
__global__ __launch_bounds__(256) void refine_k(OctaveTab tab, const unsigned int *cand, size_t cand_img,
                                                const unsigned int *cand_count, KpRec *kp, size_t kp_img,
                                                unsigned long long *valid_words, size_t vw_img,
                                                unsigned int *chunk_sums, size_t chunks_img,
                                                unsigned long long *counters, double sigma, unsigned int cap)
{
    const int b = blockIdx.y;
    const unsigned int n = min(cand_count[b], cap);
    const int lane = threadIdx.x & 63;
    unsigned int n2 = 0, n3 = 0;
    // every wave handles 64 consecutive candidates per trip so that its ballot is one bitmap word
    for (unsigned int c0 = (blockIdx.x * 256u + (threadIdx.x & ~63u)); c0 < n; c0 += gridDim.x * 256u) {
        unsigned int c = c0 + lane;
        bool ok = false, refined = false;
        KpRec rec;
        if (c < n) {
            unsigned int v = cand[(size_t)b * cand_img + c];
            unsigned int widx = v >> 6;
            int bit = v & 63;
            int o = 0;
            while (o + 1 < tab.octaves && widx >= tab.wbase[o + 1])
                o++;
            unsigned int rem = widx - tab.wbase[o];
            int nrows = tab.nrows[o], wpr = tab.wpr[o];
            int layer = rem / (nrows * wpr) + 1;
            rem -= (layer - 1) * nrows * wpr;
            int y = rem / wpr + kBorder;
            int x = (rem % wpr) * 64 + bit + kBorder;
            const int rows = tab.rows[o], cols = tab.cols[o];
            Dog3<T, FROM_DOG> g{tab, o, (size_t)b * tab.img_stride[o], tab.pitch[o]};

            int interval = layer, l = layer;
            double off[3] = {0, 0, 0};
            int i = 0;
            bool alive = true;
            DogCube<T, FROM_DOG> cube;
            const size_t img_off = (size_t)b * tab.img_stride[o];
            const int pitch = tab.pitch[o];
            cube.load(tab, o, img_off, pitch, l, x, y);
            while (i < 5) { // MAX_INTERPOLATION_STEPS
                offset3d_c(cube, off);
                if (fabs(off[0]) < 0.5 && fabs(off[1]) < 0.5 && fabs(off[2]) < 0.5)
                    break;
                // cvRound of an offset far outside int range always fails the range test below
                if (!(fabs(off[0]) < 1.0e9) || !(fabs(off[1]) < 1.0e9) || !(fabs(off[2]) < 1.0e9)) {
                    alive = false;
                    break;
                }
                x += (int)rint(off[0]);
                y += (int)rint(off[1]);
                interval += (int)rint(off[2]);
                l = interval; // idx = index - intvl + interval, as a level inside the octave
                if (interval < 1 || interval > 3 || x >= cols - 1 || x < 2 || y >= rows - 1 || y < 2) {
                    alive = false;
                    break;
                }
                i++;
                if (i < 5)
                    cube.load(tab, o, img_off, pitch, l, x, y); // the block around the new position
            }
            if (alive && i < 5) {
                double dx[3];
                derivative3d_c(cube, dx);
                double term = 0.0;
#pragma unroll
                for (int q = 0; q < 3; q++)
                    term += dx[q] * off[q];
                double contrast = fabs(cube.at(0, 0, 0) + 0.5 * term);
                if (!(contrast < 0.03 / 3)) { // DXTHRESHOLD / INTERVALS
                    refined = true;
                    // passEdgeResponse on the ORIGINAL layer at the refined x,y (sift.cpp:960): a slice of the block
                    // unless the level moved by two
                    double e[3][3]; // [y' - y + 1][x' - x + 1] of level `layer`
                    const int dl = layer - l;
                    if (dl >= -1 && dl <= 1) {
#pragma unroll
                        for (int yi = 0; yi < 3; yi++)
#pragma unroll
                            for (int xi = 0; xi < 3; xi++)
                                e[yi][xi] = dl == 0 ? cube.v[1][yi][xi] : (dl < 0 ? cube.v[0][yi][xi] : cube.v[2][yi][xi]);
                    } else {
#pragma unroll
                        for (int yi = 0; yi < 3; yi++)
#pragma unroll
                            for (int xi = 0; xi < 3; xi++)
                                e[yi][xi] = g.at(layer, x - 1 + xi, y - 1 + yi);
                    }
                    double val = e[1][1];
                    double Dxx = e[1][2] + e[1][0] - 2 * val;
                    double Dyy = e[2][1] + e[0][1] - 2 * val;
                    double Dxy = (e[2][2] + e[0][0] - e[2][0] - e[0][2]) / 4.0;
                    double Tr = Dxx + Dyy;
                    double Det = Dxx * Dyy - Dxy * Dxy;
                    const double r = 10.0; // RATIO
                    if (Det > 0 && Tr * Tr / Det < (r + 1) * (r + 1) / r) {
                        ok = true;
                        rec.x = x; rec.y = y; rec.octave = o; rec.interval = interval;
                        rec.offx = off[0]; rec.offy = off[1]; rec.offi = off[2];
                        double p2 = ldexp(1.0, o);
                        rec.dx = ((x + off[0]) * p2) / 2;
                        rec.dy = ((y + off[1]) * p2) / 2;
                        double intvl = interval + off[2];
                        // scale is the sort key of the output: the last bit of pow() orders features whose scales lie
                        // an ulp apart, so 2^y is evaluated correctly rounded, which is what the C library returns but
                        // for one argument in a thousand
                        rec.scale = (sigma * sg_exp2_cr(o + intvl / 3)) / 2;
                        rec.oscale = sigma * sg_exp2_cr(intvl / 3);
                        rec.val = val;
                    }
                }
            }
        }
        unsigned long long m2 = __ballot(refined), m3 = __ballot(ok);
        n2 += __popcll(m2);
        n3 += __popcll(m3);
        if (ok)
            kp[(size_t)b * kp_img + c] = rec;
        if (lane == 0) {
            unsigned int widx = c0 >> 6;
            valid_words[(size_t)b * vw_img + widx] = m3;
            if (m3)
                atomicAdd(&chunk_sums[(size_t)b * chunks_img + widx / kChunkWords], (unsigned int)__popcll(m3));
        }
    }
    if (lane == 0 && (n2 | n3)) {
        unsigned long long *ctr = counters + (size_t)b * 8;
        atomicAdd(&ctr[3], (unsigned long long)n2);
        atomicAdd(&ctr[4], (unsigned long long)n3);
    }
}

void launch_refine(const OctaveTab &tab, const unsigned int *cand, size_t cand_img_stride,
                   const unsigned int *cand_count, KpRec *kp, size_t kp_img_stride,
                   unsigned long long *valid_words, size_t vw_img_stride, unsigned int *chunk_sums,
                   size_t chunks_img_stride, unsigned long long *counters, double sigma, unsigned int cap, int B,
                   int prec, hipStream_t s)
{
    unsigned int blocks = (cap + 255) / 256;
    if (blocks > 512)
        blocks = 512;
    dim3 g(blocks, B);
#define SG_REF(T, FD)                                                                                             \
    refine_k<T, FD><<<g, 256, 0, s>>>(tab, cand, cand_img_stride, cand_count, kp, kp_img_stride, valid_words,     \
                                      vw_img_stride, chunk_sums, chunks_img_stride, counters, sigma, cap)
    bool fd = tab.from_dog != 0;
    if (prec == SG_F64) { if (fd) SG_REF(double, true); else SG_REF(double, false); }
    else { if (fd) SG_REF(float, true); else SG_REF(float, false); }
#undef SG_REF
}

} // namespace sg
