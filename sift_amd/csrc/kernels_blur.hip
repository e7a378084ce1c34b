// kernels_blur.hip -- separable Gaussian blur (row pass, column pass) for gfx950.
//
// Semantics: cv::GaussianBlur(src,dst,Size(0,0),sigma,sigma) on a one-channel floating plane, the only
// arithmetic the reference takes from OpenCV (reference sift/sift.cpp:323-326): BORDER_REFLECT_101,
// generic row filter  acc = k[0]*s[0]; acc += k[t]*s[t]  (left to right), then the symmetric column
// filter  acc = k[h]*c; acc += k[h+t]*(row[y+t] + row[y-t]).  The double instantiations keep exactly
// that operation order with separate multiplies and adds (the library is built with
// -ffp-contract=off), so a double pyramid is bit-identical to a scalar CPU evaluation of the same
// formulas.  Taps zero-padded to a wider kernel add exact zeros and leave the result unchanged.
//
// Both passes are HBM-bound streaming kernels: 16-byte accesses per lane, halos staged in LDS.
#include "sg_internal.h"
#include "sg_lds_dma.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <type_traits>

namespace sg {

// BORDER_REFLECT_101, repeated until in range (planes narrower than the kernel occur in the
// smallest octaves).
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len)
        p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// BORDER_REFLECT_101 for indices at most len-1 outside [0, len) (one reflection), clamped for anything
// further out (only reached by lanes whose results are masked).  Branch-free.
__device__ __forceinline__ int reflect101_once(int p, int len)
{
    p = p < 0 ? -p : p;
    p = p >= len ? 2 * (len - 1) - p : p;
    return min(max(p, 0), len - 1);
}

// acc + k*x.  The double pyramid keeps the reference's separate multiply and add (the library is built without
// contraction, and results are bit-identical to a scalar evaluation) unless F selects the fused form, which halves
// the arithmetic of the wide filters and rounds once per tap instead of twice (SG_F64_FMA: a tolerance mode, the
// pyramid then differs from the scalar evaluation in its last bits).  The float pyramid is a tolerance mode anyway
// and always takes the fused form.
template <bool F>
__device__ __forceinline__ double blur_madd(double acc, double k, double x)
{
    if constexpr (F)
        return __builtin_fma(k, x, acc);
    else
        return acc + k * x;
}
template <bool F>
__device__ __forceinline__ float blur_madd(float acc, float k, float x) { return __builtin_fmaf(k, x, acc); }

// Tile (strip group, row segment, frame) of a workgroup of the strip-walking kernels.  Plain form: a 3-D launch.
// XCD-aware form (a 1-D launch of 8 * ceil(tiles / 8) workgroups): workgroups are dealt round-robin over the eight
// XCDs, so workgroup i runs on XCD i mod 8; tile numbers are handed out such that each XCD walks a contiguous run of
// the tile list (strip groups fastest, then segments, then frames): neighbouring strips, whose halo columns overlap,
// and the segments of one frame then meet in one L2, and each XCD's address translation covers an eighth of the planes.
// (Placement is an assumption about speed only; every tile is taken exactly once wherever its workgroup runs.)
struct TileGrid {
    int nx, ny, nz;
    int remap; // 1: XCD-aware 1-D launch
};

__device__ __forceinline__ bool tile_of_block(const TileGrid &g, int &bx, int &by, int &bz)
{
    if (!g.remap) {
        bx = blockIdx.x;
        by = blockIdx.y;
        bz = blockIdx.z;
        return true;
    }
    const unsigned int total = (unsigned int)g.nx * g.ny * g.nz, per = (total + 7) / 8;
    const unsigned int t = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || t >= total)
        return false;
    bx = (int)(t % (unsigned int)g.nx);
    const unsigned int r = t / (unsigned int)g.nx;
    by = (int)(r % (unsigned int)g.ny);
    bz = (int)(r / (unsigned int)g.ny);
    return true;
}

static dim3 tile_launch_grid(const TileGrid &g)
{
    if (!g.remap)
        return dim3(g.nx, g.ny, g.nz);
    const unsigned int total = (unsigned int)g.nx * g.ny * g.nz;
    return dim3((total + 7) / 8 * 8);
}

template <typename T, int V>
struct VecOf;
template <>
struct VecOf<double, 2> { typedef double2 type; };
template <>
struct VecOf<float, 4> { typedef float4 type; };

// ---- row pass -------------------------------------------------------------------------------------
// Block = 4 waves, one row segment of TW = 64*V*U pixels per wave: each lane produces U groups of V
// adjacent outputs from a (TW + 2R)-pixel LDS image of the row segment.
template <typename T, int R, int V, int U, bool F = false>
__global__ __launch_bounds__(256) void blur_row_k(const T *__restrict__ src, T *__restrict__ dst, int rows, int cols,
                                                  int spitch, int dpitch, size_t simg, size_t dimg, Taps<T> taps)
{
    typedef typename VecOf<T, V>::type vec_t;
    constexpr int TW = 64 * V * U;
    __shared__ __attribute__((aligned(16))) T lds[4][TW + 2 * R + (V - (2 * R) % V) % V];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int y = blockIdx.y * 4 + wave;
    const int x0 = blockIdx.x * TW;
    const bool row_ok = y < rows;
    const T *s = src + blockIdx.z * simg + (size_t)(row_ok ? y : 0) * spitch;
    T *l = lds[wave];

    if (row_ok) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            int xb = x0 + (u * 64 + lane) * V;
            if (xb + V <= cols) {
                vec_t v = *reinterpret_cast<const vec_t *>(s + xb);
                const T *pv = reinterpret_cast<const T *>(&v);
#pragma unroll
                for (int q = 0; q < V; q++)
                    l[R + (u * 64 + lane) * V + q] = pv[q];
            } else {
#pragma unroll
                for (int q = 0; q < V; q++)
                    l[R + (u * 64 + lane) * V + q] = s[reflect101(xb + q, cols)];
            }
        }
        if (lane < R) {
            l[lane] = s[reflect101(x0 - R + lane, cols)];
            l[R + TW + lane] = s[reflect101(x0 + TW + lane, cols)];
        }
    }
    __syncthreads();
    if (!row_ok)
        return;
    T *d = dst + blockIdx.z * dimg + (size_t)y * dpitch;
#pragma unroll
    for (int u = 0; u < U; u++) {
        int xb = x0 + (u * 64 + lane) * V;
        if (xb >= cols)
            continue;
        T w[V + 2 * R];
#pragma unroll
        for (int q = 0; q < V + 2 * R; q++)
            w[q] = l[(u * 64 + lane) * V + q];
        T acc[V];
#pragma unroll
        for (int v = 0; v < V; v++) {
            acc[v] = taps.k[0] * w[v];
#pragma unroll
            for (int t = 1; t <= 2 * R; t++)
                acc[v] = blur_madd<F>(acc[v], taps.k[t], w[v + t]);
        }
        if (xb + V <= cols) {
            vec_t o;
            T *po = reinterpret_cast<T *>(&o);
#pragma unroll
            for (int v = 0; v < V; v++)
                po[v] = acc[v];
            *reinterpret_cast<vec_t *>(d + xb) = o;
        } else {
#pragma unroll
            for (int v = 0; v < V; v++)
                if (xb + v < cols)
                    d[xb + v] = acc[v];
        }
    }
}

// ---- column pass -----------------------------------------------------------------------------------
// Block = 4 waves over a tile of TW = 64*V columns x TH = 4*RPT rows.  The (TH + 2R) x TW input tile
// is staged in LDS (rows outside the plane by reflection); every thread then produces RPT vertically
// adjacent outputs for its V columns from a register window of RPT + 2R rows.
template <typename T, int R, int V, int RPT, bool F = false>
__global__ __launch_bounds__(256) void blur_col_k(const T *__restrict__ src, T *__restrict__ dst, int rows, int cols,
                                                  int spitch, int dpitch, size_t simg, size_t dimg, Taps<T> taps)
{
    typedef typename VecOf<T, V>::type vec_t;
    constexpr int TW = 64 * V;
    constexpr int TH = 4 * RPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *lds = reinterpret_cast<T *>(smem_raw); // [(TH + 2R)][TW]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int xb = x0 + lane * V;
    const T *s = src + blockIdx.z * simg;

    for (int r = wave; r < TH + 2 * R; r += 4) {
        int yy = reflect101(y0 - R + r, rows);
        const T *sr = s + (size_t)yy * spitch;
        vec_t v;
        T *pv = reinterpret_cast<T *>(&v);
        if (xb + V <= cols) {
            v = *reinterpret_cast<const vec_t *>(sr + xb);
        } else {
#pragma unroll
            for (int q = 0; q < V; q++)
                pv[q] = (xb + q < cols) ? sr[xb + q] : (T)0;
        }
        *reinterpret_cast<vec_t *>(lds + (size_t)r * TW + lane * V) = v;
    }
    __syncthreads();
    if (xb >= cols)
        return;
    const int r0 = wave * RPT;
    if (y0 + r0 >= rows)
        return;
    T win[RPT + 2 * R][V];
#pragma unroll
    for (int q = 0; q < RPT + 2 * R; q++) {
        vec_t v = *reinterpret_cast<const vec_t *>(lds + (size_t)(r0 + q) * TW + lane * V);
        const T *pv = reinterpret_cast<const T *>(&v);
#pragma unroll
        for (int c = 0; c < V; c++)
            win[q][c] = pv[c];
    }
    T *d = dst + blockIdx.z * dimg;
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        int y = y0 + r0 + r;
        if (y >= rows)
            break;
        vec_t o;
        T *po = reinterpret_cast<T *>(&o);
#pragma unroll
        for (int c = 0; c < V; c++) {
            T acc = taps.k[R] * win[r + R][c];
#pragma unroll
            for (int t = 1; t <= R; t++)
                acc = blur_madd<F>(acc, taps.k[R + t], win[r + R + t][c] + win[r + R - t][c]);
            po[c] = acc;
        }
        T *dr = d + (size_t)y * dpitch;
        if (xb + V <= cols) {
            *reinterpret_cast<vec_t *>(dr + xb) = o;
        } else {
#pragma unroll
            for (int c = 0; c < V; c++)
                if (xb + c < cols)
                    dr[xb + c] = po[c];
        }
    }
}

// ---- fused row + column pass ---------------------------------------------------------------------------
// One launch per blur: a block owns a strip of TW = 64*V columns and walks down a segment of rows.  Every
// input row is read from HBM once (16 bytes per lane plus the 2R halo columns), row-filtered by one wave
// through a wave-private LDS image of the raw row, and the row-filtered result enters a ring of LDS rows;
// the column filter reads the ring.  Per round the four waves take K input rows each (row pass), meet at one
// barrier, and then produce K vertically adjacent output rows each (column pass) while the loads of the
// next round are in flight.  The column pass walks the taps outwards from the centre and keeps the K rows'
// operands in two sliding register windows, so a round costs K + 2R ring reads per wave instead of K(2R+1).
// Arithmetic and operation order per pixel are those of the two-pass kernels above, so results are
// bit-identical to them; HBM traffic is one read and one write per pixel.
template <typename T, int R, int V, int K>
__global__ __launch_bounds__(256) void blur_fused_k(const T *__restrict__ src, T *__restrict__ dst, int rows, int cols,
                                                    int spitch, int dpitch, size_t simg, size_t dimg, int seg,
                                                    Taps<T> taps)
{
    typedef typename VecOf<T, V>::type vec_t;
    constexpr int TW = 64 * V;
    constexpr int ROUND = 4 * K;                       // rows per block per round
    constexpr int RING = 2 * R + 2 * ROUND;            // rows readable this round + rows written next round
    constexpr int NWIN = (V + 2 * R + V - 1) / V;      // vectors per row window
    constexpr int RAWW = (R + TW + NWIN * V + 3) & ~3; // raw row image incl. read overshoot, 16-byte rows
    constexpr int KR = 2;                              // raw rows staged at a time
    static_assert(K % KR == 0, "K must be a multiple of the raw staging depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *ring = reinterpret_cast<T *>(smem_raw); // [RING][TW]
    T *rawbase = ring + (size_t)RING * TW;     // [4][KR][RAWW]
    // the wave index as a scalar, so that ring-slot arithmetic stays on the scalar unit
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    T *raw = rawbase + (size_t)wave * KR * RAWW;
    const int x0 = blockIdx.x * TW;
    const int ys = blockIdx.y * seg;
    const int ye = min(ys + seg, rows);
    const int xb = x0 + lane * V;
    const T *s = src + blockIdx.z * simg;
    T *d = dst + blockIdx.z * dimg;
    const int n_rounds = (ye - ys + 2 * R + ROUND - 1) / ROUND;
    const bool full = xb + V <= cols;
    const int hx = lane < R ? x0 - R + lane : x0 + TW + (lane - R); // halo column of this lane (lanes < 2R)
    const int hxr = reflect101(hx, cols);

    vec_t mainv[K];
    T halo[K];
    auto fetch = [&](int t) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int m = ROUND * t + K * wave + k;
            const int yy = reflect101(ys - R + m, rows);
            const T *sr = s + (size_t)yy * spitch;
            if (full) {
                mainv[k] = *reinterpret_cast<const vec_t *>(sr + xb);
            } else {
                T *pv = reinterpret_cast<T *>(&mainv[k]);
#pragma unroll
                for (int q = 0; q < V; q++)
                    pv[q] = sr[reflect101(xb + q, cols)];
            }
            halo[k] = lane < 2 * R ? sr[hxr] : (T)0;
        }
    };
    auto ring_at = [&](int slot) { // slot in [0, 2*RING)
        slot = slot >= RING ? slot - RING : slot;
        return ring + (size_t)slot * TW + lane * V;
    };

    fetch(0);
    int slot_a = (K * wave) % RING;                                // slot of this wave's first input row of the round
    int slot_b = ((RING - (2 * R) % RING) % RING + K * wave) % RING; // slot of virtual row n = -2R + K*wave
    for (int t = 0; t < n_rounds; t++) {
        // ---- row pass: this wave's K input rows, KR at a time through the raw image ----
#pragma unroll
        for (int kb = 0; kb < K; kb += KR) {
#pragma unroll
            for (int k = 0; k < KR; k++) {
                // element stores: the image starts R columns left of the strip, so this address is 16-byte
                // aligned only when R is a multiple of V; the window reads below are the aligned side
                const T *pm = reinterpret_cast<const T *>(&mainv[kb + k]);
#pragma unroll
                for (int q = 0; q < V; q++)
                    raw[k * RAWW + R + lane * V + q] = pm[q];
                if (lane < 2 * R)
                    raw[k * RAWW + (lane < R ? lane : TW + lane)] = halo[kb + k];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // same-wave LDS hand-off between lanes
#pragma unroll
            for (int k = 0; k < KR; k++) {
                vec_t o;
                T *po = reinterpret_cast<T *>(&o);
                T prev[V]; // last element(s) of the previous vector, for the outputs that straddle vectors
#pragma unroll
                for (int q = 0; q < NWIN; q++) {
                    vec_t v = *reinterpret_cast<const vec_t *>(raw + k * RAWW + lane * V + q * V);
                    const T *pv = reinterpret_cast<const T *>(&v);
                    // window element index e = q*V + j feeds output v_out with tap tt = e - v_out
#pragma unroll
                    for (int j = 0; j < V; j++) {
                        const int e = q * V + j;
#pragma unroll
                        for (int vo = 0; vo < V; vo++) {
                            const int tt = e - vo;
                            if (tt == 0)
                                po[vo] = taps.k[0] * pv[j];
                            else if (tt > 0 && tt <= 2 * R)
                                po[vo] = blur_madd<false>(po[vo], taps.k[tt], pv[j]);
                        }
                    }
                    (void)prev;
                }
                int sl = slot_a + kb + k;
                *reinterpret_cast<vec_t *>(ring_at(sl)) = o;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the raw image is overwritten next
        }
        if (t + 1 < n_rounds)
            fetch(t + 1);
        __syncthreads();
        // ---- column pass: K adjacent output rows per wave from the ring ----
        const int n0 = ROUND * t - 2 * R + K * wave; // first output row (segment-relative); ring rows n0 .. n0+K-1+2R
        if (n0 + K > 0 && ys + n0 < ye && xb < cols) {
            vec_t acc[K], up[K], dn[K];
            // centres ring[R + j]
#pragma unroll
            for (int j = 0; j < K; j++) {
                vec_t c = *reinterpret_cast<const vec_t *>(ring_at(slot_b + R + j));
                const T *pc = reinterpret_cast<const T *>(&c);
                T *pa = reinterpret_cast<T *>(&acc[j]);
#pragma unroll
                for (int e = 0; e < V; e++)
                    pa[e] = taps.k[R] * pc[e];
                if (j + 1 < K)
                    dn[j + 1] = c; // ring[R + (j+1) - 1]
                if (j > 0)
                    up[j - 1] = c; // ring[R + (j-1) + 1]
            }
            up[K - 1] = *reinterpret_cast<const vec_t *>(ring_at(slot_b + R + K));
            dn[0] = *reinterpret_cast<const vec_t *>(ring_at(slot_b + R - 1));
#pragma unroll
            for (int tt = 1; tt <= R; tt++) {
#pragma unroll
                for (int j = 0; j < K; j++) {
                    const T *pu = reinterpret_cast<const T *>(&up[j]), *pd = reinterpret_cast<const T *>(&dn[j]);
                    T *pa = reinterpret_cast<T *>(&acc[j]);
#pragma unroll
                    for (int e = 0; e < V; e++)
                        pa[e] = blur_madd<false>(pa[e], taps.k[R + tt], pu[e] + pd[e]);
                }
                if (tt < R) {
                    // slide: up[j] <- ring[R + j + tt + 1], dn[j] <- ring[R + j - tt - 1]
#pragma unroll
                    for (int j = 0; j + 1 < K; j++)
                        up[j] = up[j + 1];
                    up[K - 1] = *reinterpret_cast<const vec_t *>(ring_at(slot_b + R + K + tt));
#pragma unroll
                    for (int j = K - 1; j > 0; j--)
                        dn[j] = dn[j - 1];
                    dn[0] = *reinterpret_cast<const vec_t *>(ring_at(slot_b + R - 1 - tt));
                }
            }
#pragma unroll
            for (int j = 0; j < K; j++) {
                const int n = n0 + j, y = ys + n;
                if (n >= 0 && y < ye) {
                    T *dr = d + (size_t)y * dpitch;
                    if (full) {
                        *reinterpret_cast<vec_t *>(dr + xb) = acc[j];
                    } else {
                        const T *pa = reinterpret_cast<const T *>(&acc[j]);
#pragma unroll
                        for (int e = 0; e < V; e++)
                            if (xb + e < cols)
                                dr[xb + e] = pa[e];
                    }
                }
            }
        }
        slot_a += ROUND;
        slot_a = slot_a >= RING ? slot_a - RING : slot_a;
        slot_b += ROUND;
        slot_b = slot_b >= RING ? slot_b - RING : slot_b;
    }
}

template <typename T, int R, int V, int K>
static size_t fused_lds_bytes()
{
    constexpr int TW = 64 * V, ROUND = 4 * K, RING = 2 * R + 2 * ROUND, KR = 2;
    constexpr int NWIN = (V + 2 * R + V - 1) / V, RAWW = (R + TW + NWIN * V + 3) & ~3;
    return sizeof(T) * ((size_t)RING * TW + (size_t)4 * KR * RAWW);
}

// ---- fused row + column pass, column window in registers ------------------------------------------------
// Every wave is on its own: it owns a strip of TW = 64*V columns and walks down a segment of rows.  An input
// row is read from HBM once (16 bytes per lane plus the 2R halo columns), row-filtered through a wave-private
// LDS image of the raw row, and the row-filtered vector enters a window of the last 2R+KR rows that the lane
// holds in registers; the column filter is evaluated on the window.  KR input rows are taken per iteration,
// so that the window is shifted by KR register moves per KR rows; the loads of the next iteration are in
// flight during the arithmetic.  No workgroup barrier, no LDS traffic for the column pass.  Arithmetic and
// operation order per pixel are those of the two-pass kernels above, so results are bit-identical to them;
// HBM traffic is one read and one write per pixel (plus 2R re-read rows per segment).
#ifndef SG_WIN_PF
#define SG_WIN_PF 2 // iterations (row pairs) of input prefetch in the register-window kernel
#endif
#ifndef SG_WIN_WPE13
#define SG_WIN_WPE13 2
#endif
#ifndef SG_WIN_WPE10
#define SG_WIN_WPE10 3
#endif
#ifndef SG_WIN_WPE8
#define SG_WIN_WPE8 3
#endif
#ifndef SG_WIN_WPE6
#define SG_WIN_WPE6 4
#endif
// resident waves per SIMD the register allocation of each radius aims at
constexpr int win_waves_per_simd(int R) { return R <= 6 ? SG_WIN_WPE6 : (R <= 8 ? SG_WIN_WPE8 : (R <= 10 ? SG_WIN_WPE10 : SG_WIN_WPE13)); }

// Input rows per iteration: more rows per iteration mean fewer window moves per row (2R per iteration) and fewer LDS
// hand-offs; worth its registers only at the widest radius, whose kernel runs at two waves per SIMD anyway.
#ifndef SG_WIN_KR13
#define SG_WIN_KR13 4
#endif
constexpr int win_rows_per_iter(int R) { return R > 10 ? SG_WIN_KR13 : 2; }

#ifndef SG_UPS_DI
#define SG_UPS_DI 3
#endif
// Gray rows of the upsampling form staged by direct-to-LDS loads (GDMA): geometry of a wave's ring.
template <typename T, int R, int V>
struct WinGray {
    static constexpr int TW = 64 * V, KR = win_rows_per_iter(R);
    static constexpr int FRONT = (R + 3) / 2 + 1 + ((R + 3) / 2 + 1) % 2; // gray columns in front of the strip's first (even)
    static constexpr int GCOLS = FRONT + TW / 2 + (R + 1) / 2 + 2;         // gray columns a row image must hold
    static constexpr int NP = (GCOLS + 1) / 2;                             // 16-byte pieces (two doubles) per gray row
    static constexpr int NI = (NP + 63) / 64;                              // load instructions per gray row
    static constexpr int ROWB = NP * 16, ROWE = NP * 2;
    static constexpr int DI = SG_UPS_DI;                                   // iterations in the ring
    // The KR consecutive output rows of an iteration (after reflection at the border and the repeat of the last rows:
    // still rows next to each other, or the same row) are formed from KR / 2 + 1 consecutive gray rows: an odd row
    // 2i + 1 needs gray rows i and i + 1, the even rows beside it need i or i + 1.
    static constexpr int GROWS = KR / 2 + 1;                               // gray rows staged per iteration
    static constexpr int SLOTB = GROWS * ROWB;
    static constexpr int WAVE_BYTES = DI * SLOTB;
    static constexpr int PER = GROWS * NI;                                 // load instructions per iteration
    static_assert((DI - 1) * PER < 64, "outstanding loads fit the counter");
};

template <typename T, int R, int V, bool UPS = false, bool GDMA = false, bool F = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(win_waves_per_simd(R), win_waves_per_simd(R))))
void blur_win_k(const T *__restrict__ src, T *__restrict__ dst, int rows, int cols, int spitch, int dpitch, size_t simg,
                size_t dimg, int seg, Taps<T> taps, T *__restrict__ dec, int dec_rows, int dec_cols, int dec_pitch,
                size_t dec_img, TileGrid tg)
{
    int bx, by, bz;
    if (!tile_of_block(tg, bx, by, bz))
        return; // the whole workgroup; there is no barrier in this kernel
    typedef typename VecOf<T, V>::type vec_t;
    constexpr int TW = 64 * V;
    constexpr int KR = win_rows_per_iter(R);           // input rows per iteration
    constexpr int WIN = 2 * R + KR;                    // rows in the register window
    constexpr int NWIN = (V + 2 * R + V - 1) / V;      // vectors per row window
    constexpr int RAWW = (R + TW + NWIN * V + 3) & ~3; // raw row image incl. read overshoot, 16-byte rows
    static_assert(KR % 2 == 0, "row parity inside an iteration is the image's");
    static_assert(!GDMA || UPS, "the gray ring belongs to the upsampling form");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    T *raw = reinterpret_cast<T *>(smem_raw) + (size_t)wave * KR * RAWW;
    // the four waves of a workgroup take four neighbouring strips of the same rows, so that the halo columns one
    // wave reads are the columns its neighbour streams at about the same time
    const int x0 = (bx * 4 + wave) * TW;
    const int ys = by * seg;
    if (x0 >= cols)
        return; // the whole wave; there is no barrier in this kernel
    const int ye = min(ys + seg, rows);
    const int xb = x0 + lane * V;
    const T *s = src + bz * simg;
    T *d = dst + bz * dimg;
    const int n_iter = (ye - ys + 2 * R + KR - 1) / KR;
    const bool strip_full = x0 + TW <= cols; // wave-uniform: every lane's vector lies inside the row
    const int hx = lane < R ? x0 - R + lane : x0 + TW + (lane - R); // halo column of this lane (lanes < 2R)
    // lanes >= 2R have no halo column: they re-read their own first pixel (a line the main load brings anyway)
    const int hxr = lane < 2 * R ? reflect101_once(hx, cols) : min(xb, cols - 1);
    int xe[V]; // element columns for the strip that crosses the right border
#pragma unroll
    for (int q = 0; q < V; q++)
        xe[q] = reflect101_once(xb + q, cols);

    // Input rows are fetched PF iterations ahead of their use: the arithmetic of one iteration is shorter than the
    // memory latency under load, and a wave has few neighbours on its SIMD to hide it.
    constexpr int PF = (UPS || KR > 2) ? 1 : SG_WIN_PF;
    vec_t mainv_s[PF][KR];
    T halo_s[PF][KR];
    // UPS: the input plane is the 2x upsampling (reference UpSample, sift/sift.cpp:111-175) of a double gray plane
    // of rows/2 x cols/2 pixels that `src` points at (spitch, simg are that plane's); its pixels are formed here
    // instead of being read.  Output (Y, X) with Y' = min(Y, rows-3), X' = min(X, cols-3) (the last two rows and
    // columns repeat row rows-3 and column cols-3), i = Y'/2, j = X'/2:  s(i,j) | (s(i,j)+s(i,j+1))/2 |
    // (s(i,j)+s(i+1,j))/2 | (s(i+1,j)+s(i,j)+s(i+1,j+1)+s(i,j+1))/4  by the parities of Y' and X'.
    const double *gsrc = reinterpret_cast<const double *>(src) + bz * simg;
    // GDMA: the gray rows an iteration needs (two per output row: i = Y'/2 and i + 1) are staged into a wave-private
    // LDS ring by direct-to-LDS loads, DI iterations ahead, and the upsampled pixels are formed from there; a gray row
    // image starts FRONT columns left of the strip's first gray column (an even column: 16-byte pieces are aligned).
    using GR = WinGray<T, R, V>;
    constexpr int kRawBytes = (4 * KR * RAWW * (int)sizeof(T) + 15) & ~15;
    const int jb = (x0 >> 1) - GR::FRONT; // gray column of element 0 of a row image
    const double *gring = reinterpret_cast<const double *>(smem_raw + kRawBytes + (size_t)wave * GR::WAVE_BYTES);
    const unsigned int lds_gring = (unsigned int)(uintptr_t)smem_raw + (unsigned int)kRawBytes + (unsigned int)wave * GR::WAVE_BYTES;
    unsigned int gvoff[GR::NI];
#pragma unroll
    for (int q = 0; q < GR::NI; q++)
        gvoff[q] = (unsigned int)min(max(jb + 2 * (lane + 64 * q), 0), spitch - 2) * 8u;
    auto gray_first = [&](int t) { // first of the gray rows iteration t's output rows are formed from (wave-uniform)
        int g = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < KR; k++)
            g = min(g, min(reflect101_once(ys - R + KR * t + k, rows), rows - 3) >> 1);
        return g;
    };
    auto stage_gray = [&](int t) { // iteration t's gray rows into slot t mod DI
        const int g_first = gray_first(t);
#pragma unroll
        for (int h = 0; h < GR::GROWS; h++) {
            const int gr = min(g_first + h, (rows >> 1) - 1);
            const unsigned long long base = (unsigned long long)(uintptr_t)(gsrc + (size_t)gr * spitch);
            const unsigned int to = lds_gring + (unsigned int)((t % GR::DI) * GR::SLOTB + h * GR::ROWB);
#pragma unroll
            for (int q = 0; q < GR::NI; q++)
                if (q + 1 < GR::NI || lane < GR::NP - 64 * (GR::NI - 1))
                    load16_to_lds(to + 1024u * q, base, gvoff[q]);
        }
    };
    int uj[V], uo[V]; // j and column parity of the lane's elements, and of its halo column
#pragma unroll
    for (int q = 0; q < V; q++) {
        const int xc = min(xe[q], cols - 3);
        uj[q] = xc >> 1;
        uo[q] = xc & 1;
    }
    const int hxc = min(hxr, cols - 3);
    const int hj = hxc >> 1, ho = hxc & 1;
    const bool strip_inner = x0 + TW <= cols - 2; // wave-uniform: no element of the strip is a repeated column
    auto up_value = [&](const double *g0, const double *g1, bool oddr, int j, int oc) -> T {
        const double s00 = g0[j], s01 = g0[j + 1];
        if (oddr) {
            const double s10 = g1[j], s11 = g1[j + 1];
            return (T)(oc ? (s10 + s00 + s11 + s01) / 4 : (s00 + s10) / 2);
        }
        return (T)(oc ? (s00 + s01) / 2 : s00);
    };
    auto fetch = [&](int t, vec_t *mainv, T *halo) {
#pragma unroll
        for (int k = 0; k < KR; k++) {
            const int yy = reflect101_once(ys - R + KR * t + k, rows);
            if (UPS) {
                const int yc = min(yy, rows - 3);
                const bool oddr = (yc & 1) != 0; // wave-uniform
                const double *g0, *g1;
                if (GDMA) { // element 0 of a staged row is gray column jb; the slot starts at gray row gray_first(t)
                    g0 = gring + ((t % GR::DI) * (GR::SLOTB / 8) + ((yc >> 1) - gray_first(t)) * GR::ROWE) - jb;
                    g1 = g0 + GR::ROWE; // (read for odd rows only, which sit on the slot's first row or share it: see WinGray)
                } else {
                    g0 = gsrc + (size_t)(yc >> 1) * spitch;
                    g1 = g0 + spitch;
                }
                T *pv = reinterpret_cast<T *>(&mainv[k]);
                if (strip_inner) {
                    // the V elements are (even, odd) column pairs over V/2 source columns
#pragma unroll
                    for (int q = 0; q < V; q += 2) {
                        const int j = (xb + q) >> 1;
                        const double s00 = g0[j], s01 = g0[j + 1];
                        if (oddr) {
                            const double s10 = g1[j], s11 = g1[j + 1];
                            pv[q] = (T)((s00 + s10) / 2);
                            pv[q + 1] = (T)((s10 + s00 + s11 + s01) / 4);
                        } else {
                            pv[q] = (T)s00;
                            pv[q + 1] = (T)((s00 + s01) / 2);
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < V; q++)
                        pv[q] = up_value(g0, g1, oddr, uj[q], uo[q]);
                }
                halo[k] = up_value(g0, g1, oddr, hj, ho);
                continue;
            }
            const T *sr = s + (size_t)yy * spitch;
            if (strip_full) {
                mainv[k] = *reinterpret_cast<const vec_t *>(sr + xb);
            } else {
                T *pv = reinterpret_cast<T *>(&mainv[k]);
#pragma unroll
                for (int q = 0; q < V; q++)
                    pv[q] = *reinterpret_cast<const volatile T *>(sr + xe[q]); // kept apart from the vector load above
            }
            halo[k] = sr[hxr];
        }
    };

    vec_t win[WIN];
#pragma unroll
    for (int j = 0; j < WIN; j++) {
        T *pw = reinterpret_cast<T *>(&win[j]);
#pragma unroll
        for (int e = 0; e < V; e++)
            pw[e] = (T)0;
    }
    if (GDMA) {
        static_assert(!GDMA || PF == 1, "one fetch per iteration, for the next one");
        for (int t = 0; t < min(GR::DI, n_iter); t++)
            stage_gray(t);
        wait_groups_ahead<GR::PER, GR::DI - 1>(n_iter - 1);
    }
#pragma unroll
    for (int p = 0; p < PF; p++)
        if (p < n_iter)
            fetch(p, mainv_s[p], halo_s[p]);
    for (int t = 0; t < n_iter; t++) {
        vec_t *mainv = mainv_s[0];
        T *halo = halo_s[0];
#pragma unroll
        for (int j = 0; j + KR < WIN; j++)
            win[j] = win[j + KR];
        // ---- row pass: KR input rows through the raw image ----
#pragma unroll
        for (int k = 0; k < KR; k++) {
            // element stores: the image starts R columns left of the strip, so this address is 16-byte
            // aligned only when R is a multiple of V; the window reads below are the aligned side
            const T *pm = reinterpret_cast<const T *>(&mainv[k]);
#pragma unroll
            for (int q = 0; q < V; q++)
                raw[k * RAWW + R + lane * V + q] = pm[q];
            if (lane < 2 * R)
                raw[k * RAWW + (lane < R ? lane : TW + lane)] = halo[k];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // same-wave LDS hand-off between lanes
        // stage 0 is consumed: the later stages move up and the row pair PF iterations ahead is requested
#pragma unroll
        for (int p = 0; p + 1 < PF; p++) {
#pragma unroll
            for (int k = 0; k < KR; k++) {
                mainv_s[p][k] = mainv_s[p + 1][k];
                halo_s[p][k] = halo_s[p + 1][k];
            }
        }
        // slot t mod DI held this iteration's gray rows: the stores above consumed what was read from it
        if (GDMA && t + GR::DI < n_iter)
            stage_gray(t + GR::DI);
        if (t + PF < n_iter) {
            if (GDMA) // iterations t+2 .. t+DI (those that exist) may stay in flight
                wait_groups_ahead<GR::PER, GR::DI - 1>(n_iter - 1 - (t + PF));
            fetch(t + PF, mainv_s[PF - 1], halo_s[PF - 1]);
        }
#pragma unroll
        for (int k = 0; k < KR; k++) {
            vec_t o;
            T *po = reinterpret_cast<T *>(&o);
            // all window reads of the row are issued before the arithmetic starts
            vec_t rv[NWIN];
#pragma unroll
            for (int q = 0; q < NWIN; q++)
                rv[q] = *reinterpret_cast<const vec_t *>(raw + k * RAWW + lane * V + q * V);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < NWIN; q++) {
                const vec_t v = rv[q];
                const T *pv = reinterpret_cast<const T *>(&v);
                // window element index e = q*V + j feeds output vo with tap tt = e - vo
#pragma unroll
                for (int j = 0; j < V; j++) {
                    const int e = q * V + j;
#pragma unroll
                    for (int vo = 0; vo < V; vo++) {
                        const int tt = e - vo;
#ifdef SG_X_WIN_NOROW // timing experiment: the row filter keeps two terms per output
                        if (tt == 0)
                            po[vo] = taps.k[0] * pv[j];
                        else if (tt == 2 * R)
                            po[vo] = po[vo] + pv[j];
#else
                        if (tt == 0)
                            po[vo] = taps.k[0] * pv[j];
                        else if (tt > 0 && tt <= 2 * R)
                            po[vo] = blur_madd<F>(po[vo], taps.k[tt], pv[j]);
#endif
                    }
                }
            }
            win[2 * R + k] = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the raw image is overwritten next
        // ---- column pass: output rows n0 .. n0+KR-1 (segment-relative) from window rows k .. k+2R ----
        const int n0 = KR * t - 2 * R;
        if (n0 + KR > 0 && xb < cols) {
#pragma unroll
            for (int k = 0; k < KR; k++) {
                if (n0 + k < 0)
                    continue; // the window is not complete for this row yet (2R is not a multiple of KR)
                vec_t acc;
                T *pa = reinterpret_cast<T *>(&acc);
                const T *pc = reinterpret_cast<const T *>(&win[R + k]);
#pragma unroll
                for (int e = 0; e < V; e++)
                    pa[e] = taps.k[R] * pc[e];
#ifdef SG_X_WIN_NOCOL // timing experiment: the column filter keeps its outermost pair
#pragma unroll
                for (int tt = R; tt <= R; tt++) {
#else
#pragma unroll
                for (int tt = 1; tt <= R; tt++) {
#endif
                    const T *pu = reinterpret_cast<const T *>(&win[R + k + tt]);
                    const T *pd = reinterpret_cast<const T *>(&win[R + k - tt]);
#pragma unroll
                    for (int e = 0; e < V; e++)
                        pa[e] = blur_madd<F>(pa[e], taps.k[R + tt], pu[e] + pd[e]);
                }
                const int y = ys + n0 + k;
#ifdef SG_X_WIN_NOSTORE // timing experiment: (almost) nothing is written
                if (y < ye && pa[0] == (T)123.456) {
#else
                if (y < ye) {
#endif
                    T *dr = d + (size_t)y * dpitch;
                    if (strip_full) {
                        *reinterpret_cast<vec_t *>(dr + xb) = acc;
                    } else {
#pragma unroll
                        for (int e = 0; e < V; e++)
                            if (xb + e < cols)
                                dr[xb + e] = pa[e];
                    }
                    // DownSample (reference sift/sift.cpp:59-103) of the level just written: dec(m, n) = dst(2m, 2n).
                    // ys and n0 are even, so the even rows are the even k.
                    if ((k & 1) == 0 && dec != nullptr && (y >> 1) < dec_rows) {
                        T *dd = dec + bz * dec_img + (size_t)(y >> 1) * dec_pitch;
#pragma unroll
                        for (int e = 0; e < V; e += 2)
                            if (((xb + e) >> 1) < dec_cols)
                                dd[(xb + e) >> 1] = pa[e];
                    }
                }
            }
        }
    }
}

template <typename T, int R, int V>
static size_t win_lds_bytes()
{
    constexpr int TW = 64 * V, KR = win_rows_per_iter(R);
    constexpr int NWIN = (V + 2 * R + V - 1) / V, RAWW = (R + TW + NWIN * V + 3) & ~3;
    return sizeof(T) * (size_t)4 * KR * RAWW;
}

// ---- the register-window kernel with its input rows staged by direct-to-LDS loads ---------------------------
// Same strips, same window, same arithmetic as blur_win_k; what differs is how an input row reaches the wave's raw
// row image.  There the lanes load the row into registers (one or two iterations ahead) and store it to LDS; here
// the wave owns a ring of DI iteration slots (KR rows each) in LDS and fills it with global_load_lds_dwordx4, DI - 1
// iterations ahead of the one it works on: no registers hold data in flight, no LDS stores are issued, and the
// distance to the loads no longer depends on the register budget.
// A row image starts A columns left of the halo, A = (V - R mod V) mod V, so that its first column (x0 - R - A) is
// a multiple of V and every 16-byte piece is aligned in memory; the lane windows start at the image's element
// lane*V (aligned 16-byte LDS reads) and tap t of output v is window element A + v + t.  Pieces that would start
// outside [0, pitch - V] are moved inside; they hold no column a stored output needs, except the reflected ones:
// those (columns -R..-1 in the first strip, cols..cols+R-1 where the halo crosses the right border) are copied
// inside LDS from their mirror columns after the row has landed.
#ifndef SG_DMA_DI
#define SG_DMA_DI 4
#endif
#ifndef SG_DMA_DI13
#define SG_DMA_DI13 3
#endif
#ifndef SG_DMA_SPLIT_ROWS
#define SG_DMA_SPLIT_ROWS(R) ((R) >= 10)
#endif
constexpr int dma_ring_iters(int R) { return R > 10 ? SG_DMA_DI13 : SG_DMA_DI; }
#ifndef SG_DMA_WPE6
#define SG_DMA_WPE6 4
#endif
#ifndef SG_DMA_WPE8
#define SG_DMA_WPE8 3
#endif
#ifndef SG_DMA_WPE10
#define SG_DMA_WPE10 3
#endif
#ifndef SG_DMA_WPE13
#define SG_DMA_WPE13 2
#endif
// resident waves per SIMD the register allocation of each radius aims at
constexpr int dma_waves_per_simd(int R) { return R <= 6 ? SG_DMA_WPE6 : (R <= 8 ? SG_DMA_WPE8 : (R <= 10 ? SG_DMA_WPE10 : SG_DMA_WPE13)); }

template <typename T, int R, int V>
struct WinDma {
    static constexpr int TW = 64 * V;
    static constexpr int KR = win_rows_per_iter(R);
    static constexpr int A = (V - R % V) % V;
    static constexpr int NWIN = (A + 2 * R + V + V - 1) / V; // 16-byte vectors per lane window
    static constexpr int NP = 64 + NWIN - 1;                 // 16-byte pieces per row image
    static constexpr int ROWB = NP * 16;
    static constexpr int DI = dma_ring_iters(R);
    static constexpr int WAVE_BYTES = DI * KR * ROWB;
    static_assert(NP > 64 && NP <= 128, "two load instructions per row");
    static_assert((DI - 1) * 2 * KR < 64, "outstanding loads fit the counter");
};

template <typename T, int R, int V, bool F = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(dma_waves_per_simd(R), dma_waves_per_simd(R))))
void blur_dma_k(const T *__restrict__ src, T *__restrict__ dst, int rows, int cols, int spitch, int dpitch, size_t simg,
                size_t dimg, int seg, Taps<T> taps, T *__restrict__ dec, int dec_rows, int dec_cols, int dec_pitch,
                size_t dec_img, TileGrid tg)
{
    int bx, by, bz;
    if (!tile_of_block(tg, bx, by, bz))
        return; // the whole workgroup, before it has requested anything; there is no barrier in this kernel
    typedef typename VecOf<T, V>::type vec_t;
    using G = WinDma<T, R, V>;
    constexpr int TW = G::TW, KR = G::KR, WIN = 2 * R + KR, A = G::A, NWIN = G::NWIN, DI = G::DI;
    constexpr int ROWE = G::ROWB / (int)sizeof(T); // elements per row image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int x0 = (bx * 4 + wave) * TW;
    const int ys = by * seg;
    if (x0 >= cols)
        return; // the whole wave, before it has requested anything; there is no barrier in this kernel
    const int ye = min(ys + seg, rows);
    const int xb = x0 + lane * V;
    const T *s = src + bz * simg;
    T *d = dst + bz * dimg;
    const int n_iter = (ye - ys + 2 * R + KR - 1) / KR;
    const bool strip_full = x0 + TW <= cols; // wave-uniform: every lane's vector lies inside the row
    const int cbase = x0 - R - A;            // column of element 0 of a row image
    T *ring = reinterpret_cast<T *>(smem_raw + (size_t)wave * G::WAVE_BYTES);
    const unsigned int lds_wave = (unsigned int)(uintptr_t)smem_raw + (unsigned int)wave * G::WAVE_BYTES;
    const unsigned int voff0 = (unsigned int)min(max(cbase + V * lane, 0), spitch - V) * (unsigned int)sizeof(T);
    const unsigned int voff1 = (unsigned int)min(max(cbase + TW + V * lane, 0), spitch - V) * (unsigned int)sizeof(T);
    const bool left_edge = x0 == 0, right_edge = x0 + TW + R > cols; // wave-uniform
    auto stage = [&](int t, int slot) { // the KR rows of iteration t into a slot
#pragma unroll
        for (int k = 0; k < KR; k++) {
            const int yy = reflect101_once(ys - R + KR * t + k, rows);
            const unsigned long long base = (unsigned long long)(uintptr_t)(s + (size_t)yy * spitch);
            const unsigned int to = lds_wave + (unsigned int)((slot * KR + k) * G::ROWB);
            load16_to_lds(to, base, voff0);
            if (lane < G::NP - 64)
                load16_to_lds(to + 1024u, base, voff1);
        }
    };

    vec_t win[WIN];
#pragma unroll
    for (int j = 0; j < WIN; j++) {
        T *pw = reinterpret_cast<T *>(&win[j]);
#pragma unroll
        for (int e = 0; e < V; e++)
            pw[e] = (T)0;
    }
    for (int t = 0; t < min(DI, n_iter); t++)
        stage(t, t);
    int slot = 0;
    for (int t = 0; t < n_iter; t++) {
        // iterations t+1 .. t+DI-1 (those that exist) may stay in flight
        wait_groups_ahead<2 * KR, DI - 1>(n_iter - 1 - t);
        T *raw = ring + slot * (KR * ROWE);
        if (left_edge) {
#pragma unroll
            for (int k = 0; k < KR; k++)
                if (lane < R) {
                    const T v = raw[k * ROWE + R + A + 1 + lane]; // column 1 + lane
                    raw[k * ROWE + R + A - 1 - lane] = v;         // is column -1 - lane
                }
        }
        if (right_edge) {
            const int to = cols + lane - cbase, from = cols - 2 - lane - cbase;
#pragma unroll
            for (int k = 0; k < KR; k++)
                if (lane < R && to < ROWE) {
                    const T v = raw[k * ROWE + from];
                    raw[k * ROWE + to] = v;
                }
        }
#pragma unroll
        for (int j = 0; j + KR < WIN; j++)
            win[j] = win[j + KR];
        // ---- row pass ----
        // The row windows are read in pieces of CH vectors, one piece ahead of the arithmetic: the reads of a piece
        // are issued, then the piece before it is used.  Two pieces are live at a time, which bounds the registers
        // (the wide radii split a row's window into two pieces; the others take a row per piece).
        {
            constexpr int NPC = SG_DMA_SPLIT_ROWS(R) ? 2 : 1; // pieces per row
            constexpr int CH = (NWIN + NPC - 1) / NPC;
            vec_t rv[2][CH];
            vec_t o[KR];
            auto read_piece = [&](int step, vec_t *buf) {
                const int k = step / NPC, c = step % NPC;
#pragma unroll
                for (int q = c * CH; q < (c + 1) * CH && q < NWIN; q++) {
                    // one 16-byte LDS read per vector (volatile, so that it stays one): where the first element of a
                    // window is not used (A = 1) the compiler otherwise re-pairs the elements into 8-byte-aligned
                    // ds_read2_b64 accesses, which collide on the banks
                    typedef T lds_vec_t __attribute__((ext_vector_type(V)));
                    typedef const volatile __attribute__((address_space(3))) lds_vec_t *lds_vec_ptr; // an LDS pointer, not a flat one
                    const lds_vec_t t = *(lds_vec_ptr)(raw + k * ROWE + lane * V + q * V);
                    T *pb = reinterpret_cast<T *>(&buf[q - c * CH]);
#pragma unroll
                    for (int e = 0; e < V; e++)
                        pb[e] = t[e];
                }
            };
            read_piece(0, rv[0]);
#pragma unroll
            for (int step = 0; step < KR * NPC; step++) {
                const int k = step / NPC, c = step % NPC;
                if (step + 1 < KR * NPC)
                    read_piece(step + 1, rv[(step + 1) & 1]);
                asm volatile("" ::: "memory");
                T *po = reinterpret_cast<T *>(&o[k]);
#pragma unroll
                for (int q = c * CH; q < (c + 1) * CH && q < NWIN; q++) {
                    const vec_t v = rv[step & 1][q - c * CH];
                    const T *pv = reinterpret_cast<const T *>(&v);
                    // window element index e = q*V + j feeds output vo with tap tt = e - vo - A
#pragma unroll
                    for (int j = 0; j < V; j++) {
                        const int e = q * V + j;
#pragma unroll
                        for (int vo = 0; vo < V; vo++) {
                            const int tt = e - vo - A;
                            if (tt == 0)
                                po[vo] = taps.k[0] * pv[j];
                            else if (tt > 0 && tt <= 2 * R)
                                po[vo] = blur_madd<F>(po[vo], taps.k[tt], pv[j]);
                        }
                    }
                }
                // the piece after next is read after this piece's arithmetic
#pragma unroll
                for (int vo = 0; vo < V; vo++)
                    asm volatile("" : "+v"(po[vo])::"memory");
                if (c == NPC - 1)
                    win[2 * R + k] = o[k];
            }
        }
        wait_lgkm0(); // the slot's reads have returned: it is loaded again
        if (t + DI < n_iter)
            stage(t + DI, slot);
        slot = slot + 1 == DI ? 0 : slot + 1;
        // ---- column pass: output rows n0 .. n0+KR-1 (segment-relative) from window rows k .. k+2R ----
        const int n0 = KR * t - 2 * R;
        if (n0 + KR > 0 && xb < cols) {
#pragma unroll
            for (int k = 0; k < KR; k++) {
                if (n0 + k < 0)
                    continue;
                vec_t acc;
                T *pa = reinterpret_cast<T *>(&acc);
                const T *pc = reinterpret_cast<const T *>(&win[R + k]);
#pragma unroll
                for (int e = 0; e < V; e++)
                    pa[e] = taps.k[R] * pc[e];
#pragma unroll
                for (int tt = 1; tt <= R; tt++) {
                    const T *pu = reinterpret_cast<const T *>(&win[R + k + tt]);
                    const T *pd = reinterpret_cast<const T *>(&win[R + k - tt]);
#pragma unroll
                    for (int e = 0; e < V; e++)
                        pa[e] = blur_madd<F>(pa[e], taps.k[R + tt], pu[e] + pd[e]);
                }
                const int y = ys + n0 + k;
                if (y < ye) {
                    T *dr = d + (size_t)y * dpitch;
                    if (strip_full) {
                        *reinterpret_cast<vec_t *>(dr + xb) = acc;
                    } else {
#pragma unroll
                        for (int e = 0; e < V; e++)
                            if (xb + e < cols)
                                dr[xb + e] = pa[e];
                    }
                    // DownSample (reference sift/sift.cpp:59-103) of the level just written: dec(m, n) = dst(2m, 2n)
                    if ((k & 1) == 0 && dec != nullptr && (y >> 1) < dec_rows) {
                        T *dd = dec + bz * dec_img + (size_t)(y >> 1) * dec_pitch;
#pragma unroll
                        for (int e = 0; e < V; e += 2)
                            if (((xb + e) >> 1) < dec_cols)
                                dd[(xb + e) >> 1] = pa[e];
                    }
                }
            }
        }
    }
}

// ---- two consecutive levels in one launch -------------------------------------------------------------------------
// Levels 1 and 2 of an octave are bound by the memory system (11 and 13 taps), and level 1 is read back only to make
// level 2.  Here a wave makes both in one walk: the level-1 rows it has just produced go to global memory and, at the
// same time, into a wave-private LDS row image from which the second row filter takes its lane windows; a second
// register window holds the row-filtered level-1 rows for the second column filter.  Per pixel the pair then moves
// one read and two writes (3 x sizeof(T)) instead of two reads and two writes.
//
// Geometry.  All stages of a wave work on the same 64*V columns [X0, X0 + TW), lane i on columns X0 + V i ...; the
// second row filter needs level 1 on RB columns either side of an output, so only the inner PITCH = TW - 2 MG columns
// of a strip are final (MG >= RB, rounded up so that PITCH columns are whole 128-byte lines): strips advance by PITCH,
// X0 = strip * PITCH - MG, and both levels are stored for the inner columns only (every pixel of both levels is
// written once; the 2 MG outer level-1 columns are computed again by the neighbouring strips).  Level-1 columns outside the plane are not blurs of reflected input -- the row filter adds
// left to right, so that would differ in the last bit from the reference, which reflects level 1 itself -- they are
// copied inside the LDS image from their mirror columns.  Rows need no such care: the column filter adds the rows
// above and below in symmetric pairs, so level 1 of a reflected row index, computed from reflected input rows, is
// bit-identical to level 1 of the mirror row.  A segment starts RA + RB rows early; what the stages produce before
// their windows have filled is never stored.  Arithmetic and operation order per pixel are those of the
// single-level kernels.
// Keeps the compiler from moving LDS accesses across this point.  The hardware needs nothing more between a wave's own
// LDS writes and its later LDS reads of the same bytes by other lanes: a wave's LDS instructions execute in issue order.
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

template <typename T, int RA, int RB, int V>
struct PairGeom {
    using GA = WinDma<T, RA, V>; // the input ring is that of the single-level kernel
    static constexpr int TW = 64 * V, KR = 2, LEAD = RA + RB;
    // columns a strip computes but does not own, either side: at least RB, and such that PITCH columns are whole
    // 128-byte lines (a strip's stores then never share a line with its neighbour's)
    static constexpr int MG = (RB * (int)sizeof(T) + 63) / 64 * 64 / (int)sizeof(T);
    static constexpr int PITCH = TW - 2 * MG;
    static constexpr int AB = (V - RB % V) % V;
    static constexpr int NWINB = (AB + 2 * RB + V + V - 1) / V;
    static constexpr int NPB = 64 + NWINB - 1;
    static constexpr int ROWB_B = NPB * 16;
    static constexpr int WAVE_BYTES = GA::WAVE_BYTES + KR * ROWB_B;
    static_assert(GA::KR == KR, "both stages take two rows per iteration");
    static_assert(PITCH % V == 0 && MG % V == 0 && MG >= RB && PITCH > 0, "strip starts stay multiples of the vector width");
};

#ifndef SG_PAIR_WPE
#define SG_PAIR_WPE 3
#endif
template <typename T, int RA, int RB, int V, bool F = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SG_PAIR_WPE, SG_PAIR_WPE)))
void blur_pair_k(const T *__restrict__ src, T *__restrict__ dst_a, T *__restrict__ dst_b, int rows, int cols, int spitch,
                 int apitch, int bpitch, size_t simg, size_t aimg, size_t bimg, int seg, Taps<T> taps_a, Taps<T> taps_b,
                 TileGrid tg)
{
    int bx, by, bz;
    if (!tile_of_block(tg, bx, by, bz))
        return; // the whole workgroup, before it has requested anything; there is no barrier in this kernel
    typedef typename VecOf<T, V>::type vec_t;
    typedef T lds_vec_t __attribute__((ext_vector_type(V)));
    typedef const volatile __attribute__((address_space(3))) lds_vec_t *lds_vec_ptr;
    using P = PairGeom<T, RA, RB, V>;
    using GA = typename P::GA;
    constexpr int TW = P::TW, KR = P::KR, LEAD = P::LEAD, PITCH = P::PITCH;
    constexpr int AA = GA::A, NWINA = GA::NWIN, DI = GA::DI, AB = P::AB, NWINB = P::NWINB;
    constexpr int ROWE_A = GA::ROWB / (int)sizeof(T), ROWE_B = P::ROWB_B / (int)sizeof(T);
    constexpr int WINA = 2 * RA + KR, WINB = 2 * RB + KR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int strip = bx * 4 + wave;
    const int own0 = strip * PITCH; // first column this wave stores
    if (own0 >= cols)
        return; // the whole wave, before it has requested anything; there is no barrier in this kernel
    const int own1 = min(own0 + PITCH, cols);
    const int X0 = own0 - P::MG; // first column the stages compute (negative in the first strip)
    const int ys = by * seg;
    const int ye = min(ys + seg, rows);
    const int xb = X0 + lane * V; // this lane's columns xb .. xb+V-1
    const bool lane_owns = xb >= own0 && xb < own1;
    const bool lane_full = lane_owns && xb + V <= own1;
    const T *s = src + bz * simg;
    T *da = dst_a + bz * aimg, *db = dst_b + bz * bimg;
    const int n_iter = (ye - ys + 2 * LEAD + KR - 1) / KR;
    // stage A: raw input rows in a ring filled by direct-to-LDS loads; element 0 of a row image is column cbase
    const int cbase = X0 - RA - AA;
    T *ring = reinterpret_cast<T *>(smem_raw + (size_t)wave * P::WAVE_BYTES);
    T *img_b = ring + DI * KR * ROWE_A; // KR level-1 row images; element 0 is column X0 - RB - AB
    const unsigned int lds_wave = (unsigned int)(uintptr_t)smem_raw + (unsigned int)wave * P::WAVE_BYTES;
    const unsigned int voff0 = (unsigned int)min(max(cbase + V * lane, 0), spitch - V) * (unsigned int)sizeof(T);
    const unsigned int voff1 = (unsigned int)min(max(cbase + TW + V * lane, 0), spitch - V) * (unsigned int)sizeof(T);
    const bool left_edge = cbase < 0, right_edge = X0 + TW + RA > cols; // wave-uniform
    const int e0_a = -cbase;                    // element of input column 0 in a stage-A image (left_edge)
    const int bbase = X0 - RB - AB;             // column of element 0 of a level-1 image
    const bool left_b = bbase < 0, right_b = bbase + ROWE_B > cols;
    auto stage = [&](int t, int slot) {
#pragma unroll
        for (int k = 0; k < KR; k++) {
            const int yy = reflect101_once(ys - LEAD + KR * t + k, rows);
            const unsigned long long base = (unsigned long long)(uintptr_t)(s + (size_t)yy * spitch);
            const unsigned int to = lds_wave + (unsigned int)((slot * KR + k) * GA::ROWB);
            load16_to_lds(to, base, voff0);
            if (lane < GA::NP - 64)
                load16_to_lds(to + 1024u, base, voff1);
        }
    };
    // row filter of the lane's V outputs from NW vectors of a row image (tap t of output v = window element A + v + t)
    auto row_filter = [&](const T *image, auto nw_tag, auto a_tag, auto r_tag, const Taps<T> &taps) -> vec_t {
        constexpr int NW = decltype(nw_tag)::value, A = decltype(a_tag)::value, R = decltype(r_tag)::value;
        lds_vec_t rv[NW];
#pragma unroll
        for (int q = 0; q < NW; q++)
            rv[q] = *(lds_vec_ptr)(image + lane * V + q * V);
        vec_t o;
        T *po = reinterpret_cast<T *>(&o);
#pragma unroll
        for (int q = 0; q < NW; q++) {
#pragma unroll
            for (int j = 0; j < V; j++) {
                const int e = q * V + j;
#pragma unroll
                for (int vo = 0; vo < V; vo++) {
                    const int tt = e - vo - A;
                    if (tt == 0)
                        po[vo] = taps.k[0] * rv[q][j];
                    else if (tt > 0 && tt <= 2 * R)
                        po[vo] = blur_madd<F>(po[vo], taps.k[tt], rv[q][j]);
                }
            }
        }
        return o;
    };
    // column filter on window rows k .. k + 2R (centre R + k)
    auto col_filter = [&](const vec_t *win, int k, auto r_tag, const Taps<T> &taps) -> vec_t {
        constexpr int R = decltype(r_tag)::value;
        vec_t acc;
        T *pa = reinterpret_cast<T *>(&acc);
        const T *pc = reinterpret_cast<const T *>(&win[R + k]);
#pragma unroll
        for (int e = 0; e < V; e++)
            pa[e] = taps.k[R] * pc[e];
#pragma unroll
        for (int tt = 1; tt <= R; tt++) {
            const T *pu = reinterpret_cast<const T *>(&win[R + k + tt]);
            const T *pd = reinterpret_cast<const T *>(&win[R + k - tt]);
#pragma unroll
            for (int e = 0; e < V; e++)
                pa[e] = blur_madd<F>(pa[e], taps.k[R + tt], pu[e] + pd[e]);
        }
        return acc;
    };
    auto store_row = [&](T *d, int pitch, int y, const vec_t &v) {
        if (y < ys || y >= ye || !lane_owns)
            return;
        T *dr = d + (size_t)y * pitch;
        if (lane_full) {
            *reinterpret_cast<vec_t *>(dr + xb) = v;
        } else {
            const T *pv = reinterpret_cast<const T *>(&v);
#pragma unroll
            for (int e = 0; e < V; e++)
                if (xb + e < own1)
                    dr[xb + e] = pv[e];
        }
    };

    vec_t win_a[WINA], win_b[WINB];
#pragma unroll
    for (int j = 0; j < WINA; j++) {
        T *pw = reinterpret_cast<T *>(&win_a[j]);
#pragma unroll
        for (int e = 0; e < V; e++)
            pw[e] = (T)0;
    }
#pragma unroll
    for (int j = 0; j < WINB; j++) {
        T *pw = reinterpret_cast<T *>(&win_b[j]);
#pragma unroll
        for (int e = 0; e < V; e++)
            pw[e] = (T)0;
    }
    // the level-1 images start defined: their margins are read (by lanes whose results are dropped) before anything is written there
    for (int e = lane; e < KR * ROWE_B; e += 64)
        img_b[e] = (T)0;
    for (int t = 0; t < min(DI, n_iter); t++)
        stage(t, t);
    int slot = 0;
    for (int t = 0; t < n_iter; t++) {
        wait_groups_ahead<2 * KR, DI - 1>(n_iter - 1 - t);
        T *raw = ring + slot * (KR * ROWE_A);
        if (left_edge) { // input columns -1 .. -RA from their mirrors 1 .. RA
#pragma unroll
            for (int k = 0; k < KR; k++)
                if (lane < RA) {
                    const T v = raw[k * ROWE_A + e0_a + 1 + lane];
                    raw[k * ROWE_A + e0_a - 1 - lane] = v;
                }
        }
        if (right_edge) {
            const int to = cols + lane - cbase, from = cols - 2 - lane - cbase;
#pragma unroll
            for (int k = 0; k < KR; k++)
                if (lane < RA && to < ROWE_A) {
                    const T v = raw[k * ROWE_A + from];
                    raw[k * ROWE_A + to] = v;
                }
        }
        lds_order(); // LDS operations of a wave execute in issue order: the copies above are seen by the reads below
        // ---- stage A: row filter of the two input rows, column filter -> two level-1 rows ----
#pragma unroll
        for (int j = 0; j + KR < WINA; j++)
            win_a[j] = win_a[j + KR];
#pragma unroll
        for (int k = 0; k < KR; k++)
            win_a[2 * RA + k] = row_filter(raw + k * ROWE_A, std::integral_constant<int, NWINA>{},
                                           std::integral_constant<int, AA>{}, std::integral_constant<int, RA>{}, taps_a);
        wait_lgkm0(); // the slot's reads have returned: it is loaded again
        if (t + DI < n_iter)
            stage(t + DI, slot);
        slot = slot + 1 == DI ? 0 : slot + 1;
        const int y1 = ys - LEAD - RA + KR * t; // level-1 row of k = 0 (a reflected index outside the plane)
#pragma unroll
        for (int k = 0; k < KR; k++) {
            const vec_t a = col_filter(win_a, k, std::integral_constant<int, RA>{}, taps_a);
            store_row(da, apitch, y1 + k, a);
            *reinterpret_cast<vec_t *>(img_b + k * ROWE_B + RB + AB + lane * V) = a;
        }
        lds_order(); // the other lanes' level-1 values are read below; in-order LDS needs no wait for that
        if (left_b) { // level-1 columns -1 .. -RB from their mirrors 1 .. RB
#pragma unroll
            for (int k = 0; k < KR; k++)
                if (lane < RB) {
                    const T v = img_b[k * ROWE_B - bbase + 1 + lane];
                    img_b[k * ROWE_B - bbase - 1 - lane] = v;
                }
        }
        if (right_b) { // level-1 columns cols .. cols+RB-1 from cols-2 .. cols-1-RB
            const int to = cols + lane - bbase, from = cols - 2 - lane - bbase;
#pragma unroll
            for (int k = 0; k < KR; k++)
                if (lane < RB && to < ROWE_B && from >= 0) {
                    const T v = img_b[k * ROWE_B + from];
                    img_b[k * ROWE_B + to] = v;
                }
        }
        lds_order();
        // ---- stage B: row filter of the two level-1 rows, column filter -> two level-2 rows ----
#pragma unroll
        for (int j = 0; j + KR < WINB; j++)
            win_b[j] = win_b[j + KR];
#pragma unroll
        for (int k = 0; k < KR; k++)
            win_b[2 * RB + k] = row_filter(img_b + k * ROWE_B, std::integral_constant<int, NWINB>{},
                                           std::integral_constant<int, AB>{}, std::integral_constant<int, RB>{}, taps_b);
        lds_order(); // the images are rewritten in the next iteration, after these reads in issue order
        const int y2 = y1 - RB;
#pragma unroll
        for (int k = 0; k < KR; k++) {
            const vec_t b = col_filter(win_b, k, std::integral_constant<int, RB>{}, taps_b);
            store_row(db, bpitch, y2 + k, b);
        }
    }
}

// true when blur_dma_k applies: 16-byte aligned rows (SIFTGPU_BLUR_DIRECT=1 keeps the register form everywhere)
template <typename T, int V>
static bool blur_dma_ok(const Plane &src)
{
    if (env().blur_direct)
        return false;
    return (uintptr_t)src.p % 16 == 0 && src.pitch % V == 0 && src.pitch >= V &&
           (src.img_stride * sizeof(T)) % 16 == 0 && (size_t)src.pitch * sizeof(T) < (1ull << 31);
}

// ---- any-radius fallback (one thread per output pixel, taps in global memory) ------------------------
template <typename T, bool F = false>
__global__ __launch_bounds__(256) void blur_row_any_k(const T *src, T *dst, int rows, int cols, int spitch,
                                                      int dpitch, size_t simg, size_t dimg, const T *taps, int n)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows)
        return;
    const T *s = src + blockIdx.z * simg + (size_t)y * spitch;
    int h = n / 2;
    T acc = taps[0] * s[reflect101(x - h, cols)];
    for (int t = 1; t < n; t++)
        acc = blur_madd<F>(acc, taps[t], s[reflect101(x - h + t, cols)]);
    dst[blockIdx.z * dimg + (size_t)y * dpitch + x] = acc;
}

template <typename T, bool F = false>
__global__ __launch_bounds__(256) void blur_col_any_k(const T *src, T *dst, int rows, int cols, int spitch,
                                                      int dpitch, size_t simg, size_t dimg, const T *taps, int n)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows)
        return;
    const T *s = src + blockIdx.z * simg + x;
    int h = n / 2;
    T acc = taps[h] * s[(size_t)y * spitch];
    for (int t = 1; t <= h; t++)
        acc = blur_madd<F>(acc, taps[h + t], s[(size_t)reflect101(y + t, rows) * spitch] + s[(size_t)reflect101(y - t, rows) * spitch]);
    dst[blockIdx.z * dimg + (size_t)y * dpitch + x] = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void copy_plane_k(const T *src, T *dst, int rows, int cols, int spitch, int dpitch,
                                                    size_t simg, size_t dimg)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows)
        return;
    dst[blockIdx.z * dimg + (size_t)y * dpitch + x] = src[blockIdx.z * simg + (size_t)y * spitch + x];
}

// ---- pyramid tail: the smallest octaves in one launch ---------------------------------------------------------
// Below a few thousand pixels a blur launch is all latency: five dependent launches per octave, and a frame alone
// spends more time in the last four octaves' launches than in the first octave's arithmetic.  Here one workgroup per
// image keeps the current level in LDS (A), filters rows into a second plane (Bp) and columns back into A, writing
// every level to its place in the pyramid on the way; the 2:1 decimation of level 3 goes out as the next octave's
// level 0 and is read back when that octave starts.  The arithmetic and its order are those of the row and column
// kernels above (borders by repeated reflection, as planes may be narrower than the kernel), so levels are
// bit-identical to theirs.
constexpr int kTailThreads = 1024;
constexpr size_t kTailLdsBytes = 150u << 10; // of the 160 KB a workgroup may take

template <typename T>
struct TailTaps {
    T k[kLevels - 1][kMaxTaps]; // level l: 2 * R[l-1] + 1 taps, the kernel centred and zero-padded to that width
    int R[kLevels - 1];         // 5, 6, 8, 10 or 13
};

constexpr int kTailHalo = 14; // columns of reflected border kept on either side of a row of A (R <= 13, pairs: +1)

// Reflected border columns of A (rows x cols pixels at pitch cols + 2 * kTailHalo, first pixel of a row at +kTailHalo)
template <typename T>
__device__ __forceinline__ void tail_fill_halo(T *A, int rows, int cols, int tid)
{
    const int pa = cols + 2 * kTailHalo;
    for (int idx = tid; idx < rows * 2 * kTailHalo; idx += kTailThreads) {
        const int y = idx / (2 * kTailHalo), j = idx - y * (2 * kTailHalo);
        const int c = j < kTailHalo ? j - kTailHalo : cols + (j - kTailHalo);
        A[y * pa + kTailHalo + c] = A[y * pa + kTailHalo + reflect101(c, cols)];
    }
}

// One level: rows of A (with its reflected border columns) filtered into Bp, columns of Bp filtered back into A and
// out to the pyramid.  A thread takes two horizontally adjacent pixels in the row pass and two vertically adjacent
// ones in the column pass: the pair shares all but one of its window's LDS reads, and LDS bandwidth is what bounds a
// workgroup that filters out of LDS.  Taps live in scalar registers.  Column windows that cross the top or bottom
// take reflected row indices (one reflection suffices while rows > R + 1, the general loop otherwise).
template <typename T, int R, bool F>
__device__ __forceinline__ void tail_level(T *A, T *Bp, T *out, int rows, int cols, int pitch, const T *kp, int tid)
{
    T k[2 * R + 1];
#pragma unroll
    for (int t = 0; t <= 2 * R; t++)
        k[t] = kp[t];
    const int pa = cols + 2 * kTailHalo;
    const int xpairs = (cols + 1) >> 1, ypairs = (rows + 1) >> 1;
    for (int idx = tid; idx < rows * xpairs; idx += kTailThreads) { // generic row filter, left to right
        const int y = idx / xpairs, x = 2 * (idx - y * xpairs);
        const T *p = A + y * pa + kTailHalo + (x - R);
        T w[2 * R + 2];
#pragma unroll
        for (int q = 0; q < 2 * R + 2; q++)
            w[q] = p[q];
        T acc0 = k[0] * w[0], acc1 = k[0] * w[1];
#pragma unroll
        for (int t = 1; t <= 2 * R; t++) {
            acc0 = blur_madd<F>(acc0, k[t], w[t]);
            acc1 = blur_madd<F>(acc1, k[t], w[t + 1]);
        }
        Bp[y * cols + x] = acc0;
        if (x + 1 < cols)
            Bp[y * cols + x + 1] = acc1;
    }
    __syncthreads();
    for (int idx = tid; idx < ypairs * cols; idx += kTailThreads) { // symmetric column filter, centre outwards
        const int yp = idx / cols, x = idx - yp * cols, y = 2 * yp;
        const bool two = y + 1 < rows;
        T acc0, acc1 = (T)0;
        if (rows > R + 1) {
            T w[2 * R + 2];
            if (y >= R && y + 1 + R < rows) {
#pragma unroll
                for (int q = 0; q < 2 * R + 2; q++)
                    w[q] = Bp[(y - R + q) * cols + x];
            } else {
#pragma unroll
                for (int q = 0; q < 2 * R + 2; q++)
                    w[q] = Bp[reflect101_once(y - R + q, rows) * cols + x];
            }
            acc0 = k[R] * w[R];
            acc1 = k[R] * w[R + 1];
#pragma unroll
            for (int t = 1; t <= R; t++) {
                acc0 = blur_madd<F>(acc0, k[R + t], w[R + t] + w[R - t]);
                acc1 = blur_madd<F>(acc1, k[R + t], w[R + 1 + t] + w[R + 1 - t]);
            }
        } else {
            acc0 = kp[R] * Bp[y * cols + x];
            for (int t = 1; t <= R; t++)
                acc0 = blur_madd<F>(acc0, kp[R + t],
                                 Bp[reflect101(y + t, rows) * cols + x] + Bp[reflect101(y - t, rows) * cols + x]);
            if (two) {
                acc1 = kp[R] * Bp[(y + 1) * cols + x];
                for (int t = 1; t <= R; t++)
                    acc1 = blur_madd<F>(acc1, kp[R + t], Bp[reflect101(y + 1 + t, rows) * cols + x] +
                                                          Bp[reflect101(y + 1 - t, rows) * cols + x]);
            }
        }
        A[y * pa + kTailHalo + x] = acc0;
        out[(size_t)y * pitch + x] = acc0;
        if (two) {
            A[(y + 1) * pa + kTailHalo + x] = acc1;
            out[(size_t)(y + 1) * pitch + x] = acc1;
        }
    }
    __syncthreads();
    tail_fill_halo(A, rows, cols, tid);
    __syncthreads();
}

template <typename T, bool F = false>
__global__ __launch_bounds__(kTailThreads) void pyr_tail_k(OctaveTab tab, int first, TailTaps<T> taps,
                                                           unsigned int a_elems)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *A = reinterpret_cast<T *>(smem_raw), *Bp = A + a_elems;
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int o = first; o < tab.octaves; o++) {
        const int rows = tab.rows[o], cols = tab.cols[o], pitch = tab.pitch[o], npx = rows * cols;
        const int pa = cols + 2 * kTailHalo;
        // level 0: in place for the first octave, written by this workgroup (below) for the later ones
        const T *lvl0 = (const T *)tab.lv[o][0] + (size_t)b * tab.img_stride[o];
        for (int idx = tid; idx < npx; idx += kTailThreads) {
            const int y = idx / cols, x = idx - y * cols;
            A[y * pa + kTailHalo + x] = lvl0[(size_t)y * pitch + x];
        }
        __syncthreads();
        tail_fill_halo(A, rows, cols, tid);
        __syncthreads();
        for (int l = 1; l < kLevels; l++) {
            T *out = (T *)tab.lv[o][l] + (size_t)b * tab.img_stride[o];
            const T *kp = taps.k[l - 1];
            switch (taps.R[l - 1]) {
            case 5: tail_level<T, 5, F>(A, Bp, out, rows, cols, pitch, kp, tid); break;
            case 6: tail_level<T, 6, F>(A, Bp, out, rows, cols, pitch, kp, tid); break;
            case 8: tail_level<T, 8, F>(A, Bp, out, rows, cols, pitch, kp, tid); break;
            case 10: tail_level<T, 10, F>(A, Bp, out, rows, cols, pitch, kp, tid); break;
            default: tail_level<T, 13, F>(A, Bp, out, rows, cols, pitch, kp, tid); break;
            }
            if (l == 3 && o + 1 < tab.octaves) { // DownSample: every second pixel of every second row
                const int r2 = tab.rows[o + 1], c2 = tab.cols[o + 1], p2 = tab.pitch[o + 1];
                T *next0 = (T *)tab.lv[o + 1][0] + (size_t)b * tab.img_stride[o + 1];
                for (int idx = tid; idx < r2 * c2; idx += kTailThreads) {
                    const int y = idx / c2, x = idx - y * c2;
                    next0[(size_t)y * p2 + x] = A[(2 * y) * pa + kTailHalo + 2 * x];
                }
                // read back at the top of the next octave, after the barriers of levels 4 and 5
            }
        }
    }
}

static int bucket_radius(int h);
template <typename T>
static Taps<T> padded_taps(const double *taps, int n, int R);

template <typename T>
static size_t tail_a_elems(const OctaveTab &tab, int first)
{
    return ((size_t)tab.rows[first] * (tab.cols[first] + 2 * kTailHalo) + 3) & ~(size_t)3;
}

template <typename T>
static size_t tail_lds_bytes(const OctaveTab &tab, int first)
{
    return (tail_a_elems<T>(tab, first) + (size_t)tab.rows[first] * tab.cols[first]) * sizeof(T);
}

int pyr_tail_first(const OctaveTab &tab, const int *ntaps, int prec)
{
    if (!env().pyr_tail || tab.from_dog)
        return tab.octaves;
    for (int l = 0; l < kLevels - 1; l++)
        if (ntaps[l] < 3 || ntaps[l] > kMaxTaps || ntaps[l] % 2 == 0)
            return tab.octaves;
    int first = tab.octaves;
    for (int o = tab.octaves - 1; o >= 1; o--) {
        // every octave of the tail halves the one before it (no dimension of 1, where DownSample copies and the
        // kernel of a blur shrinks to one tap) and fits the workgroup's LDS
        const bool halves = tab.rows[o] == tab.rows[o - 1] / 2 && tab.cols[o] == tab.cols[o - 1] / 2;
        if (tab.rows[o] < 2 || tab.cols[o] < 2 || !halves)
            break;
        const size_t bytes = prec_store(prec) == SG_F64 ? tail_lds_bytes<double>(tab, o) : tail_lds_bytes<float>(tab, o);
        if (bytes > kTailLdsBytes)
            break;
        first = o;
    }
    return first;
}

template <typename T, bool F>
static bool tail_dispatch(const OctaveTab &tab, int first, int B, const double *const *taps, const int *ntaps,
                          hipStream_t s)
{
    // more dynamic LDS than the default limit: allowed per device, once (0 not tried yet, 1 allowed, 2 refused)
    static unsigned char allowed[64] = {0};
    static std::mutex allowed_mutex;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return false;
    {
        std::lock_guard<std::mutex> lk(allowed_mutex);
        if (allowed[dev] == 0) {
            const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&pyr_tail_k<T, F>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)kTailLdsBytes) == hipSuccess;
            if (!ok)
                (void)hipGetLastError();
            allowed[dev] = ok ? 1 : 2;
        }
        if (allowed[dev] != 1)
            return false;
    }
    TailTaps<T> tt;
    for (int l = 0; l < kLevels - 1; l++) {
        const int R = bucket_radius(ntaps[l] / 2);
        if (R < 0)
            return false;
        tt.R[l] = R;
        const Taps<T> padded = padded_taps<T>(taps[l], ntaps[l], R);
        for (int i = 0; i < kMaxTaps; i++)
            tt.k[l][i] = padded.k[i];
    }
    pyr_tail_k<T, F><<<dim3(B), kTailThreads, tail_lds_bytes<T>(tab, first), s>>>(tab, first, tt,
                                                                              (unsigned int)tail_a_elems<T>(tab, first));
    return true;
}

bool launch_pyr_tail(const OctaveTab &tab, int first, int B, const double *const *taps, const int *ntaps, int prec,
                     hipStream_t s)
{
    if (first < 1 || first >= tab.octaves || B < 1)
        return false;
    if (prec_store(prec) != SG_F64)
        return tail_dispatch<float, false>(tab, first, B, taps, ntaps, s);
    return prec_fma(prec) ? tail_dispatch<double, true>(tab, first, B, taps, ntaps, s)
                          : tail_dispatch<double, false>(tab, first, B, taps, ntaps, s);
}

// ---- dispatch ---------------------------------------------------------------------------------------

static int bucket_radius(int h)
{
    static const int buckets[] = {5, 6, 8, 10, 13};
    for (int b : buckets)
        if (h <= b)
            return b;
    return -1;
}

template <typename T>
static Taps<T> padded_taps(const double *taps, int n, int R)
{
    Taps<T> t;
    for (int i = 0; i < kMaxTaps; i++)
        t.k[i] = (T)0;
    int h = n / 2;
    for (int i = 0; i < n; i++)
        t.k[R - h + i] = (T)taps[i];
    return t;
}

// Device copy of the taps for the any-radius fallback (rare: sigma outside the pyramid's own set).
template <typename T>
static T *upload_taps(const double *taps, int n, hipStream_t s)
{
    T host[kMaxAnyTaps];
    if (n > kMaxAnyTaps)
        return nullptr;
    for (int i = 0; i < n; i++)
        host[i] = (T)taps[i];
    T *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(T) * n) != hipSuccess)
        return nullptr;
    hipMemcpyAsync(d, host, sizeof(T) * n, hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);
    return d;
}

template <typename T, int V, int U, bool F>
static bool row_dispatch(const Plane &src, const Plane &dst, int B, const double *taps, int n, hipStream_t s,
                         const void *d_taps)
{
    const int rows = src.rows, cols = src.cols;
    dim3 g2((cols + 63) / 64, (rows + 3) / 4, B);
    if (n <= 1 || cols == 1) { // a 1-pixel dimension shrinks the kernel to one tap
        copy_plane_k<T><<<g2, 256, 0, s>>>((const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch, src.img_stride,
                                           dst.img_stride);
        return true;
    }
    int R = bucket_radius(n / 2);
    constexpr int TW = 64 * V * U;
    dim3 g((cols + TW - 1) / TW, (rows + 3) / 4, B);
#define SG_ROW_CASE(RR)                                                                                               \
    case RR:                                                                                                          \
        blur_row_k<T, RR, V, U, F><<<g, 256, 0, s>>>((const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch,     \
                                                  src.img_stride, dst.img_stride, padded_taps<T>(taps, n, RR));       \
        break;
    switch (R) {
        SG_ROW_CASE(5)
        SG_ROW_CASE(6)
        SG_ROW_CASE(8)
        SG_ROW_CASE(10)
        SG_ROW_CASE(13)
    default: {
        if (d_taps) {
            blur_row_any_k<T, F><<<g2, 256, 0, s>>>((const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch,
                                                 src.img_stride, dst.img_stride, (const T *)d_taps, n);
        } else {
            T *dt = upload_taps<T>(taps, n, s);
            if (!dt)
                return false; // no device memory for the taps (or more taps than the any-radius kernels take)
            blur_row_any_k<T, F><<<g2, 256, 0, s>>>((const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch,
                                                 src.img_stride, dst.img_stride, dt, n);
            hipStreamSynchronize(s);
            hipFree(dt);
        }
    }
    }
#undef SG_ROW_CASE
    return true;
}

template <typename T, int V, int RPT, bool F>
static bool col_dispatch(const Plane &src, const Plane &dst, int B, const double *taps, int n, hipStream_t s,
                         const void *d_taps)
{
    const int rows = src.rows, cols = src.cols;
    dim3 g2((cols + 63) / 64, (rows + 3) / 4, B);
    if (n <= 1 || rows == 1) {
        copy_plane_k<T><<<g2, 256, 0, s>>>((const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch, src.img_stride,
                                           dst.img_stride);
        return true;
    }
    int R = bucket_radius(n / 2);
    constexpr int TW = 64 * V, TH = 4 * RPT;
    dim3 g((cols + TW - 1) / TW, (rows + TH - 1) / TH, B);
#define SG_COL_CASE(RR)                                                                                               \
    case RR: {                                                                                                        \
        size_t lds = sizeof(T) * (size_t)(TH + 2 * RR) * TW;                                                          \
        blur_col_k<T, RR, V, RPT, F><<<g, 256, lds, s>>>((const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch, \
                                                      src.img_stride, dst.img_stride, padded_taps<T>(taps, n, RR));   \
    } break;
    switch (R) {
        SG_COL_CASE(5)
        SG_COL_CASE(6)
        SG_COL_CASE(8)
        SG_COL_CASE(10)
        SG_COL_CASE(13)
    default: {
        if (d_taps) {
            blur_col_any_k<T, F><<<g2, 256, 0, s>>>((const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch,
                                                 src.img_stride, dst.img_stride, (const T *)d_taps, n);
        } else {
            T *dt = upload_taps<T>(taps, n, s);
            if (!dt)
                return false; // no device memory for the taps (or more taps than the any-radius kernels take)
            blur_col_any_k<T, F><<<g2, 256, 0, s>>>((const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch,
                                                 src.img_stride, dst.img_stride, dt, n);
            hipStreamSynchronize(s);
            hipFree(dt);
        }
    }
    }
#undef SG_COL_CASE
    return true;
}

bool blur_fused_supported(const Plane &src, int n)
{
    if (n <= 1)
        return false;
    const int R = bucket_radius(n / 2);
    // the fused kernels reflect an index once: planes not larger than the radius go through the two-pass kernels
    return R > 0 && src.rows > R && src.cols > R;
}

// Rows per wave segment of the register-window kernel.  A wave runs for its whole segment, so the launch is sized
// from the wave slots of the chip (CUs x 4 SIMDs x resident waves per SIMD at this radius' register count): as many
// segments per strip as give about two waves per slot (measured best: one wave per slot leaves the chip idle while
// the slowest waves finish, many more re-read too many halo rows); never so short that the 2R halo rows (re-read
// and row-filtered again) dominate.  per_row = strips x frames.
static int win_segment_rows(int rows, long long per_row, int R, int waves_per_simd)
{
    const int n_cu = env().n_cu;
    const long long slots = (long long)n_cu * 4 * waves_per_simd * 2;
    int nseg_want = (int)(slots / per_row);
    if (nseg_want < 1)
        nseg_want = 1;
    int seg = (rows + nseg_want - 1) / nseg_want;
    // Shortest segment: 4R rows keep the re-filtered halo rows at a third of the work.  A launch that cannot give every
    // SIMD a wave even then (one frame alone, the small octaves) is bound by the chain of dependent iterations of
    // its waves, not by work: there the floor drops to the length that gives one wave per SIMD, down to 8 rows.
    int floor_rows = 4 * R;
    const long long one_per_simd = (long long)n_cu * 4;
    if ((long long)((rows + floor_rows - 1) / floor_rows) * per_row < one_per_simd) {
        const long long s1 = ((long long)rows * per_row + one_per_simd - 1) / one_per_simd;
        floor_rows = (int)std::min<long long>(floor_rows, std::max<long long>(8, (s1 + 3) & ~3ll));
    }
    if (seg < floor_rows)
        seg = floor_rows;
    if (seg > rows)
        seg = rows;
    return (seg + 1) & ~1; // even, so that row parity inside a segment is the image's
}

// SIFTGPU_BLUR=ring selects the workgroup-shared LDS-ring variant of the fused kernel (kept for comparison); the
// register-window variant is the default for every radius.
static bool blur_use_ring(int R, size_t elem)
{
    (void)R;
    (void)elem;
    return env().blur_mode == 2;
}

template <typename T, int V, bool F>
static bool fused_dispatch(const Plane &src, const Plane &dst, int B, const double *taps, int n, const Plane *dec,
                           hipStream_t s)
{
    const int rows = src.rows, cols = src.cols;
    if (!blur_fused_supported(src, n))
        return false;
    const int R = bucket_radius(n / 2);
    constexpr int TW = 64 * V;
    const int strips = (cols + TW - 1) / TW;
    const long long per_row = (long long)strips * B;
    if (blur_use_ring(R, sizeof(T))) {
        if (dec)
            return false; // only the register-window kernel writes the decimated copy (blur_fused_decimates)
        // workgroup-shared LDS ring variant
        constexpr int KF = 4; // rows per wave per round
        long long want = 1536;
        int seg = (int)(((long long)rows * per_row + want - 1) / want);
        if (seg < 16)
            seg = 16;
        if (seg > rows)
            seg = rows;
        seg = (seg + 4 * KF - 1) / (4 * KF) * (4 * KF);
        dim3 g(strips, (rows + seg - 1) / seg, B);
#define SG_FUSED_CASE(RR)                                                                                             \
    case RR:                                                                                                          \
        blur_fused_k<T, RR, V, KF><<<g, 256, fused_lds_bytes<T, RR, V, KF>(), s>>>(                                   \
            (const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch, src.img_stride, dst.img_stride, seg,      \
            padded_taps<T>(taps, n, RR));                                                                             \
        break;
        switch (R) {
            SG_FUSED_CASE(5)
            SG_FUSED_CASE(6)
            SG_FUSED_CASE(8)
            SG_FUSED_CASE(10)
            SG_FUSED_CASE(13)
        default:
            return false;
        }
#undef SG_FUSED_CASE
        return true;
    }
    const bool dma = blur_dma_ok<T, V>(src);
    const int seg = win_segment_rows(rows, per_row, R, dma ? dma_waves_per_simd(R) : win_waves_per_simd(R));
    const int nseg = (rows + seg - 1) / seg;
    T *dec_p = dec ? (T *)dec->p : nullptr;
    const int dec_rows = dec ? dec->rows : 0, dec_cols = dec ? dec->cols : 0, dec_pitch = dec ? dec->pitch : 0;
    const size_t dec_img = dec ? dec->img_stride : 0;
    const TileGrid tg = {(strips + 3) / 4, nseg, B, env().blur_remap};
    const dim3 g = tile_launch_grid(tg);
#define SG_WIN_CASE(RR)                                                                                               \
    case RR:                                                                                                          \
        if (dma)                                                                                                      \
            blur_dma_k<T, RR, V, F><<<g, 256, 4 * WinDma<T, RR, V>::WAVE_BYTES, s>>>(                                 \
                (const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch, src.img_stride, dst.img_stride, seg,  \
                padded_taps<T>(taps, n, RR), dec_p, dec_rows, dec_cols, dec_pitch, dec_img, tg);                      \
        else                                                                                                          \
            blur_win_k<T, RR, V, false, false, F><<<g, 256, win_lds_bytes<T, RR, V>(), s>>>(                          \
                (const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch, src.img_stride, dst.img_stride, seg,  \
                padded_taps<T>(taps, n, RR), dec_p, dec_rows, dec_cols, dec_pitch, dec_img, tg);                      \
        break;
    switch (R) {
        SG_WIN_CASE(5)
        SG_WIN_CASE(6)
        SG_WIN_CASE(8)
        SG_WIN_CASE(10)
        SG_WIN_CASE(13)
    default:
        return false;
    }
#undef SG_WIN_CASE
    return true;
}

// Upsampling form: `gray` is a double plane of dst.rows/2 x dst.cols/2; dst = blur(UpSample(gray)).  Instantiated
// for the radius of the pyramid's base blur only.
template <typename T, int V, bool F>
static bool ups_dispatch(const Plane &gray, const Plane &dst, int B, const double *taps, int n, hipStream_t s)
{
    constexpr int R = 5;
    const int rows = dst.rows, cols = dst.cols;
    if (rows != 2 * gray.rows || cols != 2 * gray.cols || gray.rows < 2 || gray.cols < 2 || n <= 1 ||
        bucket_radius(n / 2) != R || rows <= R || cols <= R)
        return false;
    constexpr int TW = 64 * V;
    const int strips = (cols + TW - 1) / TW;
    const long long per_row = (long long)strips * B;
    const int seg = win_segment_rows(rows, per_row, R, win_waves_per_simd(R));
    const int nseg = (rows + seg - 1) / seg;
    const TileGrid tg = {(strips + 3) / 4, nseg, B, env().blur_remap};
    const dim3 g = tile_launch_grid(tg);
    // gray rows staged by direct-to-LDS loads when they are 16-byte aligned (SIFTGPU_BLUR_DIRECT=1: loaded by the lanes)
    const bool gdma = !env().blur_direct && (uintptr_t)gray.p % 16 == 0 && gray.pitch % 2 == 0 &&
                      gray.pitch >= 2 && (gray.img_stride * sizeof(double)) % 16 == 0;
    if (gdma)
        blur_win_k<T, R, V, true, true, F>
            <<<g, 256, ((win_lds_bytes<T, R, V>() + 15) & ~(size_t)15) + 4 * WinGray<T, R, V>::WAVE_BYTES, s>>>(
                (const T *)gray.p, (T *)dst.p, rows, cols, gray.pitch, dst.pitch, gray.img_stride, dst.img_stride, seg,
                padded_taps<T>(taps, n, R), nullptr, 0, 0, 0, 0, tg);
    else
        blur_win_k<T, R, V, true, false, F><<<g, 256, win_lds_bytes<T, R, V>(), s>>>(
            (const T *)gray.p, (T *)dst.p, rows, cols, gray.pitch, dst.pitch, gray.img_stride, dst.img_stride, seg,
            padded_taps<T>(taps, n, R), nullptr, 0, 0, 0, 0, tg);
    return true;
}

bool launch_blur_fused_upsampled(const Plane &gray, const Plane &dst, int B, const double *taps, int n, int prec,
                                 hipStream_t s)
{
    if (env().blur_mode == 1 || env().blur_mode == 2)
        return false; // another variant is forced
    if (prec_store(prec) != SG_F64)
        return ups_dispatch<float, 4, false>(gray, dst, B, taps, n, s);
    return prec_fma(prec) ? ups_dispatch<double, 2, true>(gray, dst, B, taps, n, s)
                          : ups_dispatch<double, 2, false>(gray, dst, B, taps, n, s);
}

bool launch_blur_fused(const Plane &src, const Plane &dst, int B, const double *taps, int n, int prec, hipStream_t s,
                       const Plane *dec)
{
    if (prec_store(prec) != SG_F64)
        return fused_dispatch<float, 4, false>(src, dst, B, taps, n, dec, s);
    return prec_fma(prec) ? fused_dispatch<double, 2, true>(src, dst, B, taps, n, dec, s)
                          : fused_dispatch<double, 2, false>(src, dst, B, taps, n, dec, s);
}

// Levels a and b = blur(a) of one plane in one launch (blur_pair_k): for the 11- and 13-tap pair of the pyramid's
// levels 1 and 2, rows staged by LDS-DMA, and launches whose segments are long enough that the RA + RB rows a segment
// starts early do not matter.  False (nothing launched) otherwise: the caller then blurs level by level.
// (The 21- and 27-tap pair of levels 4 and 5 was measured too: its two register windows need more than the 256
// registers of two waves per SIMD -- 85 ms per 256 frames with the spills, 35 ms at one wave per SIMD, against 19.5 ms
// for the two single-level launches -- so the wide levels stay apart.)
template <typename T, int V, bool F>
static bool pair_dispatch(const Plane &src, const Plane &dst_a, const Plane &dst_b, int B, const double *taps_a, int n_a,
                          const double *taps_b, int n_b, hipStream_t s)
{
    constexpr int RA = 5, RB = 6;
    using P = PairGeom<T, RA, RB, V>;
    if (!env().blur_pair || env().blur_mode != 0 || n_a <= 1 || n_b <= 1 || bucket_radius(n_a / 2) != RA ||
        bucket_radius(n_b / 2) != RB)
        return false;
    const int rows = src.rows, cols = src.cols;
    if (rows < 4 * P::LEAD || cols < 4 * P::LEAD || !blur_dma_ok<T, V>(src) || dst_a.pitch % V || dst_b.pitch % V ||
        (uintptr_t)dst_a.p % 16 || (uintptr_t)dst_b.p % 16 || (dst_a.img_stride * sizeof(T)) % 16 ||
        (dst_b.img_stride * sizeof(T)) % 16)
        return false;
    const int strips = (cols + P::PITCH - 1) / P::PITCH;
    const int seg = win_segment_rows(rows, (long long)strips * B, P::LEAD, SG_PAIR_WPE);
    if (seg < 8 * P::LEAD && seg < rows && env().blur_pair != 2)
        return false; // short segments: the early rows would be a large share of the work
    const int nseg = (rows + seg - 1) / seg;
    const TileGrid tg = {(strips + 3) / 4, nseg, B, env().blur_remap};
    blur_pair_k<T, RA, RB, V, F><<<tile_launch_grid(tg), 256, 4 * P::WAVE_BYTES, s>>>(
        (const T *)src.p, (T *)dst_a.p, (T *)dst_b.p, rows, cols, src.pitch, dst_a.pitch, dst_b.pitch, src.img_stride,
        dst_a.img_stride, dst_b.img_stride, seg, padded_taps<T>(taps_a, n_a, RA), padded_taps<T>(taps_b, n_b, RB), tg);
    return true;
}

bool launch_blur_pair(const Plane &src, const Plane &dst_a, const Plane &dst_b, int B, const double *taps_a, int n_a,
                      const double *taps_b, int n_b, int prec, hipStream_t s)
{
    if (prec_store(prec) != SG_F64)
        return pair_dispatch<float, 4, false>(src, dst_a, dst_b, B, taps_a, n_a, taps_b, n_b, s);
    return prec_fma(prec) ? pair_dispatch<double, 2, true>(src, dst_a, dst_b, B, taps_a, n_a, taps_b, n_b, s)
                          : pair_dispatch<double, 2, false>(src, dst_a, dst_b, B, taps_a, n_a, taps_b, n_b, s);
}

bool blur_fused_decimates(const Plane &src, int n, int prec)
{
    return blur_fused_supported(src, n) && src.rows > 1 && src.cols > 1 &&
           !blur_use_ring(bucket_radius(n / 2), prec_store(prec) == SG_F64 ? 8 : 4);
}

bool launch_blur_row(const Plane &src, const Plane &dst, int B, const double *taps, int n, int prec, hipStream_t s,
                     const void *d_taps)
{
    if (prec_store(prec) != SG_F64)
        return row_dispatch<float, 4, 1, false>(src, dst, B, taps, n, s, d_taps);
    return prec_fma(prec) ? row_dispatch<double, 2, 2, true>(src, dst, B, taps, n, s, d_taps)
                          : row_dispatch<double, 2, 2, false>(src, dst, B, taps, n, s, d_taps);
}

bool launch_blur_col(const Plane &src, const Plane &dst, int B, const double *taps, int n, int prec, hipStream_t s,
                     const void *d_taps)
{
    if (prec_store(prec) != SG_F64)
        return col_dispatch<float, 4, 8, false>(src, dst, B, taps, n, s, d_taps);
    return prec_fma(prec) ? col_dispatch<double, 2, 8, true>(src, dst, B, taps, n, s, d_taps)
                          : col_dispatch<double, 2, 8, false>(src, dst, B, taps, n, s, d_taps);
}

} // namespace sg
