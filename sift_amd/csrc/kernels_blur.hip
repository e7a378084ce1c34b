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

template <typename T, int V>
struct VecOf;
template <>
struct VecOf<double, 2> { typedef double2 type; };
template <>
struct VecOf<float, 4> { typedef float4 type; };

// ---- row pass -------------------------------------------------------------------------------------
// Block = 4 waves, one row segment of TW = 64*V*U pixels per wave: each lane produces U groups of V
// adjacent outputs from a (TW + 2R)-pixel LDS image of the row segment.
template <typename T, int R, int V, int U>
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
                acc[v] += taps.k[t] * w[v + t];
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
template <typename T, int R, int V, int RPT>
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
                acc += taps.k[R + t] * (win[r + R + t][c] + win[r + R - t][c]);
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

// ---- any-radius fallback (one thread per output pixel, taps in global memory) ------------------------
template <typename T>
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
        acc += taps[t] * s[reflect101(x - h + t, cols)];
    dst[blockIdx.z * dimg + (size_t)y * dpitch + x] = acc;
}

template <typename T>
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
        acc += taps[h + t] * (s[(size_t)reflect101(y + t, rows) * spitch] + s[(size_t)reflect101(y - t, rows) * spitch]);
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
    T host[1024];
    if (n > 1024)
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

template <typename T, int V, int U>
static void row_dispatch(const Plane &src, const Plane &dst, int B, const double *taps, int n, hipStream_t s)
{
    const int rows = src.rows, cols = src.cols;
    dim3 g2((cols + 63) / 64, (rows + 3) / 4, B);
    if (n <= 1 || cols == 1) { // a 1-pixel dimension shrinks the kernel to one tap
        copy_plane_k<T><<<g2, 256, 0, s>>>((const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch, src.img_stride,
                                           dst.img_stride);
        return;
    }
    int R = bucket_radius(n / 2);
    constexpr int TW = 64 * V * U;
    dim3 g((cols + TW - 1) / TW, (rows + 3) / 4, B);
#define SG_ROW_CASE(RR)                                                                                               \
    case RR:                                                                                                          \
        blur_row_k<T, RR, V, U><<<g, 256, 0, s>>>((const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch,     \
                                                  src.img_stride, dst.img_stride, padded_taps<T>(taps, n, RR));       \
        break;
    switch (R) {
        SG_ROW_CASE(5)
        SG_ROW_CASE(6)
        SG_ROW_CASE(8)
        SG_ROW_CASE(10)
        SG_ROW_CASE(13)
    default: {
        T *dt = upload_taps<T>(taps, n, s);
        blur_row_any_k<T><<<g2, 256, 0, s>>>((const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch,
                                             src.img_stride, dst.img_stride, dt, n);
        hipStreamSynchronize(s);
        hipFree(dt);
    }
    }
#undef SG_ROW_CASE
}

template <typename T, int V, int RPT>
static void col_dispatch(const Plane &src, const Plane &dst, int B, const double *taps, int n, hipStream_t s)
{
    const int rows = src.rows, cols = src.cols;
    dim3 g2((cols + 63) / 64, (rows + 3) / 4, B);
    if (n <= 1 || rows == 1) {
        copy_plane_k<T><<<g2, 256, 0, s>>>((const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch, src.img_stride,
                                           dst.img_stride);
        return;
    }
    int R = bucket_radius(n / 2);
    constexpr int TW = 64 * V, TH = 4 * RPT;
    dim3 g((cols + TW - 1) / TW, (rows + TH - 1) / TH, B);
#define SG_COL_CASE(RR)                                                                                               \
    case RR: {                                                                                                        \
        size_t lds = sizeof(T) * (size_t)(TH + 2 * RR) * TW;                                                          \
        blur_col_k<T, RR, V, RPT><<<g, 256, lds, s>>>((const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch, \
                                                      src.img_stride, dst.img_stride, padded_taps<T>(taps, n, RR));   \
    } break;
    switch (R) {
        SG_COL_CASE(5)
        SG_COL_CASE(6)
        SG_COL_CASE(8)
        SG_COL_CASE(10)
        SG_COL_CASE(13)
    default: {
        T *dt = upload_taps<T>(taps, n, s);
        blur_col_any_k<T><<<g2, 256, 0, s>>>((const T *)src.p, (T *)dst.p, rows, cols, src.pitch, dst.pitch,
                                             src.img_stride, dst.img_stride, dt, n);
        hipStreamSynchronize(s);
        hipFree(dt);
    }
    }
#undef SG_COL_CASE
}

void launch_blur_row(const Plane &src, const Plane &dst, int B, const double *taps, int n, int prec, hipStream_t s)
{
    if (prec == SG_F64)
        row_dispatch<double, 2, 2>(src, dst, B, taps, n, s);
    else
        row_dispatch<float, 4, 1>(src, dst, B, taps, n, s);
}

void launch_blur_col(const Plane &src, const Plane &dst, int B, const double *taps, int n, int prec, hipStream_t s)
{
    if (prec == SG_F64)
        col_dispatch<double, 2, 8>(src, dst, B, taps, n, s);
    else
        col_dispatch<float, 4, 8>(src, dst, B, taps, n, s);
}

} // namespace sg
