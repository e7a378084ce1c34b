// kernels_image.hip -- image preparation kernels: gray conversion, 2x upsample, 2x decimation,
// level difference, synthetic test image.  gfx950 only.
#include "sg_internal.h"

namespace sg {

// ConvertToGray arithmetic (reference sift/sift.cpp:42-46): each channel /255.0, then
// b*0.114 + g*0.587 + r*0.299 in that order, all in double.
__device__ __forceinline__ double gray_of(const uint8_t *p)
{
    double b = p[0] / 255.0;
    double g = p[1] / 255.0;
    double r = p[2] / 255.0;
    return b * 0.114 + g * 0.587 + r * 0.299;
}

// A channel value has 256 possibilities, so each block tabulates (v/255.0)*weight for the three channels once (the
// same operations on the same operands as gray_of, hence the same doubles) and a pixel costs three LDS reads and two
// additions in gray_of's order.  A thread takes four neighbouring pixels of a row: with three interleaved channels
// and 4-byte aligned rows (QUADS) their 12 bytes are three dword loads.
template <typename T, bool QUADS>
__global__ __launch_bounds__(256) void gray_k(const uint8_t *src, int rows, int cols, int ch, size_t stride,
                                              size_t img_stride, T *dst, int dpitch, size_t dimg, int vec_store)
{
    __shared__ double tab[3][256];
    {
        const double v = threadIdx.x / 255.0;
        tab[0][threadIdx.x] = v * 0.114;
        tab[1][threadIdx.x] = v * 0.587;
        tab[2][threadIdx.x] = v * 0.299;
    }
    __syncthreads();
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows)
        return;
    const uint8_t *s = src + blockIdx.z * img_stride + stride * (size_t)y + (size_t)ch * x;
    T *d = dst + blockIdx.z * dimg + (size_t)y * dpitch + x;
    T out[4];
    if (QUADS && x + 4 <= cols) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(s);
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2]; // b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
        out[0] = (T)(tab[0][w0 & 255u] + tab[1][(w0 >> 8) & 255u] + tab[2][(w0 >> 16) & 255u]);
        out[1] = (T)(tab[0][w0 >> 24] + tab[1][w1 & 255u] + tab[2][(w1 >> 8) & 255u]);
        out[2] = (T)(tab[0][(w1 >> 16) & 255u] + tab[1][w1 >> 24] + tab[2][w2 & 255u]);
        out[3] = (T)(tab[0][(w2 >> 8) & 255u] + tab[1][(w2 >> 16) & 255u] + tab[2][w2 >> 24]);
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (x + q < cols)
                out[q] = (T)(tab[0][s[ch * q]] + tab[1][s[ch * q + 1]] + tab[2][s[ch * q + 2]]);
    }
    // vec_store: rows start on 16-byte boundaries and dpitch is a multiple of four elements, like x: the four outputs
    // are one aligned group (the pad columns behind a row's end may be written)
    if (vec_store && x + 4 <= dpitch) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (x + q >= cols)
                out[q] = (T)0;
        if (sizeof(T) == 8) {
            reinterpret_cast<double2 *>(d)[0] = make_double2((double)out[0], (double)out[1]);
            reinterpret_cast<double2 *>(d)[1] = make_double2((double)out[2], (double)out[3]);
        } else {
            reinterpret_cast<float4 *>(d)[0] = make_float4((float)out[0], (float)out[1], (float)out[2], (float)out[3]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (x + q < cols)
                d[q] = out[q];
    }
}

// UpSample (reference sift/sift.cpp:111-175).  One thread per source cell (i,j), i < R-1, j < C-1,
// writes the 2x2 output block; the cells of the last source row/column also write the two
// replicated output rows/columns (copies of output row 2R-3 / column 2C-3).
template <typename T, typename Src>
__device__ __forceinline__ void upsample_cell(const Src &src, int i, int j, int R, int C, T *d, int dpitch)
{
    double s00 = src(i, j), s10 = src(i + 1, j), s01 = src(i, j + 1), s11 = src(i + 1, j + 1);
    T v00 = (T)s00;
    T v10 = (T)((s00 + s10) / 2);
    T v01 = (T)((s00 + s01) / 2);
    T v11 = (T)((s10 + s00 + s11 + s01) / 4);
    int m = 2 * i, n = 2 * j;
    bool lastc = (j == C - 2), lastr = (i == R - 2);
    T *r0 = d + (size_t)m * dpitch + n;
    T *r1 = r0 + dpitch;
    r0[0] = v00; r0[1] = v01;
    r1[0] = v10; r1[1] = v11;
    if (lastc) {
        r0[2] = v01; r0[3] = v01;
        r1[2] = v11; r1[3] = v11;
    }
    if (lastr) {
        T *r2 = r1 + dpitch, *r3 = r2 + dpitch;
        r2[0] = v10; r2[1] = v11;
        r3[0] = v10; r3[1] = v11;
        if (lastc) {
            r2[2] = v11; r2[3] = v11;
            r3[2] = v11; r3[3] = v11;
        }
    }
}

template <typename T>
struct PlaneSrc {
    const T *p;
    int pitch;
    __device__ double operator()(int i, int j) const { return (double)p[(size_t)i * pitch + j]; }
};

struct GraySrc {
    const uint8_t *p;
    size_t stride;
    int ch;
    __device__ double operator()(int i, int j) const { return gray_of(p + stride * (size_t)i + (size_t)ch * j); }
};

template <typename T>
__global__ __launch_bounds__(256) void upsample_k(const T *src, int R, int C, int spitch, size_t simg, T *dst,
                                                  int dpitch, size_t dimg)
{
    int j = blockIdx.x * 64 + (threadIdx.x & 63);
    int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= R - 1 || j >= C - 1)
        return;
    PlaneSrc<T> s{src + blockIdx.z * simg, spitch};
    upsample_cell<T>(s, i, j, R, C, dst + blockIdx.z * dimg, dpitch);
}

// In the fused form a float pyramid still takes its gray values in double and rounds once.
template <typename T>
__global__ __launch_bounds__(256) void gray_upsample_k(const uint8_t *src, int R, int C, int ch, size_t stride,
                                                       size_t img_stride, T *dst, int dpitch, size_t dimg)
{
    int j = blockIdx.x * 64 + (threadIdx.x & 63);
    int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= R - 1 || j >= C - 1)
        return;
    GraySrc s{src + blockIdx.z * img_stride, stride, ch};
    upsample_cell<T>(s, i, j, R, C, dst + blockIdx.z * dimg, dpitch);
}

// DownSample (reference sift/sift.cpp:59-103): dst(m,n) = src(2m,2n).
template <typename T>
__global__ __launch_bounds__(256) void downsample_k(const T *src, int spitch, size_t simg, T *dst, int drows,
                                                    int dcols, int dpitch, size_t dimg)
{
    int n = blockIdx.x * 64 + (threadIdx.x & 63);
    int m = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (m >= drows || n >= dcols)
        return;
    dst[blockIdx.z * dimg + (size_t)m * dpitch + n] = src[blockIdx.z * simg + (size_t)(2 * m) * spitch + 2 * n];
}

// Sub (reference sift/sift.cpp:416-437): c = a - b.
template <typename T>
__global__ __launch_bounds__(256) void sub_k(const T *a, int apitch, size_t aimg, const T *b, int bpitch, size_t bimg,
                                             T *c, int rows, int cols, int cpitch, size_t cimg)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows)
        return;
    c[blockIdx.z * cimg + (size_t)y * cpitch + x] =
        a[blockIdx.z * aimg + (size_t)y * apitch + x] - b[blockIdx.z * bimg + (size_t)y * bpitch + x];
}

static inline dim3 grid2d(int cols, int rows, int B) { return dim3((cols + 63) / 64, (rows + 3) / 4, B); }

void launch_gray(const uint8_t *src, int rows, int cols, int ch, size_t stride, size_t img_stride,
                 const Plane &dst, int B, int prec, hipStream_t s)
{
    dim3 g((cols + 255) / 256, (rows + 3) / 4, B);
    // dword loads of the channel bytes: three interleaved channels and rows that start on 4-byte boundaries
    const bool quads = ch == 3 && (uintptr_t)src % 4 == 0 && stride % 4 == 0 && img_stride % 4 == 0;
    const bool vec_ok = dst.pitch % 4 == 0 && (uintptr_t)dst.p % 16 == 0 && dst.img_stride % 4 == 0;
#define SG_GRAY(TT, QQ)                                                                                               \
    gray_k<TT, QQ><<<g, 256, 0, s>>>(src, rows, cols, ch, stride, img_stride, (TT *)dst.p, dst.pitch,                \
                                     dst.img_stride, vec_ok ? 1 : 0)
    if (prec == SG_F64) {
        if (quads) SG_GRAY(double, true); else SG_GRAY(double, false);
    } else {
        if (quads) SG_GRAY(float, true); else SG_GRAY(float, false);
    }
#undef SG_GRAY
}

void launch_upsample(const Plane &src, const Plane &dst, int B, int prec, hipStream_t s)
{
    if (src.rows < 2 || src.cols < 2)
        return;
    dim3 g = grid2d(src.cols - 1, src.rows - 1, B);
    if (prec == SG_F64)
        upsample_k<double><<<g, 256, 0, s>>>((const double *)src.p, src.rows, src.cols, src.pitch, src.img_stride,
                                             (double *)dst.p, dst.pitch, dst.img_stride);
    else
        upsample_k<float><<<g, 256, 0, s>>>((const float *)src.p, src.rows, src.cols, src.pitch, src.img_stride,
                                            (float *)dst.p, dst.pitch, dst.img_stride);
}

void launch_gray_upsample(const uint8_t *src, int rows, int cols, int ch, size_t stride, size_t img_stride,
                          const Plane &dst, int B, int prec, hipStream_t s)
{
    dim3 g = grid2d(cols - 1, rows - 1, B);
    if (prec == SG_F64)
        gray_upsample_k<double><<<g, 256, 0, s>>>(src, rows, cols, ch, stride, img_stride, (double *)dst.p, dst.pitch,
                                                  dst.img_stride);
    else
        gray_upsample_k<float><<<g, 256, 0, s>>>(src, rows, cols, ch, stride, img_stride, (float *)dst.p, dst.pitch,
                                                 dst.img_stride);
}

void launch_downsample(const Plane &src, const Plane &dst, int B, int prec, hipStream_t s)
{
    dim3 g = grid2d(dst.cols, dst.rows, B);
    if (prec == SG_F64)
        downsample_k<double><<<g, 256, 0, s>>>((const double *)src.p, src.pitch, src.img_stride, (double *)dst.p,
                                               dst.rows, dst.cols, dst.pitch, dst.img_stride);
    else
        downsample_k<float><<<g, 256, 0, s>>>((const float *)src.p, src.pitch, src.img_stride, (float *)dst.p, dst.rows,
                                              dst.cols, dst.pitch, dst.img_stride);
}

void launch_sub(const Plane &a, const Plane &b, const Plane &c, int B, int prec, hipStream_t s)
{
    dim3 g = grid2d(c.cols, c.rows, B);
    if (prec == SG_F64)
        sub_k<double><<<g, 256, 0, s>>>((const double *)a.p, a.pitch, a.img_stride, (const double *)b.p, b.pitch,
                                        b.img_stride, (double *)c.p, c.rows, c.cols, c.pitch, c.img_stride);
    else
        sub_k<float><<<g, 256, 0, s>>>((const float *)a.p, a.pitch, a.img_stride, (const float *)b.p, b.pitch,
                                       b.img_stride, (float *)c.p, c.rows, c.cols, c.pitch, c.img_stride);
}

// ---- synthetic input: 5-octave value noise from a stateless 64-bit hash --------------------------

__device__ __forceinline__ uint64_t sm64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ double lat(int ix, int iy, int k, uint64_t seed)
{
    uint64_t c = (uint64_t)(uint32_t)ix | ((uint64_t)(uint32_t)iy << 32);
    uint64_t h = sm64(seed ^ sm64(c) ^ sm64(0x1000 + (uint64_t)k));
    return ((double)(h >> 11) * 0x1.0p-53) * 2 - 1;
}

__device__ __forceinline__ double vnoise(double x, double y, int k, uint64_t seed)
{
    double fx0 = floor(x), fy0 = floor(y);
    int ix = (int)fx0, iy = (int)fy0;
    double fx = x - fx0, fy = y - fy0;
    fx = fx * fx * (3 - 2 * fx);
    fy = fy * fy * (3 - 2 * fy);
    double a = lat(ix, iy, k, seed), b = lat(ix + 1, iy, k, seed);
    double c = lat(ix, iy + 1, k, seed), d = lat(ix + 1, iy + 1, k, seed);
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy;
}

__global__ __launch_bounds__(256) void vnoise5_k(uint8_t *dst, int rows, int cols, uint64_t seed)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows)
        return;
    const double L[5] = {4, 8, 16, 32, 64};
    const double A[5] = {20, 28, 36, 40, 40};
    double v = 128;
#pragma unroll
    for (int k = 0; k < 5; k++)
        v += A[k] * vnoise(x / L[k], y / L[k], k, seed);
    double q = rint(v);
    q = q < 0 ? 0 : (q > 255 ? 255 : q);
    uint8_t u = (uint8_t)q;
    uint8_t *p = dst + ((size_t)y * cols + x) * 3;
    p[0] = u; p[1] = u; p[2] = u;
}

void launch_vnoise5(uint8_t *dst, int rows, int cols, uint64_t seed, hipStream_t s)
{
    vnoise5_k<<<grid2d(cols, rows, 1), 256, 0, s>>>(dst, rows, cols, seed);
}

} // namespace sg
