// Access-pattern microbenchmark 2: copy B planes of 3840x2160 doubles with variations of the blur kernels' walk.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// block = NW waves = NW adjacent strips of 64*V doubles, same rows; PF rows prefetched ahead; XCD: remap block ids so
// that each XCD gets a contiguous run of (image, segment) tiles
template <int NW, int V, int PF, bool NT>
__global__ __launch_bounds__(64 * NW) void strip_k(const double *__restrict__ src, double *__restrict__ dst, int rows, int cols, int pitch, size_t img, int seg, int nseg, int nsx, int remap)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int id = blockIdx.x;
    const int total = gridDim.x;
    if (remap) { const int per = total / 8; id = (id % 8) * per + id / 8; if (id >= total) return; }
    const int bx = id % nsx; id /= nsx;
    const int sidx = id % nseg; const int z = id / nseg;
    const int strip = bx * NW + wave;
    const int x = strip * 64 * V + lane * V;
    const int ys = sidx * seg;
    if (ys >= rows || x >= cols) return;
    const int ye = min(ys + seg, rows);
    const double *s = src + z * img + x;
    double *d = dst + z * img + x;
    double2 buf[PF][V / 2];
#pragma unroll
    for (int k = 0; k < PF; k++)
#pragma unroll
        for (int q = 0; q < V / 2; q++) buf[k][q] = *(const double2 *)(s + (size_t)min(ys + k, ye - 1) * pitch + 2 * q);
    for (int y = ys; y < ye; y += PF) {
        double2 cur[PF][V / 2];
#pragma unroll
        for (int k = 0; k < PF; k++)
#pragma unroll
            for (int q = 0; q < V / 2; q++) cur[k][q] = buf[k][q];
#pragma unroll
        for (int k = 0; k < PF; k++)
#pragma unroll
            for (int q = 0; q < V / 2; q++) buf[k][q] = *(const double2 *)(s + (size_t)min(y + PF + k, ye - 1) * pitch + 2 * q);
#pragma unroll
        for (int k = 0; k < PF; k++) if (y + k < ye) {
#pragma unroll
            for (int q = 0; q < V / 2; q++) { double2 v = cur[k][q]; v.x = v.x * 1.0000001 + 1.0;
                if (NT) { typedef double d2v __attribute__((ext_vector_type(2))); d2v w; w.x = v.x; w.y = v.y; __builtin_nontemporal_store(w, (d2v *)(d + (size_t)(y + k) * pitch + 2 * q)); }
                else *(double2 *)(d + (size_t)(y + k) * pitch + 2 * q) = v; }
        }
    }
}

// The same walk with other read : write mixes, as the pyramid's launches have them: RDIV = 0 writes only, RDIV = n reads
// one input row per n output rows (the base blur reads a quarter-size gray plane: 1 byte read per 4 written), NOUT = 2
// writes two planes per plane read (levels 1 and 2 of an octave from one launch).  Four waves per block, 128 columns
// per wave, segments of `seg` rows.  The reported rate counts the bytes actually moved.
template <int RDIV, int NOUT>
__global__ __launch_bounds__(256) void strip_mix_k(const double *__restrict__ src, double *__restrict__ dst, double *__restrict__ dst2, int rows, int cols, int pitch, size_t img, int seg, int nseg, int nsx)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int id = blockIdx.x;
    const int bx = id % nsx; id /= nsx;
    const int sidx = id % nseg; const int z = id / nseg;
    const int x = (bx * 4 + wave) * 128 + lane * 2;
    const int ys = sidx * seg;
    if (ys >= rows || x >= cols) return;
    const int ye = min(ys + seg, rows);
    const double *s = src + z * img + x;
    double *d = dst + z * img + x, *d2 = dst2 + z * img + x;
    double2 v = {1.0, 2.0};
    for (int y = ys; y < ye; y++) {
        if (RDIV > 0 && (y - ys) % RDIV == 0) {
            const double2 r = *(const double2 *)(s + (size_t)(RDIV > 1 ? ys + (y - ys) / RDIV : y) * pitch);
            v.x += r.x; v.y += r.y;
        }
        v.x = v.x * 1.0000001 + 1.0;
        *(double2 *)(d + (size_t)y * pitch) = v;
        if (NOUT == 2) { v.y += 1.0; *(double2 *)(d2 + (size_t)y * pitch) = v; }
    }
}

__global__ __launch_bounds__(256) void linear_k(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { double2 v = src[i]; v.x += 1.0; dst[i] = v; }
}
__global__ __launch_bounds__(256) void read_k(const double2 *__restrict__ src, double *__restrict__ out, size_t n)
{
    double acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { double2 v = src[i]; acc += v.x + v.y; }
    if (acc == 1.2345) out[0] = acc;
}
__global__ __launch_bounds__(256) void write_k(double2 *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { double2 v; v.x = (double)i; v.y = 1.0; dst[i] = v; }
}

int main(int argc, char **argv)
{
    const int rows = 2160, cols = 3840, B = argc > 1 ? atoi(argv[1]) : 64;
    const size_t img = (size_t)rows * cols, n = img * B;
    double *a, *b;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
    CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, double bytes, auto fn) {
        fn(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int i = 0; i < 5; i++) fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9); fflush(stdout);
    };
    timeit("linear copy", 2.0 * n * 8, [&] { linear_k<<<256 * 16, 256>>>((const double2 *)a, (double2 *)b, n / 2); });
    timeit("linear copy x32 blocks", 2.0 * n * 8, [&] { linear_k<<<256 * 32, 256>>>((const double2 *)a, (double2 *)b, n / 2); });
    timeit("read only", 1.0 * n * 8, [&] { read_k<<<256 * 16, 256>>>((const double2 *)a, b, n / 2); });
    timeit("write only", 1.0 * n * 8, [&] { write_k<<<256 * 16, 256>>>((double2 *)b, n / 2); });
    {
        double *c2;
        CK(hipMalloc(&c2, n * 8));
        CK(hipMemset(c2, 0, n * 8));
        const int mix_segs[3] = {2160, 540, 136};
        for (int seg : mix_segs) {
            const int nseg = (rows + seg - 1) / seg, nsx = (cols / 128 + 3) / 4, tot = nsx * nseg * B;
            char nm[96];
            snprintf(nm, 96, "strip walk, write only, seg %d", seg);
            timeit(nm, 1.0 * n * 8, [&] { strip_mix_k<0, 1><<<tot, 256>>>(a, b, c2, rows, cols, cols, img, seg, nseg, nsx); });
            snprintf(nm, 96, "strip walk, 1 read : 4 written, seg %d", seg);
            timeit(nm, 1.25 * n * 8, [&] { strip_mix_k<4, 1><<<tot, 256>>>(a, b, c2, rows, cols, cols, img, seg, nseg, nsx); });
            snprintf(nm, 96, "strip walk, 1 read : 1 written, seg %d", seg);
            timeit(nm, 2.0 * n * 8, [&] { strip_mix_k<1, 1><<<tot, 256>>>(a, b, c2, rows, cols, cols, img, seg, nseg, nsx); });
            snprintf(nm, 96, "strip walk, 1 read : 2 written, seg %d", seg);
            timeit(nm, 3.0 * n * 8, [&] { strip_mix_k<1, 2><<<tot, 256>>>(a, b, c2, rows, cols, cols, img, seg, nseg, nsx); });
        }
        CK(hipFree(c2));
    }
    if (argc > 2 && atoi(argv[2]) == 0)
        return 0; // the mixes only
    int segs[16] = {540, 270, 136, 68}, nsegs = 4;
    if (argc > 2) { nsegs = 0; for (int i = 2; i < argc && nsegs < 16; i++) segs[nsegs++] = atoi(argv[i]); }
    for (int si = 0; si < nsegs; si++) {
        const int seg = segs[si];
        int nseg = (rows + seg - 1) / seg;
        char nm[80];
#define RUN(NW, V, PF, NT, REMAP) { const int nsx = (cols / (64 * V) + NW - 1) / NW; int tot = nsx * nseg * B; if (REMAP) tot = (tot + 7) / 8 * 8; \
        snprintf(nm, 80, "NW%d V%d PF%d nt%d remap%d seg %d", NW, V, PF, NT, REMAP, seg); \
        timeit(nm, 2.0 * n * 8, [&] { strip_k<NW, V, PF, NT><<<tot, 64 * NW>>>(a, b, rows, cols, cols, img, seg, nseg, nsx, REMAP); }); }
        RUN(4, 2, 4, false, 0)
        RUN(4, 2, 4, false, 1)
        RUN(4, 2, 4, true, 0)
        RUN(8, 2, 4, false, 0)
        RUN(4, 4, 2, false, 0)
        RUN(4, 4, 4, false, 0)
        RUN(2, 4, 4, false, 0)
        RUN(1, 2, 4, false, 0)
        RUN(16, 2, 4, false, 0)
        RUN(16, 2, 4, false, 1)
        RUN(4, 2, 1, false, 0)
        RUN(4, 2, 2, false, 0)
        RUN(4, 2, 2, false, 1)
    }
    return 0;
}
