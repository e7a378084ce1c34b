// Access-pattern microbenchmark 3: what a one-read-one-write stream reaches on this part when the copy kernel is
// written the way a bandwidth test is (several independent 16-byte loads in flight per lane, grid sized from the
// resident-wave slots, optional non-temporal loads / stores, optional XCD-contiguous chunking), on buffers of
// 2 GB and more.  Prints bytes read + bytes written per second.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ f4 ld(const f4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void st(f4 *p, f4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// grid-stride copy, U independent loads in flight per lane; consecutive loads of a lane are one grid apart
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_gs_k(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = blockIdx.x * 256ull + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ld<NTL>(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; u++) st<NTS>(dst + i + u * stride, v[u]);
    }
    for (; i < n; i += stride) st<NTS>(dst + i, ld<NTL>(src + i));
}

// every workgroup owns one contiguous chunk of U*256 vectors per trip; REMAP gives each XCD (blocks b, b+8, ...) a
// contiguous eighth of the buffer
template <int U, bool NTL, bool NTS, bool REMAP>
__global__ __launch_bounds__(256) void copy_chunk_k(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    const size_t chunk = (size_t)U * 256;
    const size_t nchunks = n / chunk;
    size_t nb = gridDim.x, b = blockIdx.x;
    if (REMAP) {
        // block b of XCD x = b % 8 walks chunks [x * nchunks/8, (x+1) * nchunks/8) with the other blocks of its XCD
        const size_t per = nchunks / 8, x = b % 8, j = b / 8, nj = nb / 8;
        for (size_t c = j; c < per; c += nj) {
            const size_t i = (x * per + c) * chunk + threadIdx.x;
            f4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = ld<NTL>(src + i + u * 256);
#pragma unroll
            for (int u = 0; u < U; u++) st<NTS>(dst + i + u * 256, v[u]);
        }
        return;
    }
    for (size_t c = b; c < nchunks; c += nb) {
        const size_t i = c * chunk + threadIdx.x;
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ld<NTL>(src + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; u++) st<NTS>(dst + i + u * 256, v[u]);
    }
}

// one trip per thread: the grid covers the buffer (what hipMemcpy-style kernels do)
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_once_k(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * (U * 256) + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (i + u * 256 < n) v[u] = ld<NTL>(src + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; u++) if (i + u * 256 < n) st<NTS>(dst + i + u * 256, v[u]);
}

template <int U, bool NTL>
__global__ __launch_bounds__(256) void read_k(const f4 *__restrict__ src, float *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ld<NTL>(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) out[0] = acc.x;
}

template <int U, bool NTS>
__global__ __launch_bounds__(256) void write_k(f4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        f4 v = {(float)i, 1.f, 2.f, 3.f};
#pragma unroll
        for (int u = 0; u < U; u++) st<NTS>(dst + i + u * stride, v);
    }
}

int main(int argc, char **argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 2.0; // per buffer
    const size_t n = (size_t)(gb * (1ull << 30)) / 16 / (8 * 2048) * (8 * 2048);
    f4 *a, *b;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 0, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("buffers: 2 x %.2f GB\n", n * 16 / 1e9);
    auto timeit = [&](const char *name, double bytes, auto fn) {
        fn(); fn(); CK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0;
        for (int r = 0; r < 5; r++) {
            CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms; if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("%-58s avg %.3f ms %.2f TB/s   best %.2f TB/s\n", name, sum / 5, bytes / (sum / 5) / 1e9, bytes / best / 1e9); fflush(stdout);
    };
    const double rw = 2.0 * n * 16, one = 1.0 * n * 16;
    char nm[96];
    timeit("hipMemcpyAsync device to device", rw, [&] { CK(hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0)); });
    const int grids[] = {1024, 2048, 4096, 8192, 16384};
    for (int g : grids) {
#define GS(U, NTL, NTS) snprintf(nm, 96, "grid-stride U%d ntl%d nts%d grid %d", U, NTL, NTS, g); \
        timeit(nm, rw, [&] { copy_gs_k<U, NTL, NTS><<<g, 256>>>(a, b, n); });
        GS(1, false, false) GS(2, false, false) GS(4, false, false) GS(8, false, false)
        GS(4, true, false) GS(4, false, true) GS(4, true, true) GS(8, true, true)
#define CH(U, NTL, NTS, RM) snprintf(nm, 96, "chunk U%d ntl%d nts%d xcd-remap%d grid %d", U, NTL, NTS, RM, g); \
        timeit(nm, rw, [&] { copy_chunk_k<U, NTL, NTS, RM><<<g, 256>>>(a, b, n); });
        CH(4, false, false, false) CH(8, false, false, false) CH(4, false, false, true) CH(8, false, false, true)
        CH(4, true, true, true) CH(8, true, true, true) CH(8, false, true, true)
    }
#define ON(U, NTL, NTS) snprintf(nm, 96, "one trip per thread U%d ntl%d nts%d", U, NTL, NTS); \
    timeit(nm, rw, [&] { copy_once_k<U, NTL, NTS><<<(unsigned)((n + U * 256 - 1) / (U * 256)), 256>>>(a, b, n); });
    ON(1, false, false) ON(2, false, false) ON(4, false, false) ON(8, false, false) ON(4, true, true) ON(4, false, true)
    for (int g : {2048, 8192}) {
        snprintf(nm, 96, "read only U4 grid %d", g); timeit(nm, one, [&] { read_k<4, false><<<g, 256>>>(a, (float *)b, n); });
        snprintf(nm, 96, "read only U8 nt grid %d", g); timeit(nm, one, [&] { read_k<8, true><<<g, 256>>>(a, (float *)b, n); });
        snprintf(nm, 96, "write only U4 grid %d", g); timeit(nm, one, [&] { write_k<4, false><<<g, 256>>>(b, n); });
        snprintf(nm, 96, "write only U4 nt grid %d", g); timeit(nm, one, [&] { write_k<4, true><<<g, 256>>>(b, n); });
    }
    return 0;
}
