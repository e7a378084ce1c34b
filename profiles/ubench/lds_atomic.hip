#include <hip/hip_runtime.h>
#include <stdio.h>
// LDS atomic cost vs active lanes / width / conflicts; 20 waves per CU (5 per SIMD), ns of the CU's LDS pipe per wave-instruction
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, int iters, int active, int spread)
{
    __shared__ double lds[4 * 130 + 64];
    const int lane = threadIdx.x;
    for (int q = lane; q < 4 * 130 + 64; q += 64) lds[q] = 0.0;
    __syncthreads();
    double x = 1.0 + lane; float f = x; unsigned int u = lane; unsigned long long w = lane;
    // spread: 0 = all active lanes one address per copy (lane&3), 1 = distinct addresses
    const unsigned int a = (unsigned int)(uintptr_t)(lds + (lane & 3) * 130 + (spread ? (lane >> 2) * 5 % 120 : 17));
    if (lane < active) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (MODE == 0) asm volatile("ds_add_f64 %0, %1 offset:%2" : : "v"(a), "v"(x), "n"(8 * q) : "memory");
                if (MODE == 1) asm volatile("ds_add_f32 %0, %1 offset:%2" : : "v"(a), "v"(f), "n"(8 * q) : "memory");
                if (MODE == 2) asm volatile("ds_add_u32 %0, %1 offset:%2" : : "v"(a), "v"(u), "n"(8 * q) : "memory");
                if (MODE == 3) asm volatile("ds_add_u64 %0, %1 offset:%2" : : "v"(a), "v"(w), "n"(8 * q) : "memory");
                if (MODE == 4) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(a), "v"(x), "n"(8 * q) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    out[blockIdx.x * 64 + lane] = lds[lane];
}
template <int MODE> void run(const char *name, double *d, int active, int spread)
{
    const int iters = 4000, blocks = 256 * 20;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(d, 10, active, spread); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 64>>>(d, iters, active, spread); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s active %2d %s: %6.2f ns per wave-instruction per CU\n", name, active, spread ? "distinct addr" : "same addr/copy", ms * 1e6 / ((double)iters * 8 * 20));
}
int main()
{
    double *d; hipMalloc(&d, 256 * 20 * 64 * 8);
    for (int spread = 1; spread >= 0; spread--)
        for (int act : {64, 32, 16, 8}) {
            run<0>("ds_add_f64", d, act, spread); run<1>("ds_add_f32", d, act, spread); run<2>("ds_add_u32", d, act, spread);
            run<3>("ds_add_u64", d, act, spread); run<4>("ds_write_b64", d, act, spread);
        }
    return 0;
}
