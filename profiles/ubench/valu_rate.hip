// Issue cost of the vector and LDS instructions the histogram kernels are made of, on gfx950 (a measuring tool, not part
// of the library).  Every kernel runs 8 independent chains of one instruction per lane, 5 waves per SIMD on all 256 CUs;
// the result is ns of a SIMD per wave-instruction (LDS: ns of a CU's LDS pipe per wave-instruction, 20 waves per CU).
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

enum { ADD64, MUL64, FMA64, RSQ64, RCP64, FRACT64, FLOOR64, CVT_I2D, CVT_D2I, MAX64, CMP64, FMA32, CNDMASK, MOV32, MOV64,
       MULLO, LSHLADD, DSADD64_SPREAD, DSADD64_SAME16, DSREAD64, N_MODES };
static const char *names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_rsq_f64", "v_rcp_f64", "v_fract_f64", "v_floor_f64",
                              "v_cvt_f64_i32", "v_cvt_i32_f64", "v_max_f64", "v_cmp_gt_f64", "v_fma_f32", "v_cndmask_b32",
                              "v_mov_b32", "v_mov_b64", "v_mul_lo_u32", "v_lshl_add_u32",
                              "ds_add_f64 (64 lanes, 64 addresses, 4 copies by lane&3)",
                              "ds_add_f64 (16 lanes per address)", "ds_read_b64"};

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, int iters, double a, double b)
{
    __shared__ double lds[4 * 130 + 64];
    const int lane = threadIdx.x;
    double x[8];
    float f[8];
    int n[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        x[q] = 1.0 + 0.01 * (lane + q);
        f[q] = (float)x[q];
        n[q] = lane + q;
    }
    for (int q = lane; q < 4 * 130 + 64; q += 64)
        lds[q] = 0.0;
    __syncthreads();
    const unsigned long long mask = __ballot(lane & 1);
    const unsigned int a_spread = (unsigned int)(uintptr_t)(lds + (lane & 3) * 130 + (lane >> 2) * 5 % 128);
    const unsigned int a_same = (unsigned int)(uintptr_t)(lds + (lane & 3) * 130 + 17);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#define S_ADD(q) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[q]) : "v"(a));
#define S_MUL(q) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[q]) : "v"(a));
#define S_FMA(q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[q]) : "v"(a), "v"(b));
#define S_RSQ(q) asm volatile("v_rsq_f64 %0, %0" : "+v"(x[q]));
#define S_RCP(q) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[q]));
#define S_FRACT(q) asm volatile("v_fract_f64 %0, %0" : "+v"(x[q]));
#define S_FLOOR(q) asm volatile("v_floor_f64 %0, %0" : "+v"(x[q]));
#define S_I2D(q) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x[q]) : "v"(n[q]));
#define S_D2I(q) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[q]) : "v"(x[q]));
#define S_MAX(q) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[q]) : "v"(a));
#define S_CMP(q) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(x[q]), "v"(a) : "vcc");
#define S_F32(q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[q]) : "v"((float)a), "v"((float)b));
#define S_CND(q) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(n[q]) : "v"(lane), "s"(mask));
#define S_MOV(q) asm volatile("v_mov_b32 %0, %1" : "=v"(n[q]) : "v"(n[(q + 1) & 7]));
#define S_MOV64(q) asm volatile("v_mov_b64 %0, %1" : "=v"(x[q]) : "v"(x[(q + 1) & 7]));
#define S_MULLO(q) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[q]) : "v"(lane | 1));
#define S_LSHLADD(q) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(n[q]) : "v"(lane));
#define S_DSA(q) asm volatile("ds_add_f64 %0, %1 offset:%2" : : "v"(a_spread), "v"(x[q]), "n"(8 * q) : "memory");
#define S_DSS(q) asm volatile("ds_add_f64 %0, %1 offset:%2" : : "v"(a_same), "v"(x[q]), "n"(8 * q) : "memory");
#define S_DSR(q) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[q]) : "v"(a_spread), "n"(8 * q) : "memory");
            if (MODE == ADD64) { REP8(S_ADD) }
            if (MODE == MUL64) { REP8(S_MUL) }
            if (MODE == FMA64) { REP8(S_FMA) }
            if (MODE == RSQ64) { REP8(S_RSQ) }
            if (MODE == RCP64) { REP8(S_RCP) }
            if (MODE == FRACT64) { REP8(S_FRACT) }
            if (MODE == FLOOR64) { REP8(S_FLOOR) }
            if (MODE == CVT_I2D) { REP8(S_I2D) }
            if (MODE == CVT_D2I) { REP8(S_D2I) }
            if (MODE == MAX64) { REP8(S_MAX) }
            if (MODE == CMP64) { REP8(S_CMP) }
            if (MODE == FMA32) { REP8(S_F32) }
            if (MODE == CNDMASK) { REP8(S_CND) }
            if (MODE == MOV32) { REP8(S_MOV) }
            if (MODE == MOV64) { REP8(S_MOV64) }
            if (MODE == MULLO) { REP8(S_MULLO) }
            if (MODE == LSHLADD) { REP8(S_LSHLADD) }
            if (MODE == DSADD64_SPREAD) { REP8(S_DSA) }
            if (MODE == DSADD64_SAME16) { REP8(S_DSS) }
            if (MODE == DSREAD64) { REP8(S_DSR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        }
        if (MODE == DSADD64_SPREAD || MODE == DSADD64_SAME16)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    double s = 0;
#pragma unroll
    for (int q = 0; q < 8; q++)
        s += x[q] + f[q] + n[q];
    out[blockIdx.x * 64 + lane] = s + lds[lane];
}

template <int MODE>
void run(double *d)
{
    const int iters = 2000, waves_per_simd = 5;
    const int blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(d, 10, 1.0000001, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(d, iters, 1.0000001, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 32 * waves_per_simd; // wave-instructions per SIMD
    const bool lds = MODE >= DSADD64_SPREAD;
    printf("%-58s %7.2f ns per wave-instruction per %s\n", names[MODE], ms * 1e6 / (per_simd * (lds ? 4 : 1)), lds ? "CU (LDS pipe)" : "SIMD");
    fflush(stdout);
}

template <int M>
struct All {
    static void go(double *d)
    {
        run<M>(d);
        All<M + 1>::go(d);
    }
};
template <>
struct All<N_MODES> {
    static void go(double *) {}
};

int main()
{
    double *d;
    hipMalloc(&d, 256 * 4 * 5 * 64 * 8);
    All<0>::go(d);
    return 0;
}
