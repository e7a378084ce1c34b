// How instruction classes share a SIMD's issue on gfx950 (a measuring tool, not part of the library): a block of 32
// independent v_fma_f64 per loop pass, alone and interleaved one-to-one with 32 instructions of another class, at 1, 2
// and 5 waves per SIMD on all 256 CUs.  Output: ns of a SIMD per loop pass of one wave (wall time / passes / waves per
// SIMD).  If a class issues beside the vector instructions of other waves, its row costs what the plain row costs.
//   hipcc -O3 --offload-arch=gfx950 -o issue_mix issue_mix.hip && ./issue_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

enum { FMA_ONLY, FMA_SALU, FMA_NOP, FMA_VMOV, FMA_BRANCH, FMA_EXEC, SALU_ONLY, FMA_DEP, FMA_DEP_SALU, VMOV_ONLY, FMA_LDSREAD,
       FMA_WAITCNT, FMA_VCMP, FMA_EXECONLY, FMA_EXECMOV, FMA_CMP_CND_SGPR, FMA_CMP_CND_VCC, FMA_READLANE, FMA_CMP_SAND, FMA_DSADD64, FMA_DSWRITE16, N_MODES };
static const char *names[] = {"32 v_fma_f64 (8 chains)",
                              "  + 32 s_add_u32",
                              "  + 32 s_nop 0",
                              "  + 32 v_mov_b32",
                              "  + 32 s_cbranch_scc1 (not taken) with s_cmp",
                              "  + 16 (v_cmp, s_and_saveexec, s_or exec)",
                              "32 s_add_u32 alone",
                              "32 v_fma_f64, one dependent chain",
                              "  (dependent chain) + 32 s_add_u32",
                              "32 v_mov_b32 alone",
                              "  + 32 ds_read_b64 (no wait inside the block)",
                              "  + 32 s_waitcnt lgkmcnt(0) (nothing outstanding)",
                              "  + 32 v_cmp_gt_i32 vcc",
                              "  + 16 (s_and_saveexec, s_or exec), mask in SGPRs",
                              "  + 32 s_mov_b64 exec, s[20:21]",
                              "  + 16 (v_cmp_gt_i32_e64 s[22:23]; v_cndmask_b32_e64 .. s[22:23])",
                              "  + 16 (v_cmp_gt_i32 vcc; v_cndmask_b32_e32 .. vcc)",
                              "  + 32 v_readfirstlane_b32",
                              "  + 16 (v_cmp_gt_i32 vcc; s_and_b64 s[22:23], vcc, s[20:21])",
                              "  + 32 ds_add_u64 (64 lanes, 64 addresses)",
                              "  + 32 ds_write_b16 (64 neighbouring entries)"};

#define REP4(S) S S S S
#define REP8(S) REP4(S) REP4(S)
#define REP32(S) REP8(S) REP8(S) REP8(S) REP8(S)

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, int iters, double a, double b)
{
    __shared__ double lds[512];
    const int lane = threadIdx.x;
    double x0 = 1.0 + lane, x1 = 2.0 + lane, x2 = 3.0 + lane, x3 = 4.0 + lane, x4 = 5.0 + lane, x5 = 6.0 + lane,
           x6 = 7.0 + lane, x7 = 8.0 + lane;
    int n0 = lane, n1 = lane + 1, n2 = 0;
    double r0 = 0, r1 = 0;
    for (int q = lane; q < 512; q += 64)
        lds[q] = q;
    __syncthreads();
    const unsigned int la = (unsigned int)(size_t)(__attribute__((address_space(3))) double *)(lds + lane);
    const unsigned int la2 = (unsigned int)(size_t)(__attribute__((address_space(3))) double *)lds + 2u * lane;
    for (int i = 0; i < iters; i++) {
#define FMA8                                                                                                            \
    "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"           \
    "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
#define FMA8_WITH(X)                                                                                                    \
    "v_fma_f64 %0, %0, %8, %9\n" X "v_fma_f64 %1, %1, %8, %9\n" X "v_fma_f64 %2, %2, %8, %9\n" X                           \
    "v_fma_f64 %3, %3, %8, %9\n" X "v_fma_f64 %4, %4, %8, %9\n" X "v_fma_f64 %5, %5, %8, %9\n" X                           \
    "v_fma_f64 %6, %6, %8, %9\n" X "v_fma_f64 %7, %7, %8, %9\n" X
#define DEP8_WITH(X)                                                                                                    \
    "v_fma_f64 %0, %0, %8, %9\n" X "v_fma_f64 %0, %0, %8, %9\n" X "v_fma_f64 %0, %0, %8, %9\n" X                           \
    "v_fma_f64 %0, %0, %8, %9\n" X "v_fma_f64 %0, %0, %8, %9\n" X "v_fma_f64 %0, %0, %8, %9\n" X                           \
    "v_fma_f64 %0, %0, %8, %9\n" X "v_fma_f64 %0, %0, %8, %9\n" X
#define OPS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)
        if (MODE == FMA_ONLY) {
            asm volatile(REP4(FMA8) OPS);
        } else if (MODE == FMA_SALU) {
            asm volatile(REP4(FMA8_WITH("s_add_u32 s20, s20, 1\n")) OPS : "s20", "scc");
        } else if (MODE == FMA_NOP) {
            asm volatile(REP4(FMA8_WITH("s_nop 0\n")) OPS);
        } else if (MODE == FMA_VMOV) {
            asm volatile(REP4(FMA8_WITH("v_mov_b32 %10, %11\n")) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5),
                         "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(n0), "v"(n1));
        } else if (MODE == FMA_BRANCH) {
            asm volatile("s_cmp_eq_u32 s20, s20\n" REP4(FMA8_WITH("s_cbranch_scc0 1f\n")) "1:\n" OPS : "s20", "scc");
        } else if (MODE == FMA_EXEC) {
            asm volatile(REP4(FMA8_WITH("v_cmp_gt_i32 vcc, %10, %11\n s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                         : "v"(a), "v"(b), "v"(n1), "v"(n0) : "s20", "s21", "vcc", "scc");
        } else if (MODE == SALU_ONLY) {
            asm volatile(REP32("s_add_u32 s20, s20, 1\n") ::: "s20", "scc");
        } else if (MODE == FMA_DEP) {
            asm volatile(REP4(DEP8_WITH("")) OPS);
        } else if (MODE == FMA_DEP_SALU) {
            asm volatile(REP4(DEP8_WITH("s_add_u32 s20, s20, 1\n")) OPS : "s20", "scc");
        } else if (MODE == VMOV_ONLY) {
            asm volatile(REP32("v_mov_b32 %0, %1\n") : "+v"(n0) : "v"(n1));
        } else if (MODE == FMA_LDSREAD) {
            asm volatile(REP4(FMA8_WITH("ds_read_b64 %10, %11\n")) "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                         : "v"(a), "v"(b), "v"(r0), "v"(la) : "memory");
        } else if (MODE == FMA_WAITCNT) {
            asm volatile(REP4(FMA8_WITH("s_waitcnt lgkmcnt(0)\n")) OPS);
        } else if (MODE == FMA_VCMP) {
            asm volatile(REP4(FMA8_WITH("v_cmp_gt_i32 vcc, %10, %11\n"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                         : "v"(a), "v"(b), "v"(n1), "v"(n0) : "vcc");
        } else if (MODE == FMA_EXECONLY) {
            asm volatile("s_mov_b64 s[22:23], exec\n"
                         REP4("v_fma_f64 %0, %0, %8, %9\n s_and_saveexec_b64 s[20:21], s[22:23]\n v_fma_f64 %1, %1, %8, %9\n s_or_b64 exec, exec, s[20:21]\n"
                              "v_fma_f64 %2, %2, %8, %9\n s_and_saveexec_b64 s[20:21], s[22:23]\n v_fma_f64 %3, %3, %8, %9\n s_or_b64 exec, exec, s[20:21]\n"
                              "v_fma_f64 %4, %4, %8, %9\n s_and_saveexec_b64 s[20:21], s[22:23]\n v_fma_f64 %5, %5, %8, %9\n s_or_b64 exec, exec, s[20:21]\n"
                              "v_fma_f64 %6, %6, %8, %9\n s_and_saveexec_b64 s[20:21], s[22:23]\n v_fma_f64 %7, %7, %8, %9\n s_or_b64 exec, exec, s[20:21]\n")
                         OPS : "s20", "s21", "s22", "s23", "scc");
        } else if (MODE == FMA_EXECMOV) {
            asm volatile("s_mov_b64 s[20:21], exec\n" REP4(FMA8_WITH("s_mov_b64 exec, s[20:21]\n")) OPS : "s20", "s21");
        } else if (MODE == FMA_CMP_CND_SGPR) {
            asm volatile(REP4("v_fma_f64 %0, %0, %8, %9\n v_cmp_gt_i32_e64 s[22:23], %10, %11\n v_fma_f64 %1, %1, %8, %9\n v_cndmask_b32_e64 v40, %10, %11, s[22:23]\n"
                              "v_fma_f64 %2, %2, %8, %9\n v_cmp_gt_i32_e64 s[22:23], %10, %11\n v_fma_f64 %3, %3, %8, %9\n v_cndmask_b32_e64 v40, %10, %11, s[22:23]\n"
                              "v_fma_f64 %4, %4, %8, %9\n v_cmp_gt_i32_e64 s[22:23], %10, %11\n v_fma_f64 %5, %5, %8, %9\n v_cndmask_b32_e64 v40, %10, %11, s[22:23]\n"
                              "v_fma_f64 %6, %6, %8, %9\n v_cmp_gt_i32_e64 s[22:23], %10, %11\n v_fma_f64 %7, %7, %8, %9\n v_cndmask_b32_e64 v40, %10, %11, s[22:23]\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                         : "v"(a), "v"(b), "v"(n1), "v"(n0) : "s22", "s23", "v40");
        } else if (MODE == FMA_CMP_CND_VCC) {
            asm volatile(REP4("v_fma_f64 %0, %0, %8, %9\n v_cmp_gt_i32 vcc, %10, %11\n v_fma_f64 %1, %1, %8, %9\n v_cndmask_b32_e32 v40, %10, %11, vcc\n"
                              "v_fma_f64 %2, %2, %8, %9\n v_cmp_gt_i32 vcc, %10, %11\n v_fma_f64 %3, %3, %8, %9\n v_cndmask_b32_e32 v40, %10, %11, vcc\n"
                              "v_fma_f64 %4, %4, %8, %9\n v_cmp_gt_i32 vcc, %10, %11\n v_fma_f64 %5, %5, %8, %9\n v_cndmask_b32_e32 v40, %10, %11, vcc\n"
                              "v_fma_f64 %6, %6, %8, %9\n v_cmp_gt_i32 vcc, %10, %11\n v_fma_f64 %7, %7, %8, %9\n v_cndmask_b32_e32 v40, %10, %11, vcc\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                         : "v"(a), "v"(b), "v"(n1), "v"(n0) : "vcc", "v40");
        } else if (MODE == FMA_READLANE) {
            asm volatile(REP4(FMA8_WITH("v_readfirstlane_b32 s20, %10\n"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                         : "v"(a), "v"(b), "v"(n1) : "s20");
        } else if (MODE == FMA_DSADD64) {
            asm volatile(REP4(FMA8_WITH("ds_add_u64 %11, %10\n")) "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                         : "v"(a), "v"(b), "v"(r0), "v"(la) : "memory");
        } else if (MODE == FMA_DSWRITE16) {
            asm volatile(REP4(FMA8_WITH("ds_write_b16 %11, %10\n")) "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                         : "v"(a), "v"(b), "v"(n0), "v"(la2) : "memory");
        } else if (MODE == FMA_CMP_SAND) {
            asm volatile("s_mov_b64 s[20:21], exec\n"
                         REP4("v_fma_f64 %0, %0, %8, %9\n v_cmp_gt_i32 vcc, %10, %11\n v_fma_f64 %1, %1, %8, %9\n s_and_b64 s[22:23], vcc, s[20:21]\n"
                              "v_fma_f64 %2, %2, %8, %9\n v_cmp_gt_i32 vcc, %10, %11\n v_fma_f64 %3, %3, %8, %9\n s_and_b64 s[22:23], vcc, s[20:21]\n"
                              "v_fma_f64 %4, %4, %8, %9\n v_cmp_gt_i32 vcc, %10, %11\n v_fma_f64 %5, %5, %8, %9\n s_and_b64 s[22:23], vcc, s[20:21]\n"
                              "v_fma_f64 %6, %6, %8, %9\n v_cmp_gt_i32 vcc, %10, %11\n v_fma_f64 %7, %7, %8, %9\n s_and_b64 s[22:23], vcc, s[20:21]\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                         : "v"(a), "v"(b), "v"(n1), "v"(n0) : "vcc", "s20", "s21", "s22", "s23", "scc");
        }
    }
    out[blockIdx.x * 64 + lane] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + n0 + n2 + r0 + r1;
}

template <int MODE>
static void run(double *d_out, int waves_per_simd, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 4 * waves_per_simd;
    k<MODE><<<blocks, 64>>>(d_out, 10, 0.999, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(d_out, iters, 0.999, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  %d waves/SIMD: %7.2f ns", waves_per_simd, (double)ms * 1e6 / iters / waves_per_simd);
}

template <int MODE>
static void row(double *d_out)
{
    printf("%-58s", names[MODE]);
    run<MODE>(d_out, 1, 20000);
    run<MODE>(d_out, 2, 20000);
    run<MODE>(d_out, 5, 10000);
    printf("\n");
    fflush(stdout);
}

int main()
{
    double *d_out;
    if (hipMalloc(&d_out, 256 * 4 * 8 * 64 * sizeof(double)) != hipSuccess)
        return 1;
    printf("ns of a SIMD per loop pass of one wave (32 v_fma_f64 per pass, plus what the row names)\n");
    row<FMA_ONLY>(d_out);
    row<FMA_SALU>(d_out);
    row<FMA_NOP>(d_out);
    row<FMA_VMOV>(d_out);
    row<FMA_BRANCH>(d_out);
    row<FMA_EXEC>(d_out);
    row<FMA_LDSREAD>(d_out);
    row<FMA_WAITCNT>(d_out);
    row<SALU_ONLY>(d_out);
    row<VMOV_ONLY>(d_out);
    row<FMA_DEP>(d_out);
    row<FMA_DEP_SALU>(d_out);
    row<FMA_VCMP>(d_out);
    row<FMA_EXECONLY>(d_out);
    row<FMA_EXECMOV>(d_out);
    row<FMA_CMP_CND_SGPR>(d_out);
    row<FMA_CMP_CND_VCC>(d_out);
    row<FMA_READLANE>(d_out);
    row<FMA_CMP_SAND>(d_out);
    row<FMA_DSADD64>(d_out);
    row<FMA_DSWRITE16>(d_out);
    hipFree(d_out);
    return 0;
}
