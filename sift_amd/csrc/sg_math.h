// sg_math.h -- double-precision device math for the histogram kernels (gfx950).
//
// The orientation and descriptor kernels evaluate a gradient magnitude and angle per window sample
// (reference sift/sift.cpp:1109-1136: sqrt(dx*dx+dy*dy), atan2(dy,dx)); those two calls dominate
// their instruction streams.  The forms below are accurate to about 2 ulp over the ranges that occur
// (pixel differences in [-2, 2]) and cost well under half of the library routines.
#pragma once

#include <hip/hip_runtime.h>

namespace sg {

// a*b + c as one v_fma_f64 with three register operands.  (Left to itself the compiler turns a Horner step
// "p*z + constant" into a register copy of the constant followed by the two-operand accumulate form.)
__device__ __forceinline__ double sg_fma3(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// sqrt for 0 <= x in the normal range: v_rsq_f64 seed, one coupled Newton-Raphson step (Goldschmidt form)
// and one residual correction; within 1 ulp.
__device__ __forceinline__ double sg_sqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x > 0.0 ? g : 0.0;
}

// n / d for d > 0 in the normal range: v_rcp_f64 seed, one Newton step on the reciprocal and one residual
// correction of the quotient (itself a Newton step); within 1 ulp.
__device__ __forceinline__ double sg_div(double n, double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}

// atan of a ratio mn/mx, 0 <= mn <= mx, split as  k*pi/16 + atan(u):  one of five centres c = tan(k*pi/16) is
// chosen by comparing mn with mx*tan((2k-1)*pi/32), and u = (mn - c*mx) / (mx + c*mn) has |u| <= tan(pi/32), so an
// 8-term odd series (truncation < 5e-19) gives atan(u).  One division in all; mn = mx = 0 gives k = 0, u = 0.
// `centres` is a 5-entry table of (c, caller's multiple of k) pairs, normally in LDS (sg_atan_centres_fill).
__device__ __forceinline__ double sg_atan_split(double mn, double mx, const double2 *centres, double &k_term)
{
    // the thresholds ascend, so the last one passed gives k; formed as the entry's byte offset (one select each)
    unsigned int koff = mn > 0.09849140335716425 * mx ? 16u : 0u;
    koff = mn > 0.30334668360734239 * mx ? 32u : koff;
    koff = mn > 0.53451113595079158 * mx ? 48u : koff;
    koff = mn > 0.82067879082866024 * mx ? 64u : koff;
    const double2 ck = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(centres) + koff);
    k_term = ck.y;
    const double num = __builtin_fma(-ck.x, mx, mn);
    const double den = fmax(__builtin_fma(ck.x, mn, mx), 1e-300);
    const double u = sg_div(num, den);
    const double z = u * u;
    double p = -1.0 / 15.0;
    p = sg_fma3(p, z, 1.0 / 13.0);
    p = sg_fma3(p, z, -1.0 / 11.0);
    p = sg_fma3(p, z, 1.0 / 9.0);
    p = sg_fma3(p, z, -1.0 / 7.0);
    p = sg_fma3(p, z, 1.0 / 5.0);
    p = sg_fma3(p, z, -1.0 / 3.0);
    return __builtin_fma(u * z, p, u);
}

// ---- forms for sums of many smooth terms ---------------------------------------------------------------------
// The descriptor histogram adds about 1 600 weighted samples per feature and every step from a sample to its eight
// bin weights is continuous, so a relative error of 1e-14 per sample is as good there as 1e-16 (the quantised bins sit
// 1e-9 and more away from their truncation boundaries); these forms stop one correction step earlier than the
// 1-ulp ones above.  They are not used where a rounding decision follows directly (orientation bins).

// sqrt(x), x >= 0: v_rsq_f64 seed (about 2^-23 relative) and one coupled Newton-Raphson step: relative error below
// 3e-14; exactly 0 for x = 0 (the seed of the clamped argument is finite).
__device__ __forceinline__ double sg_sqrt_sum(double x)
{
    const double y = __builtin_amdgcn_rsq(__builtin_fmax(x, 1e-300));
    const double g = x * y;
    const double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    return __builtin_fma(g, r, g);
}

// n / d, d > 0: v_rcp_f64 seed and one Newton step on the reciprocal: relative error below 3e-14
__device__ __forceinline__ double sg_div_sum(double n, double d)
{
    double r = __builtin_amdgcn_rcp(d);
    const double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return n * r;
}

// sg_atan_split with that division, the odd series cut after u^11/11, and the centre picked from a 35-entry table
// by the single-precision ratio: entry i serves
// mn/mx in [i/33, (i+1)/33) and holds (tan(k pi/16), k/4) for the k nearest to atan((i + 1/2)/33), so that
// |u| <= tan(pi/32 + 1/66) = 0.114 and the series' truncation stays below 5e-14.  Two conversions, one v_rcp_f32, two
// single-precision products and one LDS read replace four double products, four compares and four selects.  A zero
// gradient (0/0) converts to entry 0 and gives u = 0.
constexpr int kAtanLutEntries = 35;
__device__ __forceinline__ void sg_atan_lut_fill(double2 *lut, int lane, bool radians = false)
{
    if (lane < kAtanLutEntries) {
        const float mid = ((float)lane + 0.5f) * (1.0f / 33.0f);
        int k = (int)(atanf(mid) * (16.0f / 3.14159265f) + 0.5f);
        k = k > 4 ? 4 : k;
        double2 e;
        e.x = k == 0 ? 0.0 : k == 1 ? 0.19891236737965800 : k == 2 ? 0.41421356237309503 : k == 3 ? 0.66817863791929888 : 1.0;
        e.y = radians ? 0.19634954084936207 * k : 0.25 * k; // k pi/16, or k/4 in units of pi/4
        lut[lane] = e;
    }
}

__device__ __forceinline__ double sg_atan_split_lut(double mn, double mx, const double2 *lut, double &k_term)
{
    const float t = (float)mn * __builtin_amdgcn_rcpf((float)mx);
    const unsigned int idx = (unsigned int)(t * 33.0f); // NaN -> 0; a ratio that rounds above 1 -> entry 33
    const double2 ck = lut[idx < (unsigned int)kAtanLutEntries ? idx : kAtanLutEntries - 1];
    k_term = ck.y;
    const double num = __builtin_fma(-ck.x, mx, mn);
    const double den = fmax(__builtin_fma(ck.x, mn, mx), 1e-300);
    const double u = sg_div_sum(num, den);
    const double z = u * u;
    double p = -1.0 / 11.0;
    p = sg_fma3(p, z, 1.0 / 9.0);
    p = sg_fma3(p, z, -1.0 / 7.0);
    p = sg_fma3(p, z, 1.0 / 5.0);
    p = sg_fma3(p, z, -1.0 / 3.0);
    return __builtin_fma(u * z, p, u);
}

// centres[k] = (tan(k*pi/16), k*pi/16) if radians, else (tan(k*pi/16), k/4) -- the angle in units of pi/4 -- for
// k = 0..4; call with lanes 0..4 at least, then make the stores visible to the readers.
__device__ __forceinline__ void sg_atan_centres_fill(double2 *centres, int lane, bool radians)
{
    if (lane < 5) {
        double2 e;
        e.x = lane == 0 ? 0.0
            : lane == 1 ? 0.19891236737965800
            : lane == 2 ? 0.41421356237309503
            : lane == 3 ? 0.66817863791929888
                        : 1.0;
        const double rad = lane == 0 ? 0.0
                         : lane == 1 ? 0.19634954084936207
                         : lane == 2 ? 0.39269908169872414
                         : lane == 3 ? 0.58904862254808621
                                     : 0.78539816339744828;
        e.y = radians ? rad : 0.25 * lane;
        centres[lane] = e;
    }
}

// atan2(y, x) in radians; centres filled with radians = true.  atan2(0,0) = 0 like the C library.
__device__ __forceinline__ double sg_atan2(double y, double x, const double2 *centres)
{
    const double ax = fabs(x), ay = fabs(y);
    const double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    double off;
    const double a = sg_atan_split(mn, mx, centres, off);
    double r = off + a;
    if (ay > ax)
        r = 1.5707963267948966 - r;
    if (x < 0.0)
        r = 3.1415926535897932 - r;
    return y < 0.0 ? -r : r;
}

} // namespace sg
