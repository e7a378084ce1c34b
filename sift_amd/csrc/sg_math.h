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

// a*b + c with c in scalar registers (a wave-uniform value, e.g. a constant): no vector register is spent on it.
__device__ __forceinline__ double sg_fma3s(double a, double b, double c_uniform)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
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
//
// sqrt(x), x >= 0, in that class (written out in the descriptor kernel's sample loop, which also needs its parts):
//   y = v_rsq_f64(max(x, 1e-300))   the seed, about 2^-23 relative
//   g = x y, h = y / 2, r = 1/2 - h g, sqrt = g + g r      one coupled Newton-Raphson step: relative error below 3e-14,
// exactly 0 for x = 0 (the seed of the clamped argument is finite).

// Direction of a gradient from its magnitude's reciprocal, with no division.  The sample loop already holds
// y ~ 1/sqrt(dx^2+dy^2) (the v_rsq_f64 seed) and the residual r of its Newton step, so sin(theta) = mn*y for the
// angle theta in [0, pi/4] of (mx, mn) costs one product.  sin(theta), rounded to a multiple of 1/32 by adding
// 2^47 (the sum's low mantissa bits are then the integer k = 0..23), picks a centre phi_k with sin(phi_k) = k/32; the
// table holds cos(phi_k) and phi_k in units of pi/4 (sin(phi_k) = k/32 comes back out of the sum, exactly).  Then
//   u = sin(theta - phi_k) = (mn * cos(phi_k) - mx * k/32) / mag,   |u| <= 0.0221,
// with 1/mag = y(1 + r), and theta - phi_k = asin(u) by its series to u^7 (truncation below 1e-16).  The
// reciprocal carries the seed's error squared (2^-45), which the size of u reduces to 7e-16 of a radian.
// Eleven double-precision operations in all, against a v_rcp_f64, fifteen more and a single-precision index computation
// for the quotient form.  A zero gradient gives k = 0, U = 0 and the angle 0.
constexpr int kSinLutEntries = 24;
__device__ const double kSinLut[kSinLutEntries][2] = { // (generated by tools/gen_sin_lut.py)
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.ffbffbff7fec0p-1, 0x1.46009d2eee445p-5},
    {0x1.feffbfdfebf1fp-1, 0x1.462972893d260p-4},
    {0x1.fdbeba917c3f5p-1, 0x1.e9a49d9c4ecf6p-4},
    {0x1.fbfbf7ebc755fp-1, 0x1.46cdde5cc9717p-3},
    {0x1.f9b61d0237250p-1, 0x1.991ceda75d74cp-3},
    {0x1.f6eb62d27730dp-1, 0x1.ebd5c6ffa1500p-3},
    {0x1.f3998f1b1886cp-1, 0x1.1f87d9e8cdc56p-2},
    {0x1.efbdeb14f4edap-1, 0x1.49719f07a5377p-2},
    {0x1.eb5537b1434dap-1, 0x1.73b53d57e94afp-2},
    {0x1.e65b9edeba38ep-1, 0x1.9e60aff8536d7p-2},
    {0x1.e0cca12e97895p-1, 0x1.c9831560c195bp-2},
    {0x1.daa2fefaae1d8p-1, 0x1.f52cebe2d7795p-2},
    {0x1.d3d89be176072p-1, 0x1.10b82d6ef85b0p-1},
    {0x1.cc665b0328622p-1, 0x1.2730c645f555fp-1},
    {0x1.c443f1d4d22afp-1, 0x1.3e0b8edacdea9p-1},
    {0x1.bb67ae8584caap-1, 0x1.5555555555556p-1},
    {0x1.b1c62db2564fep-1, 0x1.6d1cc1f0b5912p-1},
    {0x1.a751f9447b724p-1, 0x1.8572cc9d34757p-1},
    {0x1.9bfb076d236ebp-1, 0x1.9e6b58ffe0d2fp-1},
    {0x1.8fae0c15ad38ap-1, 0x1.b81e0985cc8eap-1},
    {0x1.8253878ae2e09p-1, 0x1.d2a7638e7e783p-1},
    {0x1.73ce704fb7b23p-1, 0x1.ee2a6e8b77d33p-1},
    {0x1.63fa3f3c02962p-1, 0x1.05698a0e94436p+0},
};
constexpr double kAsinC0 = 0x1.45f306dc9c883p+0; // 4/pi: asin(u)/(pi/4) = u (c0 + c1 u^2 + c2 u^4 + c3 u^6)
constexpr double kAsinC1 = 0x1.b2995e7b7b604p-3; // c0 / 6
constexpr double kAsinC2 = 0x1.8723a1d588a36p-4; // c0 * 3/40
constexpr double kAsinC3 = 0x1.d1a452f204304p-5; // c0 * 5/112

// Copies the table into LDS; call with `lane` covering 0..kSinLutEntries-1, then make the stores visible.
__device__ __forceinline__ void sg_sin_lut_fill(double2 *lut, int lane)
{
    if (lane < kSinLutEntries) {
        double2 e;
        e.x = kSinLut[lane][0];
        e.y = kSinLut[lane][1];
        lut[lane] = e;
    }
}

// The table entry for the direction of (mx, mn), 0 <= mn <= mx: its byte offset (16 k), and the centre's sine k/32.
// y = the reciprocal-square-root seed of mx^2 + mn^2 (clamped away from zero).
__device__ __forceinline__ unsigned int sg_sin_lut_index(double mn, double y, double &sin_k)
{
    const double t = __builtin_fma(mn, y, 0x1p47);
    sin_k = t - 0x1p47;
    return (unsigned int)__double_as_longlong(t) << 4;
}

// atan(mn/mx) in units of pi/4 from the entry ck = {cos, angle} and mxk = mx * sin_k; r = the residual
// of the Newton step on y (1/mag = y + y*r).
__device__ __forceinline__ double sg_angle_sin_lut(double mn, double mxk, double y, double r, double2 ck)
{
    const double w = __builtin_fma(mn, ck.x, -mxk);
    const double wy = w * y;
    const double U = __builtin_fma(wy, r, wy);
    const double z = U * U;
    double p = sg_fma3s(z, kAsinC3, kAsinC2); // (kAsinC3 in vector registers, the other three coefficients in scalar ones)
    p = sg_fma3s(p, z, kAsinC1);
    p = sg_fma3s(p, z, kAsinC0);
    return __builtin_fma(U, p, ck.y);
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
