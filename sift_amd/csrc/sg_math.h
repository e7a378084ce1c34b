// sg_math.h -- double-precision device math for the histogram kernels (gfx950).
//
// The orientation and descriptor kernels evaluate a gradient magnitude and angle per window sample
// (reference sift/sift.cpp:1109-1136: sqrt(dx*dx+dy*dy), atan2(dy,dx)); those two calls dominate
// their instruction streams.  The forms below are accurate to about 2 ulp over the ranges that occur
// (pixel differences in [-2, 2]) and cost well under half of the library routines.
#pragma once

#include <hip/hip_runtime.h>

namespace sg {

// sqrt for 0 <= x in the normal range: v_rsq_f64 seed, two Newton-Raphson steps (Goldschmidt form).
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
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x > 0.0 ? g : 0.0;
}

// atan2(y, x): octant folding to t = min/max in [0,1], then one of five centres c = tan(k*pi/16) with
// u = (mn - c*mx) / (mx + c*mn), |u| <= tan(pi/32), so atan(t) = k*pi/16 + atan(u) with an 8-term odd
// series (truncation < 5e-19).  One division in all.  atan2(0,0) = 0 like the C library.
__device__ __forceinline__ double sg_atan2(double y, double x)
{
    const double ax = fabs(x), ay = fabs(y);
    const double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    // interval boundaries tan((2k+1)*pi/32), centres tan(k*pi/16), offsets k*pi/16
    double c = 0.0, off = 0.0;
    if (mn > 0.09849140335716425 * mx) { c = 0.19891236737965800; off = 0.19634954084936207; }
    if (mn > 0.30334668360734239 * mx) { c = 0.41421356237309503; off = 0.39269908169872414; }
    if (mn > 0.53451113595079158 * mx) { c = 0.66817863791929888; off = 0.58904862254808621; }
    if (mn > 0.82067879082866024 * mx) { c = 1.0; off = 0.78539816339744828; }
    const double num = __builtin_fma(-c, mx, mn);
    const double den = __builtin_fma(c, mn, mx);
    const double u = den > 0.0 ? num / den : 0.0;
    const double z = u * u;
    double p = -1.0 / 15.0;
    p = __builtin_fma(p, z, 1.0 / 13.0);
    p = __builtin_fma(p, z, -1.0 / 11.0);
    p = __builtin_fma(p, z, 1.0 / 9.0);
    p = __builtin_fma(p, z, -1.0 / 7.0);
    p = __builtin_fma(p, z, 1.0 / 5.0);
    p = __builtin_fma(p, z, -1.0 / 3.0);
    double r = off + __builtin_fma(u * z, p, u);
    if (ay > ax)
        r = 1.5707963267948966 - r;
    if (x < 0.0)
        r = 3.1415926535897932 - r;
    return y < 0.0 ? -r : r;
}

} // namespace sg
