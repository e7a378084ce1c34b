// sg_dd.h -- double-double evaluation of the two library functions whose last bit decides something discrete on the
// path: pow(2, y) (a feature's scale is the sort key of the output order, reference sift/sift.cpp:1065-1077, 1640-1643)
// and atan2 next to an orientation-bin boundary (cvRound of the bin index, sift.cpp:1190).  The reference takes both
// from the C library, whose results are correctly rounded except when the exact value lies within a few hundredths of
// an ulp of a rounding boundary; a correctly rounded result from here therefore agrees with it in all but such cases,
// where a faithful (1 ulp) routine would agree in only about half of the cases that matter.  Used once per keypoint
// and in a rarely taken branch: speed is no concern.  Plain C++ apart from SG_DD_FN, so that a host build can check it
// against the C library (tests/test_cpu_dd.py).
#pragma once

#ifndef SG_DD_FN
#define SG_DD_FN __host__ __device__ __forceinline__
#endif

namespace sg {

struct dd {
    double hi, lo;
};

SG_DD_FN dd dd_two_sum(double a, double b)
{
    const double s = a + b, bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
SG_DD_FN dd dd_quick_two_sum(double a, double b) // |a| >= |b|
{
    const double s = a + b;
    return dd{s, b - (s - a)};
}
SG_DD_FN dd dd_two_prod(double a, double b)
{
    const double p = a * b;
    return dd{p, __builtin_fma(a, b, -p)};
}
SG_DD_FN dd dd_add(dd a, dd b)
{
    dd s = dd_two_sum(a.hi, b.hi);
    const dd t = dd_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = dd_quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return dd_quick_two_sum(s.hi, s.lo);
}
SG_DD_FN dd dd_add_d(dd a, double b)
{
    dd s = dd_two_sum(a.hi, b);
    s.lo += a.lo;
    return dd_quick_two_sum(s.hi, s.lo);
}
SG_DD_FN dd dd_mul(dd a, dd b)
{
    dd p = dd_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return dd_quick_two_sum(p.hi, p.lo);
}
SG_DD_FN dd dd_mul_d(dd a, double b)
{
    dd p = dd_two_prod(a.hi, b);
    p.lo += a.lo * b;
    return dd_quick_two_sum(p.hi, p.lo);
}
SG_DD_FN dd dd_div(dd a, dd b)
{
    const double q1 = a.hi / b.hi;
    dd r = dd_add(a, dd_mul_d(b, -q1));
    const double q2 = r.hi / b.hi;
    r = dd_add(r, dd_mul_d(b, -q2));
    const double q3 = r.hi / b.hi;
    dd q = dd_quick_two_sum(q1, q2);
    return dd_add_d(q, q3);
}
SG_DD_FN dd dd_neg(dd a) { return dd{-a.hi, -a.lo}; }

// 2^y correctly rounded (the evaluation's own relative error is about 2^-100), |y| < 1000
SG_DD_FN double sg_exp2_cr(double y)
{
    const double n = __builtin_rint(y);
    const double f = y - n; // exact, |f| <= 1/2
    // z = f ln 2 / 8, |z| < 0.0434; exp(z) by its Taylor series to z^15 (Horner), then three squarings
    dd z = dd_mul_d(dd{0.6931471805599453, 2.3190468138462996e-17}, f);
    z.hi *= 0.125;
    z.lo *= 0.125;
    const dd inv_fact[16] = {
        {1.0, 0.0},
        {1.0, 0.0},
        {0.5, 0.0},
        {0.16666666666666666, 9.25185853854297e-18},
        {0.041666666666666664, 2.3129646346357427e-18},
        {0.008333333333333333, 1.1564823173178714e-19},
        {0.001388888888888889, -5.300543954373577e-20},
        {0.0001984126984126984, 1.7209558293420705e-22},
        {2.48015873015873e-05, 2.1511947866775882e-23},
        {2.7557319223985893e-06, -1.858393274046472e-22},
        {2.755731922398589e-07, 2.3767714622250297e-23},
        {2.505210838544172e-08, -1.448814070935912e-24},
        {2.08767569878681e-09, -1.20734505911326e-25},
        {1.6059043836821613e-10, 1.2585294588752098e-26},
        {1.1470745597729725e-11, 2.0655512752830745e-28},
        {7.647163731819816e-13, 7.03872877733453e-30},
    };
    dd s = inv_fact[15];
    for (int k = 14; k >= 0; k--)
        s = dd_add(dd_mul(s, z), inv_fact[k]);
    s = dd_mul(s, s);
    s = dd_mul(s, s);
    s = dd_mul(s, s);
    return __builtin_ldexp(s.hi, (int)n); // s.hi is hi + lo rounded to nearest
}

// atan2(y, x) correctly rounded in the same sense (finite arguments; atan2(0, 0) = 0 like the C library for +0)
SG_DD_FN double sg_atan2_cr(double y, double x)
{
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    if (mx == 0.0)
        return 0.0;
    const dd t = dd_div(dd{mn, 0.0}, dd{mx, 0.0}); // in [0, 1]
    // nearest centre c_k = tan(k pi/16), k = 0..4: atan t = k pi/16 + atan u, u = (t - c)/(1 + c t), |u| <= tan(pi/32)
    const dd tan_k[5] = {{0.0, 0.0},
                         {0.198912367379658, 8.391794477636538e-19},
                         {0.41421356237309503, 1.4349369327986523e-17},
                         {0.6681786379192989, 4.1042270233610004e-17},
                         {1.0, 0.0}};
    const dd ang_k[5] = {{0.0, 0.0},
                         {0.19634954084936207, 7.654042494670958e-18},
                         {0.39269908169872414, 1.5308084989341915e-17},
                         {0.5890486225480862, 2.296212748401287e-17},
                         {0.7853981633974483, 3.061616997868383e-17}};
    int k = t.hi > 0.09849140335716425 ? 1 : 0;
    k = t.hi > 0.30334668360734239 ? 2 : k;
    k = t.hi > 0.53451113595079158 ? 3 : k;
    k = t.hi > 0.82067879082866024 ? 4 : k;
    const dd c = tan_k[k];
    const dd u = dd_div(dd_add(t, dd_neg(c)), dd_add_d(dd_mul(c, t), 1.0));
    const dd z = dd_mul(u, u);
    const dd odd[18] = {
        {1.0, 0.0},
        {0.3333333333333333, 1.850371707708594e-17},
        {0.2, -1.1102230246251566e-17},
        {0.14285714285714285, 7.93016446160826e-18},
        {0.1111111111111111, 6.1679056923619804e-18},
        {0.09090909090909091, -2.523234146875356e-18},
        {0.07692307692307693, -4.270088556250602e-18},
        {0.06666666666666667, 9.251858538542971e-19},
        {0.058823529411764705, 8.163404592832033e-19},
        {0.05263157894736842, 2.921639538487254e-18},
        {0.047619047619047616, 2.64338815386942e-18},
        {0.043478260869565216, 1.206764157201257e-18},
        {0.04, -8.326672684688674e-19},
        {0.037037037037037035, 2.05596856412066e-18},
        {0.034482758620689655, 4.785444071660157e-19},
        {0.03225806451612903, 8.953411488912552e-19},
        {0.030303030303030304, -8.410780489584519e-19},
        {0.02857142857142857, 8.921435019309293e-19},
    };
    // atan u = u (1 - z/3 + z^2/5 - ...): Horner in -z
    const dd mz = dd_neg(z);
    dd p = odd[17];
    for (int j = 16; j >= 0; j--)
        p = dd_add(dd_mul(p, mz), odd[j]);
    dd r = dd_add(ang_k[k], dd_mul(u, p));
    if (ay > ax)
        r = dd_add(dd{1.5707963267948966, 6.123233995736766e-17}, dd_neg(r));
    if (x < 0.0)
        r = dd_add(dd{3.141592653589793, 1.2246467991473532e-16}, dd_neg(r));
    return y < 0.0 ? -r.hi : r.hi;
}

} // namespace sg
