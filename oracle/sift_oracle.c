/*
 * sift_oracle.c -- CPU oracle for the SIFT extraction path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C99, double precision, scalar, one thread per image: a restatement of the algorithm
 * of the reference's sift/sift.cpp, function by function, with the reference lines cited at
 * each function.  Used as the checker by tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py; never by the product library.
 *
 * PARITY UNPINNED (see sift_oracle.h): the reference holds no tests or fixtures and cannot be
 * built here (OpenCV absent); cv::GaussianBlur is restated from OpenCV's published algorithm.
 *
 * Build with -ffp-contract=off: every multiply and add below rounds separately, as the
 * reference's generic x86-64 code does, and the GPU path's double-precision mode reproduces
 * the same operation order.
 */
#define _GNU_SOURCE
#include "sift_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* constants: sift/sift.h:19-70 */
#define SO_INIT_SIGMA 0.5
#define SO_INTERVALS 3
#define SO_RATIO 10
#define SO_MAX_INTERP_STEPS 5
#define SO_DXTHRESHOLD 0.03
#define SO_ORI_HIST_BINS 36
#define SO_ORI_SIGMA_TIMES 1.5
#define SO_ORI_SMOOTH_TIMES 2
#define SO_ORI_PEAK_RATIO 0.8
#define SO_DESCR_HIST_BINS 8
#define SO_IMG_BORDER 5
#define SO_DESCR_WINDOW_WIDTH 4
#define SO_DESCR_SCALE_ADJUST 3
#define SO_DESCR_MAG_THR 0.2
#define SO_INT_DESCR_FCTR 512.0
#define SO_PI 3.1415926535897932384626433832795 /* CV_PI */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* cvRound: round half to even (SSE2 cvtsd2si / lrint under the default rounding mode). */
static int so_round(double v) { return (int)lrint(v); }
static int so_floor(double v) { return (int)floor(v); }

size_t so_sizeof_keypoint(void) { return sizeof(so_keypoint); }
void so_free(void *p) { free(p); }

/* ------------------------------------------------------------------ image preparation */

/* sift.cpp:20-51 ConvertToGray: b,g,r each divided by 255.0, then b*0.114 + g*0.587 + r*0.299. */
void so_convert_to_gray(const uint8_t *src, int rows, int cols, int channels, size_t step, double *dst)
{
    for (int i = 0; i < rows; i++) {
        for (int j = 0; j < cols; j++) {
            const uint8_t *p = src + step * (size_t)i + (size_t)channels * (size_t)j;
            double b = p[0] / 255.0;
            double g = p[1] / 255.0;
            double r = p[2] / 255.0;
            dst[(size_t)i * cols + j] = b * 0.114 + g * 0.587 + r * 0.299;
        }
    }
}

/* sift.cpp:111-175 UpSample: 2x, interior from the (rows-1)x(cols-1) source cells, then the last
 * two columns copy column 2C-3 (all rows), then the last two rows copy row 2R-3 (all columns). */
void so_upsample(const double *src, int rows, int cols, double *dst)
{
    int R2 = rows * 2, C2 = cols * 2;
    for (int j = 0; j < cols - 1; j++) {
        for (int i = 0; i < rows - 1; i++) {
            int m = 2 * i, n = 2 * j;
            double s00 = src[(size_t)i * cols + j];
            double s10 = src[(size_t)(i + 1) * cols + j];
            double s01 = src[(size_t)i * cols + j + 1];
            double s11 = src[(size_t)(i + 1) * cols + j + 1];
            dst[(size_t)m * C2 + n] = s00;
            double rs = s00 + s10;
            dst[(size_t)(m + 1) * C2 + n] = rs / 2;
            double cs = s00 + s01;
            dst[(size_t)m * C2 + n + 1] = cs / 2;
            double center = s10 + s00 + s11 + s01;
            dst[(size_t)(m + 1) * C2 + n + 1] = center / 4;
        }
    }
    if (R2 < 3 || C2 < 3)
        return;
    for (int k = R2 - 1; k >= 0; k--) {
        dst[(size_t)k * C2 + C2 - 2] = dst[(size_t)k * C2 + C2 - 3];
        dst[(size_t)k * C2 + C2 - 1] = dst[(size_t)k * C2 + C2 - 3];
    }
    for (int k = C2 - 1; k >= 0; k--) {
        dst[(size_t)(R2 - 2) * C2 + k] = dst[(size_t)(R2 - 3) * C2 + k];
        dst[(size_t)(R2 - 1) * C2 + k] = dst[(size_t)(R2 - 3) * C2 + k];
    }
}

/* sift.cpp:59-103 DownSample: dst(m,n) = src(2m,2n), dst is floor(rows/2) x floor(cols/2);
 * a source with a dimension <= 1 is copied (not reached by the pipeline). */
void so_downsample(const double *src, int rows, int cols, double *dst)
{
    if (cols <= 1 || rows <= 1) {
        memcpy(dst, src, sizeof(double) * (size_t)rows * cols);
        return;
    }
    int dr = rows / 2, dc = cols / 2;
    for (int m = 0; m < dr; m++)
        for (int n = 0; n < dc; n++)
            dst[(size_t)m * dc + n] = src[(size_t)(2 * m) * cols + 2 * n];
}

/* ------------------------------------------------------------------ Gaussian blur */

/* OpenCV GaussianBlur with Size(0,0): ksize = cvRound(sigma*4*2 + 1) | 1 for non-8-bit depth. */
int so_gaussian_ksize(double sigma) { return so_round(sigma * 4 * 2 + 1) | 1; }

/* OpenCV getGaussianKernel (sigma > 0): k[i] = exp(-0.5*x^2/sigma^2), x = i-(n-1)/2, scaled by 1/sum. */
void so_gaussian_kernel(double sigma, int n, double *k)
{
    double scale2x = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = exp(scale2x * x * x);
        k[i] = t;
        sum += t;
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++)
        k[i] *= sum;
}

/* OpenCV borderInterpolate, BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba), repeated until in range. */
static int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len) {
        if (p < 0)
            p = -p;
        else
            p = 2 * (len - 1) - p;
    }
    return p;
}

/* sift.cpp:323-326 GaussianSmooth -> cv::GaussianBlur(src,dst,Size(0,0),sigma,sigma), CV_64FC1.
 * Separable: the generic row filter  s = k[0]*S[0]; s += k[t]*S[t]  (left to right) into an f64
 * intermediate, then the symmetric column filter  s = k[h]*C; s += k[h+t]*(row[y+t] + row[y-t]). */
void so_gaussian_blur(const double *src, double *dst, int rows, int cols, double sigma)
{
    int n = so_gaussian_ksize(sigma);
    if (n <= 1) {
        if (dst != src)
            memcpy(dst, src, sizeof(double) * (size_t)rows * cols);
        return;
    }
    int h = n / 2;
    double *k = (double *)malloc(sizeof(double) * (size_t)n);
    so_gaussian_kernel(sigma, n, k);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)rows * cols);
    double *ext = (double *)malloc(sizeof(double) * (size_t)(cols + 2 * h));
    int *ytab = (int *)malloc(sizeof(int) * (size_t)(rows + 2 * h));

    /* row pass */
    for (int y = 0; y < rows; y++) {
        const double *s = src + (size_t)y * cols;
        for (int x = -h; x < cols + h; x++)
            ext[x + h] = s[reflect101(x, cols)];
        double *t = tmp + (size_t)y * cols;
        if (cols == 1) { /* OpenCV shrinks the kernel to 1 tap along a 1-pixel dimension */
            t[0] = s[0];
            continue;
        }
        for (int x = 0; x < cols; x++) {
            const double *e = ext + x;
            double acc = k[0] * e[0];
            for (int i = 1; i < n; i++)
                acc += k[i] * e[i];
            t[x] = acc;
        }
    }
    /* column pass */
    for (int y = -h; y < rows + h; y++)
        ytab[y + h] = reflect101(y, rows);
    for (int y = 0; y < rows; y++) {
        double *d = dst + (size_t)y * cols;
        const double *c = tmp + (size_t)y * cols;
        if (rows == 1) {
            memcpy(d, c, sizeof(double) * (size_t)cols);
            continue;
        }
        for (int x = 0; x < cols; x++)
            d[x] = k[h] * c[x];
        for (int i = 1; i <= h; i++) {
            const double *a = tmp + (size_t)ytab[y + i + h] * cols;
            const double *b = tmp + (size_t)ytab[y - i + h] * cols;
            double ki = k[h + i];
            for (int x = 0; x < cols; x++)
                d[x] += ki * (a[x] + b[x]);
        }
    }
    free(ytab);
    free(ext);
    free(tmp);
    free(k);
}

/* ------------------------------------------------------------------ base image, pyramids */

/* sift.cpp:335-346 CreateInitSmoothGray: gray -> upsample -> blur with sqrt(sigma^2 - (2*0.5)^2). */
void so_create_init_smooth_gray(const uint8_t *src, int rows, int cols, int channels, size_t step,
                                double sigma, double *dst)
{
    double *gray = (double *)malloc(sizeof(double) * (size_t)rows * cols);
    double *up = (double *)malloc(sizeof(double) * (size_t)rows * cols * 4);
    so_convert_to_gray(src, rows, cols, channels, step, gray);
    so_upsample(gray, rows, cols, up);
    double sigma_init = sqrt(sigma * sigma - (SO_INIT_SIGMA * 2) * (SO_INIT_SIGMA * 2));
    so_gaussian_blur(up, dst, rows * 2, cols * 2, sigma_init);
    free(up);
    free(gray);
}

/* sift.cpp:1653: int octaves = log((double)min(rows,cols)) / log(2.0) - 2;  (truncating) */
int so_num_octaves(int base_rows, int base_cols)
{
    int m = base_rows < base_cols ? base_rows : base_cols;
    return (int)(log((double)m) / log(2.0) - 2);
}

/* sift.cpp:361-372: sigmas[0]=sigma; sigmas[i]=sqrt(tot^2-prev^2), prev=k^(i-1)*sigma, tot=prev*k. */
void so_pyramid_sigmas(double sigma, int intervals, double *sigmas)
{
    double k = pow(2.0, 1.0 / intervals);
    sigmas[0] = sigma;
    for (int i = 1; i < intervals + 3; i++) {
        double sig_prev = pow(k, i - 1) * sigma;
        double sig_total = sig_prev * k;
        sigmas[i] = sqrt(sig_total * sig_total - sig_prev * sig_prev);
    }
}

/* sift.cpp:357-407 GaussianPyramid, :416-461 Sub / DogPyramid. */
so_pyramid *so_build_pyramid(const double *base, int rows, int cols, int octaves, int intervals, double sigma)
{
    int S = intervals;
    so_pyramid *p = (so_pyramid *)calloc(1, sizeof(so_pyramid));
    p->octaves = octaves;
    p->intervals = S;
    p->rows = (int *)calloc((size_t)octaves + 1, sizeof(int));
    p->cols = (int *)calloc((size_t)octaves + 1, sizeof(int));
    p->gauss = (double **)calloc((size_t)octaves * (S + 3) + 1, sizeof(double *));
    p->dog = (double **)calloc((size_t)octaves * (S + 2) + 1, sizeof(double *));
    double *sigmas = (double *)malloc(sizeof(double) * (size_t)(S + 3));
    so_pyramid_sigmas(sigma, S, sigmas);

    for (int o = 0; o < octaves; o++) {
        if (o == 0) {
            p->rows[0] = rows;
            p->cols[0] = cols;
        } else {
            int pr = p->rows[o - 1], pc = p->cols[o - 1];
            if (pr <= 1 || pc <= 1) { /* DownSample copies (sift.cpp:64-68) */
                p->rows[o] = pr;
                p->cols[o] = pc;
            } else {
                p->rows[o] = pr / 2;
                p->cols[o] = pc / 2;
            }
        }
        size_t npx = (size_t)p->rows[o] * p->cols[o];
        for (int i = 0; i < S + 3; i++) {
            double *m = (double *)malloc(sizeof(double) * npx);
            if (o == 0 && i == 0)
                memcpy(m, base, sizeof(double) * npx);
            else if (i == 0)
                so_downsample(p->gauss[(o - 1) * (S + 3) + S], p->rows[o - 1], p->cols[o - 1], m);
            else
                so_gaussian_blur(p->gauss[o * (S + 3) + i - 1], m, p->rows[o], p->cols[o], sigmas[i]);
            p->gauss[o * (S + 3) + i] = m;
        }
    }
    free(sigmas);
    return p;
}

static void build_dog(so_pyramid *p)
{
    int S = p->intervals;
    for (int o = 0; o < p->octaves; o++) {
        size_t npx = (size_t)p->rows[o] * p->cols[o];
        for (int i = 1; i < S + 3; i++) {
            const double *a = p->gauss[o * (S + 3) + i];
            const double *b = p->gauss[o * (S + 3) + i - 1];
            double *c = (double *)malloc(sizeof(double) * npx);
            for (size_t q = 0; q < npx; q++)
                c[q] = a[q] - b[q];
            p->dog[o * (S + 2) + i - 1] = c;
        }
    }
}

void so_free_pyramid(so_pyramid *p)
{
    if (!p)
        return;
    int S = p->intervals;
    for (int i = 0; i < p->octaves * (S + 3); i++)
        free(p->gauss[i]);
    for (int i = 0; i < p->octaves * (S + 2); i++)
        free(p->dog[i]);
    free(p->gauss);
    free(p->dog);
    free(p->rows);
    free(p->cols);
    free(p);
}

/* ------------------------------------------------------------------ detection */

#define DOG(p, index, o, x, y) ((p)->dog[index][(size_t)(y) * (p)->cols[o] + (x)])

/* sift.cpp:473-526 isExtremum (3x3x3, non-strict, centre included) and :528-581 (in-layer 3x3). */
static int is_extremum(const so_pyramid *p, int o, int index, int x, int y, int nointerval)
{
    double val = DOG(p, index, o, x, y);
    int lo = nointerval ? 0 : -1, hi = nointerval ? 0 : 1;
    if (val > 0) {
        for (int i = lo; i <= hi; i++)
            for (int j = -1; j <= 1; j++)
                for (int k = -1; k <= 1; k++)
                    if (val < DOG(p, index + i, o, x + k, y + j))
                        return 0;
    } else {
        for (int i = lo; i <= hi; i++)
            for (int j = -1; j <= 1; j++)
                for (int k = -1; k <= 1; k++)
                    if (val > DOG(p, index + i, o, x + k, y + j))
                        return 0;
    }
    return 1;
}

/* sift.cpp:600-630 passEdgeResponse, r = 10. */
static int pass_edge_response(const so_pyramid *p, int o, int index, int x, int y)
{
    double r = SO_RATIO;
    double val = DOG(p, index, o, x, y);
    double Dxx = DOG(p, index, o, x + 1, y) + DOG(p, index, o, x - 1, y) - 2 * val;
    double Dyy = DOG(p, index, o, x, y + 1) + DOG(p, index, o, x, y - 1) - 2 * val;
    double Dxy = (DOG(p, index, o, x + 1, y + 1) + DOG(p, index, o, x - 1, y - 1) -
                  DOG(p, index, o, x - 1, y + 1) - DOG(p, index, o, x + 1, y - 1)) / 4.0;
    double Tr_h = Dxx + Dyy;
    double Det_h = Dxx * Dyy - Dxy * Dxy;
    if (Det_h <= 0)
        return 0;
    if (Tr_h * Tr_h / Det_h < (r + 1) * (r + 1) / r)
        return 1;
    return 0;
}

/* sift.cpp:653-662 DerivativeOf3D */
static void derivative3d(const so_pyramid *p, int o, int index, int x, int y, double *dx)
{
    dx[0] = (DOG(p, index, o, x + 1, y) - DOG(p, index, o, x - 1, y)) / 2.0;
    dx[1] = (DOG(p, index, o, x, y + 1) - DOG(p, index, o, x, y - 1)) / 2.0;
    dx[2] = (DOG(p, index + 1, o, x, y) - DOG(p, index - 1, o, x, y)) / 2.0;
}

/* sift.cpp:677-703 Hessian3D */
static void hessian3d(const so_pyramid *p, int o, int index, int x, int y, double *H)
{
    double val = DOG(p, index, o, x, y);
    double Dxx = DOG(p, index, o, x + 1, y) + DOG(p, index, o, x - 1, y) - 2 * val;
    double Dyy = DOG(p, index, o, x, y + 1) + DOG(p, index, o, x, y - 1) - 2 * val;
    double Dss = DOG(p, index + 1, o, x, y) + DOG(p, index - 1, o, x, y) - 2 * val;
    double Dxy = (DOG(p, index, o, x + 1, y + 1) + DOG(p, index, o, x - 1, y - 1) -
                  DOG(p, index, o, x + 1, y - 1) - DOG(p, index, o, x - 1, y + 1)) / 4.0;
    double Dxs = (DOG(p, index + 1, o, x + 1, y) + DOG(p, index - 1, o, x - 1, y) -
                  DOG(p, index - 1, o, x + 1, y) - DOG(p, index + 1, o, x - 1, y)) / 4.0;
    double Dys = (DOG(p, index + 1, o, x, y + 1) + DOG(p, index - 1, o, x, y - 1) -
                  DOG(p, index + 1, o, x, y - 1) - DOG(p, index - 1, o, x, y + 1)) / 4.0;
    H[0] = Dxx; H[4] = Dyy; H[8] = Dss;
    H[3] = H[1] = Dxy;
    H[6] = H[2] = Dxs;
    H[7] = H[5] = Dys;
}

/* sift.cpp:720-764 Inverse3D: adjugate / determinant; 0 (and H_inv untouched) if |det| < 1e-10. */
int so_inverse3d(const double *H, double *Hi)
{
#define HA(i, j) (H[(i) * 3 + (j)])
    double A = HA(0, 0) * HA(1, 1) * HA(2, 2) + HA(0, 1) * HA(1, 2) * HA(2, 0) + HA(0, 2) * HA(1, 0) * HA(2, 1) -
               HA(0, 0) * HA(1, 2) * HA(2, 1) - HA(0, 1) * HA(1, 0) * HA(2, 2) - HA(0, 2) * HA(1, 1) * HA(2, 0);
    if (fabs(A) < 1e-10)
        return 0;
    Hi[0] = HA(1, 1) * HA(2, 2) - HA(2, 1) * HA(1, 2);
    Hi[1] = -(HA(0, 1) * HA(2, 2) - HA(2, 1) * HA(0, 2));
    Hi[2] = HA(0, 1) * HA(1, 2) - HA(0, 2) * HA(1, 1);
    Hi[3] = HA(1, 2) * HA(2, 0) - HA(2, 2) * HA(1, 0);
    Hi[4] = -(HA(0, 2) * HA(2, 0) - HA(0, 0) * HA(2, 2));
    Hi[5] = HA(0, 2) * HA(1, 0) - HA(0, 0) * HA(1, 2);
    Hi[6] = HA(1, 0) * HA(2, 1) - HA(1, 1) * HA(2, 0);
    Hi[7] = -(HA(0, 0) * HA(2, 1) - HA(0, 1) * HA(2, 0));
    Hi[8] = HA(0, 0) * HA(1, 1) - HA(0, 1) * HA(1, 0);
    for (int i = 0; i < 9; i++)
        Hi[i] /= A;
#undef HA
    return 1;
}

/* sift.cpp:775-795 GetOffsetX: offset = -H^-1 * dD; a singular Hessian leaves H^-1 = 0 => zero offset. */
static void get_offset_x(const so_pyramid *p, int o, int index, int x, int y, double *off)
{
    double H[9], Hi[9] = {0}, dx[3];
    hessian3d(p, o, index, x, y, H);
    so_inverse3d(H, Hi);
    derivative3d(p, o, index, x, y, dx);
    for (int i = 0; i < 3; i++) {
        off[i] = 0.0;
        for (int j = 0; j < 3; j++)
            off[i] += Hi[i * 3 + j] * dx[j];
        off[i] = -off[i];
    }
}

/* sift.cpp:808-823 GetFabsDx: |D + 0.5 * dD . offset| */
static double get_fabs_dx(const so_pyramid *p, int o, int index, int x, int y, const double *off)
{
    double dx[3];
    derivative3d(p, o, index, x, y, dx);
    double term = 0.0;
    for (int i = 0; i < 3; i++)
        term += dx[i] * off[i];
    double val = DOG(p, index, o, x, y);
    return fabs(val + 0.5 * term);
}

/* cvRound of an offset far outside int range yields INT_MIN on x86 (cvtsd2si), which always fails
 * the range test of sift.cpp:870-872; model that as an explicit rejection. */
static int round_offset(double v, int *out)
{
    if (!(fabs(v) < 1.0e9))
        return 0;
    *out = so_round(v);
    return 1;
}

/* sift.cpp:838-910 InterploationExtremum. Returns 1 and fills kp on success. */
static int interpolate_extremum(const so_pyramid *p, int o, int index, int x, int y, int interval, so_keypoint *kp)
{
    double off[3] = {0};
    int rows = p->rows[o], cols = p->cols[o];
    int idx = index;
    int intvl = interval;
    int i = 0;
    while (i < SO_MAX_INTERP_STEPS) {
        get_offset_x(p, o, idx, x, y, off);
        if (fabs(off[0]) < 0.5 && fabs(off[1]) < 0.5 && fabs(off[2]) < 0.5)
            break;
        int rx, ry, ri;
        if (!round_offset(off[0], &rx) || !round_offset(off[1], &ry) || !round_offset(off[2], &ri))
            return 0;
        x += rx;
        y += ry;
        interval += ri;
        idx = index - intvl + interval;
        if (interval < 1 || interval > SO_INTERVALS || x >= cols - 1 || x < 2 || y >= rows - 1 || y < 2)
            return 0;
        i++;
    }
    if (i >= SO_MAX_INTERP_STEPS)
        return 0;
    if (get_fabs_dx(p, o, idx, x, y, off) < SO_DXTHRESHOLD / SO_INTERVALS)
        return 0;
    kp->x = x;
    kp->y = y;
    kp->offset_x = off[0];
    kp->offset_y = off[1];
    kp->interval = interval;
    kp->offset_interval = off[2];
    kp->octave = o;
    kp->dx = (x + off[0]) * pow(2.0, o);
    kp->dy = (y + off[1]) * pow(2.0, o);
    return 1;
}

/* sift.cpp:920-987 DetectionLocalExtrema and :989-1056 DetectionLocalExtrema_nointerval. */
size_t so_detect(const so_pyramid *p, int nointerval, so_keypoint **out, size_t count, so_counters *ctr)
{
    int S = p->intervals;
    size_t cap = count ? count : 0;
    so_keypoint *v = *out;
    so_counters c = {0, 0, 0, 0, 0};
    double thresh = 0.5 * SO_DXTHRESHOLD / S;
    for (int o = 0; o < p->octaves; o++) {
        int rows = p->rows[o], cols = p->cols[o];
        for (int i = 1; i < (S + 2) - 1; i++) {
            int index = o * (S + 2) + i;
            for (int y = SO_IMG_BORDER; y < rows - SO_IMG_BORDER; y++) {
                for (int x = SO_IMG_BORDER; x < cols - SO_IMG_BORDER; x++) {
                    c.cc00++;
                    double val = DOG(p, index, o, x, y);
                    if (fabs(val) > thresh) {
                        c.cc0++;
                        if (is_extremum(p, o, index, x, y, nointerval)) {
                            c.cc1++;
                            so_keypoint kp;
                            memset(&kp, 0, sizeof kp);
                            if (interpolate_extremum(p, o, index, x, y, i, &kp)) {
                                c.cc2++;
                                if (pass_edge_response(p, o, index, kp.x, kp.y)) {
                                    kp.val = DOG(p, index, o, kp.x, kp.y);
                                    c.cc3++;
                                    if (count == cap) {
                                        cap = cap ? cap * 2 : 1024;
                                        v = (so_keypoint *)realloc(v, cap * sizeof(so_keypoint));
                                    }
                                    v[count++] = kp;
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    if (ctr)
        *ctr = c;
    *out = v;
    return count;
}

/* ------------------------------------------------------------------ scale, orientation */

/* sift.cpp:1065-1077 CalculateScale */
void so_calculate_scale(so_keypoint *kp, size_t n, double sigma, int intervals)
{
    for (size_t i = 0; i < n; i++) {
        double intvl = kp[i].interval + kp[i].offset_interval;
        kp[i].scale = sigma * pow(2.0, kp[i].octave + intvl / intervals);
        kp[i].octave_scale = sigma * pow(2.0, intvl / intervals);
    }
}

/* sift.cpp:1085-1096 HalfFeatures */
void so_half_features(so_keypoint *kp, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        kp[i].dx /= 2;
        kp[i].dy /= 2;
        kp[i].scale /= 2;
    }
}

/* sift.cpp:1109-1136 CalcGradMagOri: differences without the 1/2 factor; invalid on the border. */
static int calc_grad_mag_ori(const double *g, int rows, int cols, int x, int y, double *mag, double *ori)
{
    if (x > 0 && x < cols - 1 && y > 0 && y < rows - 1) {
        double dx = g[(size_t)y * cols + (x + 1)] - g[(size_t)y * cols + (x - 1)];
        double dy = g[(size_t)(y + 1) * cols + x] - g[(size_t)(y - 1) * cols + x];
        *mag = sqrt(dx * dx + dy * dy);
        *ori = atan2(dy, dx);
        return 1;
    }
    return 0;
}

/* sift.cpp:1149-1199 CalculateOrientationHistogram: i is the x offset (outer), j the y offset. */
void so_orientation_histogram(const double *gauss, int rows, int cols, int x, int y, int radius,
                              double sigma, double *hist)
{
    const int bins = SO_ORI_HIST_BINS;
    for (int i = 0; i < bins; i++)
        hist[i] = 0.0;
    const double PI2 = 2.0 * SO_PI;
    double econs = -1.0 / (2.0 * sigma * sigma);
    for (int i = -radius; i <= radius; i++) {
        for (int j = -radius; j <= radius; j++) {
            double mag, ori;
            if (calc_grad_mag_ori(gauss, rows, cols, x + i, y + j, &mag, &ori)) {
                double weight = exp((i * i + j * j) * econs);
                int bin = so_round(bins * (SO_PI - ori) / PI2);
                bin = bin < bins ? bin : 0;
                hist[bin] += mag * weight;
            }
        }
    }
}

/* sift.cpp:1209-1221 GaussSmoothOriHist: circular [1/4 1/2 1/4], in place. */
static void smooth_ori_hist(double *hist, int n)
{
    double prev = hist[n - 1], temp, h0 = hist[0];
    for (int i = 0; i < n; i++) {
        temp = hist[i];
        hist[i] = 0.25 * prev + 0.5 * hist[i] + 0.25 * (i + 1 >= n ? h0 : hist[i + 1]);
        prev = temp;
    }
}

static so_keypoint *push_kp(so_keypoint *v, size_t *count, size_t *cap, const so_keypoint *k)
{
    if (*count == *cap) {
        *cap = *cap ? *cap * 2 : 1024;
        v = (so_keypoint *)realloc(v, *cap * sizeof(so_keypoint));
    }
    v[(*count)++] = *k;
    return v;
}

/* sift.cpp:1317-1339 OrientationAssignment with :1231-1240 DominantDirection and
 * :1280-1307 CalcOriFeatures (one feature per peak >= 0.8*max, parabola-refined bin). */
size_t so_orientation_assignment(const so_keypoint *extrema, size_t n, const so_pyramid *p,
                                 so_keypoint **features, size_t count)
{
    const int nb = SO_ORI_HIST_BINS;
    size_t cap = count;
    so_keypoint *v = *features;
    const double PI2 = SO_PI * 2.0;
    for (size_t e = 0; e < n; e++) {
        const so_keypoint *kp = &extrema[e];
        int o = kp->octave;
        const double *g = p->gauss[o * (SO_INTERVALS + 3) + kp->interval];
        double hist[SO_ORI_HIST_BINS];
        so_orientation_histogram(g, p->rows[o], p->cols[o], kp->x, kp->y,
                                 so_round(3.0 * SO_ORI_SIGMA_TIMES * kp->octave_scale),
                                 SO_ORI_SIGMA_TIMES * kp->octave_scale, hist);
        for (int j = 0; j < SO_ORI_SMOOTH_TIMES; j++)
            smooth_ori_hist(hist, nb);
        double maxd = hist[0];
        for (int i = 1; i < nb; i++)
            if (hist[i] > maxd)
                maxd = hist[i];
        double mag_thr = maxd * SO_ORI_PEAK_RATIO;
        for (int i = 0; i < nb; i++) {
            int l = (i == 0) ? nb - 1 : i - 1;
            int r = (i + 1) % nb;
            if (hist[i] > hist[l] && hist[i] > hist[r] && hist[i] >= mag_thr) {
                double bin = i + (0.5 * ((hist[l]) - (hist[r])) / ((hist[l]) - 2.0 * (hist[i]) + (hist[r])));
                bin = (bin < 0) ? (bin + nb) : (bin >= nb ? (bin - nb) : bin);
                so_keypoint nk = *kp; /* CopyKeypoint (:1249-1267) leaves descriptor fields unset */
                nk.descr_length = 0;
                memset(nk.descriptor, 0, sizeof nk.descriptor);
                nk.ori = ((PI2 * bin) / nb) - SO_PI;
                v = push_kp(v, &count, &cap, &nk);
            }
        }
    }
    *features = v;
    return count;
}

/* ------------------------------------------------------------------ descriptors */

/* sift.cpp:1352-1438 InterpHistEntry: trilinear spread into hist[r][c][o]. */
static void interp_hist_entry(double *hist /* [4][4][8] */, double xbin, double ybin, double obin, double mag,
                              int bins, int d)
{
    int r0 = so_floor(ybin), c0 = so_floor(xbin), o0 = so_floor(obin);
    double d_r = ybin - r0, d_c = xbin - c0, d_o = obin - o0;
    for (int r = 0; r <= 1; r++) {
        int rb = r0 + r;
        if (rb >= 0 && rb < d) {
            double v_r = mag * ((r == 0) ? 1.0 - d_r : d_r);
            for (int c = 0; c <= 1; c++) {
                int cb = c0 + c;
                if (cb >= 0 && cb < d) {
                    double v_c = v_r * ((c == 0) ? 1.0 - d_c : d_c);
                    for (int o = 0; o <= 1; o++) {
                        int ob = (o0 + o) % bins;
                        double v_o = v_c * ((o == 0) ? 1.0 - d_o : d_o);
                        hist[(rb * d + cb) * bins + ob] += v_o;
                    }
                }
            }
        }
    }
}

/* sift.cpp:1454-1535 CalculateDescrHist: i is the y offset (outer), j the x offset. */
static void calc_descr_hist(const double *g, int rows, int cols, int x, int y, double octave_scale, double ori,
                            double *hist)
{
    const int bins = SO_DESCR_HIST_BINS, width = SO_DESCR_WINDOW_WIDTH;
    for (int i = 0; i < width * width * bins; i++)
        hist[i] = 0.0;
    double cos_ori = cos(ori);
    double sin_ori = sin(ori);
    double sigma = 0.5 * width;
    double conste = -1.0 / (2 * sigma * sigma);
    double PI2 = SO_PI * 2;
    double sub_hist_width = SO_DESCR_SCALE_ADJUST * octave_scale;
    int radius = (int)((sub_hist_width * sqrt(2.0) * (width + 1)) / 2.0);
    for (int i = -radius; i <= radius; i++) {
        for (int j = -radius; j <= radius; j++) {
            double rot_x = (cos_ori * j - sin_ori * i) / sub_hist_width;
            double rot_y = (sin_ori * j + cos_ori * i) / sub_hist_width;
            double xbin = rot_x + width / 2 - 0.5;
            double ybin = rot_y + width / 2 - 0.5;
            if (xbin > -1.0 && xbin < width && ybin > -1.0 && ybin < width) {
                double grad_mag, grad_ori;
                if (calc_grad_mag_ori(g, rows, cols, x + j, y + i, &grad_mag, &grad_ori)) {
                    grad_ori = (SO_PI - grad_ori) - ori;
                    while (grad_ori < 0.0)
                        grad_ori += PI2;
                    while (grad_ori >= PI2)
                        grad_ori -= PI2;
                    double obin = grad_ori * (bins / PI2);
                    double weight = exp(conste * (rot_x * rot_x + rot_y * rot_y));
                    interp_hist_entry(hist, xbin, ybin, obin, grad_mag * weight, bins, width);
                }
            }
        }
    }
}

/* sift.cpp:1542-1557 NormalizeDescr */
static void normalize_descr(double *d, int n)
{
    double len_sq = 0.0;
    for (int i = 0; i < n; i++)
        len_sq += d[i] * d[i];
    double len_inv = 1.0 / sqrt(len_sq);
    for (int i = 0; i < n; i++)
        d[i] *= len_inv;
}

/* sift.cpp:1567-1597 HistToDescriptor: normalise, clamp at 0.2, normalise, min(255,(int)(512*v)). */
void so_descriptor_one(so_keypoint *f, const so_pyramid *p, double *prequant)
{
    int o = f->octave;
    const double *g = p->gauss[o * (SO_INTERVALS + 3) + f->interval];
    double hist[128];
    calc_descr_hist(g, p->rows[o], p->cols[o], f->x, f->y, f->octave_scale, f->ori, hist);
    const int k = 128;
    for (int i = 0; i < k; i++)
        f->descriptor[i] = hist[i];
    f->descr_length = k;
    normalize_descr(f->descriptor, k);
    for (int i = 0; i < k; i++)
        if (f->descriptor[i] > SO_DESCR_MAG_THR)
            f->descriptor[i] = SO_DESCR_MAG_THR;
    normalize_descr(f->descriptor, k);
    if (prequant)
        memcpy(prequant, f->descriptor, sizeof(double) * (size_t)k);
    for (int i = 0; i < k; i++) {
        int int_val = (int)(SO_INT_DESCR_FCTR * f->descriptor[i]);
        f->descriptor[i] = int_val < 255 ? int_val : 255;
    }
}

/* sift.cpp:1607-1631 DescriptorRepresentation */
void so_descriptors(so_keypoint *f, size_t n, const so_pyramid *p)
{
    for (size_t i = 0; i < n; i++)
        so_descriptor_one(&f[i], p, NULL);
}

/* sift.cpp:1640-1643 FeatureCmp + :1703 std::sort: ascending scale. Stable merge sort here:
 * ties keep emission order (octave, interval, y, x, orientation bin). */
void so_sort_features(so_keypoint *f, size_t n)
{
    if (n < 2)
        return;
    so_keypoint *tmp = (so_keypoint *)malloc(n * sizeof(so_keypoint));
    so_keypoint *a = f, *b = tmp;
    for (size_t w = 1; w < n; w *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            size_t mid = lo + w < n ? lo + w : n;
            size_t hi = lo + 2 * w < n ? lo + 2 * w : n;
            size_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                if (a[j].scale < a[i].scale)
                    b[k++] = a[j++];
                else
                    b[k++] = a[i++];
            }
            while (i < mid)
                b[k++] = a[i++];
            while (j < hi)
                b[k++] = a[j++];
        }
        so_keypoint *t = a;
        a = b;
        b = t;
    }
    if (a != f)
        memcpy(f, a, n * sizeof(so_keypoint));
    free(tmp);
}

/* ------------------------------------------------------------------ drivers */

/* sift.cpp:1646-1707 Sift and :1709-1770 Sift_nointerval. */
so_keypoint *so_sift_debug(const uint8_t *src, int rows, int cols, int channels, size_t step,
                           double sigma, int intervals, int nointerval,
                           size_t *n, so_keypoint **keypoints, size_t *n_keypoints,
                           so_pyramid **pyr, so_counters *ctr)
{
    double *base = (double *)malloc(sizeof(double) * (size_t)rows * cols * 4);
    so_create_init_smooth_gray(src, rows, cols, channels, step, sigma, base);
    int octaves = so_num_octaves(rows * 2, cols * 2);
    so_pyramid *p = so_build_pyramid(base, rows * 2, cols * 2, octaves, intervals, sigma);
    free(base);
    build_dog(p);
    so_keypoint *ext = NULL;
    size_t ne = so_detect(p, nointerval, &ext, 0, ctr);
    so_calculate_scale(ext, ne, sigma, intervals);
    so_half_features(ext, ne);
    so_keypoint *feat = NULL;
    size_t nf = so_orientation_assignment(ext, ne, p, &feat, 0);
    so_descriptors(feat, nf, p);
    if (keypoints)
        *keypoints = ext;
    else
        free(ext);
    if (n_keypoints)
        *n_keypoints = ne;
    if (pyr)
        *pyr = p;
    else
        so_free_pyramid(p);
    *n = nf;
    return feat; /* unsorted: emission order */
}

so_keypoint *so_sift(const uint8_t *src, int rows, int cols, int channels, size_t step,
                     double sigma, int intervals, int nointerval,
                     size_t *n, size_t *n_keypoints, so_counters *ctr, so_timings *tm)
{
    so_timings t;
    double t0 = now_s(), t1;
    double *base = (double *)malloc(sizeof(double) * (size_t)rows * cols * 4);
    so_create_init_smooth_gray(src, rows, cols, channels, step, sigma, base);
    t1 = now_s(); t.init = t1 - t0;
    int octaves = so_num_octaves(rows * 2, cols * 2);
    so_pyramid *p = so_build_pyramid(base, rows * 2, cols * 2, octaves, intervals, sigma);
    free(base);
    double t2 = now_s(); t.gauss = t2 - t1;
    build_dog(p);
    double t3 = now_s(); t.dog = t3 - t2;
    so_keypoint *ext = NULL;
    size_t ne = so_detect(p, nointerval, &ext, 0, ctr);
    so_calculate_scale(ext, ne, sigma, intervals);
    so_half_features(ext, ne);
    double t4 = now_s(); t.detect = t4 - t3;
    so_keypoint *feat = NULL;
    size_t nf = so_orientation_assignment(ext, ne, p, &feat, 0);
    double t5 = now_s(); t.orient = t5 - t4;
    so_descriptors(feat, nf, p);
    double t6 = now_s(); t.descr = t6 - t5;
    so_sort_features(feat, nf);
    double t7 = now_s(); t.sort = t7 - t6;
    t.total = t7 - t0;
    free(ext);
    so_free_pyramid(p);
    if (n_keypoints)
        *n_keypoints = ne;
    if (tm)
        *tm = t;
    *n = nf;
    return feat;
}

/* ------------------------------------------------------------------ threaded batch (CPU baseline) */

typedef struct batch_job {
    const uint8_t *imgs;
    int n_images, rows, cols;
    size_t *counts;
    int next;
    pthread_mutex_t mu;
} batch_job;

static void *batch_worker(void *arg)
{
    batch_job *job = (batch_job *)arg;
    for (;;) {
        pthread_mutex_lock(&job->mu);
        int i = job->next++;
        pthread_mutex_unlock(&job->mu);
        if (i >= job->n_images)
            break;
        size_t n = 0;
        const uint8_t *img = job->imgs + (size_t)i * job->rows * job->cols * 3;
        so_keypoint *f = so_sift(img, job->rows, job->cols, 3, (size_t)job->cols * 3, 1.6, SO_INTERVALS, 0,
                                 &n, NULL, NULL, NULL);
        free(f);
        if (job->counts)
            job->counts[i] = n;
    }
    return NULL;
}

double so_sift_batch(const uint8_t *imgs, int n_images, int rows, int cols, int threads, size_t *counts)
{
    if (threads < 1)
        threads = 1;
    if (threads > 256)
        threads = 256;
    batch_job job = {imgs, n_images, rows, cols, counts, 0, PTHREAD_MUTEX_INITIALIZER};
    pthread_t th[256];
    double t0 = now_s();
    for (int i = 0; i < threads; i++)
        pthread_create(&th[i], NULL, batch_worker, &job);
    for (int i = 0; i < threads; i++)
        pthread_join(th[i], NULL);
    return now_s() - t0;
}

/* ------------------------------------------------------------------ text format */

/* sift.cpp:1910-1931 write_features: "n 128", then per feature "dy dx scale ori", a line break
 * before every 20th descriptor value, default ostream formatting (== %g). */
int so_write_features(const so_keypoint *f, size_t n, const char *path)
{
    FILE *fp = fopen(path, "w");
    if (!fp)
        return -1;
    fprintf(fp, "%zu %d\n", n, 128);
    for (size_t i = 0; i < n; i++) {
        fprintf(fp, "%g %g %g %g\n", f[i].dy, f[i].dx, f[i].scale, f[i].ori);
        for (int j = 0; j < 128; j++) {
            if (j % 20 == 0)
                fputc('\n', fp);
            fprintf(fp, "%g ", f[i].descriptor[j]);
        }
        fputc('\n', fp);
        fputc('\n', fp);
    }
    fclose(fp);
    return 0;
}

/* ------------------------------------------------------------------ synthetic input */

/* ---- descriptor matching (demo main, sift/源.cpp:88-156) ---- */
size_t so_match(const so_keypoint *f1, size_t n1, const so_keypoint *f2, size_t n2, double ratio, int accumulate_bug,
                int *index_1, int *index_2, double *best2)
{
    size_t nm = 0;
    for (size_t i = 0; i < n1; i++) {
        double odest = 0; /* 源.cpp:97 */
        int min_index = 0;
        double min_value = DBL_MAX, pen_value = DBL_MAX; /* 源.cpp:99-102 */
        double min_sq = DBL_MAX, pen_sq = DBL_MAX;
        for (size_t j = 0; j < n2; j++) {
            if (!accumulate_bug)
                odest = 0;
            double sq = 0;
            for (int k = 0; k < 128; k++) { /* 源.cpp:106-107 */
                double d = f1[i].descriptor[k] - f2[j].descriptor[k];
                odest += d * d;
                sq += d * d;
            }
            odest = sqrt(odest); /* 源.cpp:110 */
            if (odest < min_value) { /* 源.cpp:114-129 */
                pen_value = min_value;
                min_value = odest;
                min_index = (int)j;
            } else if (odest < pen_value) {
                pen_value = odest;
            }
            if (sq < min_sq) {
                pen_sq = min_sq;
                min_sq = sq;
            } else if (sq < pen_sq) {
                pen_sq = sq;
            }
        }
        if (best2) {
            best2[2 * i] = min_sq;
            best2[2 * i + 1] = pen_sq;
        }
        if (min_value / pen_value < ratio) { /* 源.cpp:149-155 */
            index_1[nm] = (int)i;
            index_2[nm] = min_index;
            nm++;
        }
    }
    return nm;
}

static uint64_t sm64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static double lat(int ix, int iy, int k, uint64_t seed)
{
    uint64_t c = (uint64_t)(uint32_t)ix | ((uint64_t)(uint32_t)iy << 32);
    uint64_t h = sm64(seed ^ sm64(c) ^ sm64(0x1000 + (uint64_t)k));
    return ((double)(h >> 11) * 0x1.0p-53) * 2 - 1;
}

static double vnoise(double x, double y, int k, uint64_t seed)
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

/* 5-octave value noise: lattice periods 4..64 px, amplitudes 20,28,36,40,40 around 128. */
void so_vnoise5(uint8_t *dst, int rows, int cols, uint64_t seed)
{
    static const double L[5] = {4, 8, 16, 32, 64};
    static const double A[5] = {20, 28, 36, 40, 40};
    for (int y = 0; y < rows; y++) {
        for (int x = 0; x < cols; x++) {
            double v = 128;
            for (int k = 0; k < 5; k++)
                v += A[k] * vnoise(x / L[k], y / L[k], k, seed);
            long q = lrint(v);
            if (q < 0)
                q = 0;
            if (q > 255)
                q = 255;
            uint8_t *p = dst + ((size_t)y * cols + x) * 3;
            p[0] = p[1] = p[2] = (uint8_t)q;
        }
    }
}
