// kernels_describe.hip -- orientation assignment, 4x4x8 descriptors, scale ordering and output packing
// for gfx950.  All histogram arithmetic is double precision whatever the pyramid's storage type.
#include "sg_internal.h"

namespace sg {

#define SG_PI 3.1415926535897932384626433832795 /* CV_PI */

// CalcGradMagOri (reference sift/sift.cpp:1109-1136): differences without the 1/2 factor.
template <typename T>
__device__ __forceinline__ bool grad_at(const T *g, int rows, int cols, int pitch, int x, int y, double &dx, double &dy)
{
    if (x > 0 && x < cols - 1 && y > 0 && y < rows - 1) {
        const T *p = g + (size_t)y * pitch + x;
        dx = (double)p[1] - (double)p[-1];
        dy = (double)p[pitch] - (double)p[-pitch];
        return true;
    }
    return false;
}

// ---- orientation ---------------------------------------------------------------------------------------
// One wave per keypoint.  CalculateOrientationHistogram (sift.cpp:1149-1199): every lane accumulates its
// samples into a lane-private 36-bin column of an LDS table (no atomics), lane b then adds the 64 partial
// sums of bin b in lane order, so the histogram is reproducible run to run.  Two passes of the circular
// [1/4 1/2 1/4] filter (sift.cpp:1209-1221), the maximum (sift.cpp:1231-1240) and the peak test with
// parabolic refinement (sift.cpp:1280-1307) run on lanes 0..35.
constexpr int kOriStride = 65; // doubles per bin row: 64 lanes + 1 pad (conflict-free column sums)
constexpr int kOriWaveDoubles = kOriBins * kOriStride + 64;

template <typename T>
__global__ __launch_bounds__(256) void orient_k(OctaveTab tab, const KpRec *kp, size_t kp_img,
                                                const unsigned int *kp_src, size_t src_img,
                                                const unsigned int *kp_count, double *ori_tmp, size_t ori_img,
                                                unsigned long long *peak_words, size_t pw_img,
                                                unsigned int *chunk_sums, size_t chunks_img, unsigned int cap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *hl = reinterpret_cast<double *>(smem_raw) + (size_t)wave * kOriWaveDoubles;
    double *hs = hl + kOriBins * kOriStride;
    const int b = blockIdx.y;
    const unsigned int n = min(kp_count[b], cap);
    const double PI2 = 2.0 * SG_PI;

    for (unsigned int j0 = blockIdx.x * 4u; j0 < n; j0 += gridDim.x * 4u) {
        const unsigned int j = j0 + wave;
        const bool active = j < n;
        if (active) {
            const KpRec r = kp[(size_t)b * kp_img + kp_src[(size_t)b * src_img + j]];
            const int o = r.octave;
            const T *g = (const T *)tab.lv[o][r.interval] + (size_t)b * tab.img_stride[o];
            const int rows = tab.rows[o], cols = tab.cols[o], pitch = tab.pitch[o];
            const int radius = (int)rint(3.0 * 1.5 * r.oscale); // ORI_WINDOW_RADIUS * octave_scale
            const double sig = 1.5 * r.oscale;                   // ORI_SIGMA_TIMES * octave_scale
            const double econs = -1.0 / (2.0 * sig * sig);
#pragma unroll 4
            for (int q = 0; q < kOriBins; q++)
                hl[q * kOriStride + lane] = 0.0;
            const int W = 2 * radius + 1, total = W * W;
            for (int s = lane; s < total; s += 64) {
                int i = s / W - radius;  // x offset (outer loop of the reference)
                int jj = s % W - radius; // y offset
                double dx, dy;
                if (grad_at(g, rows, cols, pitch, r.x + i, r.y + jj, dx, dy)) {
                    double mag = sqrt(dx * dx + dy * dy);
                    double ori = atan2(dy, dx);
                    double weight = exp((i * i + jj * jj) * econs);
                    int bin = (int)rint(kOriBins * (SG_PI - ori) / PI2);
                    bin = bin < kOriBins ? bin : 0;
                    hl[bin * kOriStride + lane] += mag * weight;
                }
            }
        }
        __syncthreads();
        if (active && lane < kOriBins) {
            double acc = 0.0;
            for (int l = 0; l < 64; l++)
                acc += hl[lane * kOriStride + l];
            hs[lane] = acc;
        }
        __syncthreads();
        double c = 0.0;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) { // ORI_SMOOTH_TIMES
            if (active && lane < kOriBins) {
                double prev = hs[lane == 0 ? kOriBins - 1 : lane - 1];
                double next = hs[lane + 1 >= kOriBins ? 0 : lane + 1];
                c = 0.25 * prev + 0.5 * hs[lane] + 0.25 * next;
            }
            __syncthreads();
            if (active && lane < kOriBins)
                hs[lane] = c;
            __syncthreads();
        }
        double mx = (active && lane < kOriBins) ? c : -1.0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            double t = __shfl_xor(mx, d);
            mx = t > mx ? t : mx;
        }
        bool peak = false;
        if (active && lane < kOriBins) {
            double l = hs[lane == 0 ? kOriBins - 1 : lane - 1];
            double rr = hs[(lane + 1) % kOriBins];
            double thr = mx * 0.8; // ORI_PEAK_RATIO
            if (c > l && c > rr && c >= thr) {
                peak = true;
                double bin = lane + (0.5 * (l - rr) / (l - 2.0 * c + rr));
                bin = (bin < 0) ? (bin + kOriBins) : (bin >= kOriBins ? (bin - kOriBins) : bin);
                ori_tmp[(size_t)b * ori_img + (size_t)j * kOriBins + lane] = ((PI2 * bin) / kOriBins) - SG_PI;
            }
        }
        unsigned long long m = __ballot(peak);
        if (active && lane == 0) {
            peak_words[(size_t)b * pw_img + j] = m;
            if (m)
                atomicAdd(&chunk_sums[(size_t)b * chunks_img + j / kChunkWords], (unsigned int)__popcll(m));
        }
        __syncthreads();
    }
}

void launch_orient(const OctaveTab &tab, const KpRec *kp, size_t kp_img_stride, const unsigned int *kp_src,
                   size_t src_img_stride, const unsigned int *kp_count, double *ori_tmp, size_t ori_img_stride,
                   unsigned long long *peak_words, size_t pw_img_stride, unsigned int *chunk_sums,
                   size_t chunks_img_stride, unsigned int cap, int B, int prec, hipStream_t s)
{
    unsigned int blocks = (cap + 3) / 4;
    if (blocks > 2048)
        blocks = 2048;
    dim3 g(blocks, B);
    size_t lds = sizeof(double) * 4 * kOriWaveDoubles;
    if (prec == SG_F64)
        orient_k<double><<<g, 256, lds, s>>>(tab, kp, kp_img_stride, kp_src, src_img_stride, kp_count, ori_tmp,
                                             ori_img_stride, peak_words, pw_img_stride, chunk_sums, chunks_img_stride, cap);
    else
        orient_k<float><<<g, 256, lds, s>>>(tab, kp, kp_img_stride, kp_src, src_img_stride, kp_count, ori_tmp,
                                            ori_img_stride, peak_words, pw_img_stride, chunk_sums, chunks_img_stride, cap);
}

// ---- descriptors -----------------------------------------------------------------------------------------
// One 256-thread workgroup per feature.  CalculateDescrHist + InterpHistEntry (sift.cpp:1352-1535): the
// threads walk the (2r+1)^2 window, accepted samples spread trilinearly into a wave-private 4x4x8 histogram
// with LDS double atomics; the four wave histograms are added in wave order.  HistToDescriptor
// (sift.cpp:1567-1597): normalise, clamp at 0.2, normalise, min(255,(int)(512 v)).

__device__ __forceinline__ void lds_add(double *p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ double block_sum_128(double v, double *red, int tid)
{
    // v is meaningful on threads 0..127 (waves 0 and 1); result broadcast to every thread
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += __shfl_xor(v, d);
    __syncthreads();
    if ((tid & 63) == 0)
        red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1];
}

template <typename T>
__global__ __launch_bounds__(256) void descr_k(OctaveTab tab, const KpRec *kp, size_t kp_img,
                                               const unsigned int *kp_src, size_t src_img,
                                               const unsigned int *feat_src, size_t fsrc_img,
                                               const unsigned int *feat_count, const double *ori_tmp, size_t ori_img,
                                               uint8_t *desc, size_t desc_img, double *prequant, size_t pq_img,
                                               double *fscale, size_t fs_img, unsigned int cap)
{
    __shared__ double hist[4][128];
    __shared__ double red[4];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int b = blockIdx.y;
    const unsigned int n = min(feat_count[b], cap);
    const double PI2 = SG_PI * 2;

    for (unsigned int f = blockIdx.x; f < n; f += gridDim.x) {
        const unsigned int fs = feat_src[(size_t)b * fsrc_img + f];
        const unsigned int j = fs >> 6, pbin = fs & 63;
        const KpRec r = kp[(size_t)b * kp_img + kp_src[(size_t)b * src_img + j]];
        const double ori = ori_tmp[(size_t)b * ori_img + (size_t)j * kOriBins + pbin];
        const int o = r.octave;
        const T *g = (const T *)tab.lv[o][r.interval] + (size_t)b * tab.img_stride[o];
        const int rows = tab.rows[o], cols = tab.cols[o], pitch = tab.pitch[o];

        double *hflat = &hist[0][0]; // 512 doubles, 2 per thread
        hflat[tid] = 0.0;
        hflat[tid + 256] = 0.0;
        __syncthreads();

        const double cos_ori = cos(ori), sin_ori = sin(ori);
        const double conste = -1.0 / (2 * (0.5 * 4) * (0.5 * 4));
        const double shw = 3 * r.oscale;                           // DESCR_SCALE_ADJUST * octave_scale
        const int radius = (int)((shw * sqrt(2.0) * (4 + 1)) / 2.0); // truncating cast (sift.cpp:1488)
        const int W = 2 * radius + 1, total = W * W;
        double *h = hist[wave];
        for (int s = tid; s < total; s += 256) {
            const int i = s / W - radius;  // y offset (outer loop of the reference)
            const int jj = s % W - radius; // x offset
            double rot_x = (cos_ori * jj - sin_ori * i) / shw;
            double rot_y = (sin_ori * jj + cos_ori * i) / shw;
            double xbin = rot_x + 4 / 2 - 0.5;
            double ybin = rot_y + 4 / 2 - 0.5;
            if (xbin > -1.0 && xbin < 4 && ybin > -1.0 && ybin < 4) {
                double dx, dy;
                if (grad_at(g, rows, cols, pitch, r.x + jj, r.y + i, dx, dy)) {
                    double grad_mag = sqrt(dx * dx + dy * dy);
                    double grad_ori = (SG_PI - atan2(dy, dx)) - ori;
                    while (grad_ori < 0.0)
                        grad_ori += PI2;
                    while (grad_ori >= PI2)
                        grad_ori -= PI2;
                    double obin = grad_ori * (8 / PI2);
                    double weight = exp(conste * (rot_x * rot_x + rot_y * rot_y));
                    double mag = grad_mag * weight;
                    // InterpHistEntry
                    int r0 = (int)floor(ybin), c0 = (int)floor(xbin), o0 = (int)floor(obin);
                    double d_r = ybin - r0, d_c = xbin - c0, d_o = obin - o0;
#pragma unroll
                    for (int rr = 0; rr <= 1; rr++) {
                        int rb = r0 + rr;
                        if (rb >= 0 && rb < 4) {
                            double v_r = mag * ((rr == 0) ? 1.0 - d_r : d_r);
#pragma unroll
                            for (int cc = 0; cc <= 1; cc++) {
                                int cb = c0 + cc;
                                if (cb >= 0 && cb < 4) {
                                    double v_c = v_r * ((cc == 0) ? 1.0 - d_c : d_c);
#pragma unroll
                                    for (int oo = 0; oo <= 1; oo++) {
                                        int ob = (o0 + oo) % 8;
                                        double v_o = v_c * ((oo == 0) ? 1.0 - d_o : d_o);
                                        lds_add(&h[(rb * 4 + cb) * 8 + ob], v_o);
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        double v = 0.0;
        if (tid < 128)
            v = ((hist[0][tid] + hist[1][tid]) + hist[2][tid]) + hist[3][tid];
        double len_sq = block_sum_128(tid < 128 ? v * v : 0.0, red, tid);
        double len_inv = 1.0 / sqrt(len_sq);
        v *= len_inv;
        if (v > 0.2) // DESCR_MAG_THR
            v = 0.2;
        len_sq = block_sum_128(tid < 128 ? v * v : 0.0, red, tid);
        len_inv = 1.0 / sqrt(len_sq);
        v *= len_inv;
        if (tid < 128) {
            if (prequant)
                prequant[(size_t)b * pq_img + (size_t)f * 128 + tid] = v;
            int q = 0;
            if (v == v) { // an all-zero window gives 0/0; keep such a descriptor at zero
                q = (int)(512.0 * v); // INT_DESCR_FCTR, truncating
                q = q < 255 ? q : 255;
            }
            desc[(size_t)b * desc_img + (size_t)f * 128 + tid] = (uint8_t)q;
        }
        if (tid == 0)
            fscale[(size_t)b * fs_img + f] = r.scale;
        __syncthreads();
    }
}

void launch_descr(const OctaveTab &tab, const KpRec *kp, size_t kp_img_stride, const unsigned int *kp_src,
                  size_t src_img_stride, const unsigned int *feat_src, size_t fsrc_img_stride,
                  const unsigned int *feat_count, const double *ori_tmp, size_t ori_img_stride, uint8_t *desc,
                  size_t desc_img_stride, double *prequant, size_t pq_img_stride, double *fscale,
                  size_t fs_img_stride, unsigned int cap, int B, int prec, hipStream_t s)
{
    unsigned int blocks = cap < 4096 ? cap : 4096;
    if (blocks == 0)
        return;
    dim3 g(blocks, B);
    if (prec == SG_F64)
        descr_k<double><<<g, 256, 0, s>>>(tab, kp, kp_img_stride, kp_src, src_img_stride, feat_src, fsrc_img_stride,
                                          feat_count, ori_tmp, ori_img_stride, desc, desc_img_stride, prequant,
                                          pq_img_stride, fscale, fs_img_stride, cap);
    else
        descr_k<float><<<g, 256, 0, s>>>(tab, kp, kp_img_stride, kp_src, src_img_stride, feat_src, fsrc_img_stride,
                                         feat_count, ori_tmp, ori_img_stride, desc, desc_img_stride, prequant,
                                         pq_img_stride, fscale, fs_img_stride, cap);
}

// ---- ordering: stable rank by scale -----------------------------------------------------------------------
// rank(i) = #{ j : scale_j < scale_i  or  (scale_j == scale_i and j < i) } -- the position of feature i in a
// stable ascending sort by scale (reference sift/sift.cpp:1640-1643,1703 sorts by scale alone).
__global__ __launch_bounds__(256) void rank_k(const double *fscale, size_t fs_img, const unsigned int *feat_count,
                                              unsigned int *rank, size_t rank_img, unsigned int cap)
{
    __shared__ double sc[256];
    const int b = blockIdx.y, tid = threadIdx.x;
    const unsigned int n = min(feat_count[b], cap);
    if (blockIdx.x * 256u >= n)
        return;
    const double *fsb = fscale + (size_t)b * fs_img;
    const unsigned int i = blockIdx.x * 256u + tid;
    const double my = i < n ? fsb[i] : 0.0;
    unsigned int cnt = 0;
    for (unsigned int base = 0; base < n; base += 256) {
        unsigned int j = base + tid;
        sc[tid] = j < n ? fsb[j] : 0.0;
        __syncthreads();
        unsigned int m = min(256u, n - base);
        for (unsigned int q = 0; q < m; q++) {
            double s = sc[q];
            cnt += (s < my || (s == my && base + q < i)) ? 1u : 0u;
        }
        __syncthreads();
    }
    if (i < n)
        rank[(size_t)b * rank_img + i] = cnt;
}

void launch_rank(const double *fscale, size_t fs_img_stride, const unsigned int *feat_count, unsigned int *rank,
                 size_t rank_img_stride, unsigned int cap, int B, hipStream_t s)
{
    if (cap == 0)
        return;
    dim3 g((cap + 255) / 256, B);
    rank_k<<<g, 256, 0, s>>>(fscale, fs_img_stride, feat_count, rank, rank_img_stride, cap);
}

// ---- output packing ------------------------------------------------------------------------------------------
// One wave per feature; position = rank (sorted) or emission index.
template <int MODE>
__global__ __launch_bounds__(256) void pack_k(const KpRec *kp, size_t kp_img, const unsigned int *kp_src, size_t src_img,
                                              const unsigned int *feat_src, size_t fsrc_img,
                                              const unsigned int *feat_count, const double *ori_tmp, size_t ori_img,
                                              const uint8_t *desc, size_t desc_img, const unsigned int *rank,
                                              size_t rank_img, void *out, size_t out_img, unsigned int out_cap,
                                              unsigned int *out_counts, unsigned int *overflow, int sorted,
                                              unsigned int cap)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const unsigned int n = min(feat_count[b], cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (out_counts)
            out_counts[b] = n < out_cap ? n : out_cap;
        if (n > out_cap)
            atomicMax(overflow, n);
    }
    for (unsigned int f = blockIdx.x * 4u + (threadIdx.x >> 6); f < n; f += gridDim.x * 4u) {
        const unsigned int pos = sorted ? rank[(size_t)b * rank_img + f] : f;
        if (pos >= out_cap)
            continue;
        const unsigned int fs = feat_src[(size_t)b * fsrc_img + f];
        const unsigned int j = fs >> 6, pbin = fs & 63;
        const KpRec r = kp[(size_t)b * kp_img + kp_src[(size_t)b * src_img + j]];
        const double ori = ori_tmp[(size_t)b * ori_img + (size_t)j * kOriBins + pbin];
        const uint8_t *d = desc + (size_t)b * desc_img + (size_t)f * 128;
        if (MODE == 0) {
            sg_packed_feature *o = reinterpret_cast<sg_packed_feature *>(out) + (size_t)b * out_img + pos;
            if (lane < 32)
                reinterpret_cast<uint32_t *>(o->desc)[lane] = reinterpret_cast<const uint32_t *>(d)[lane];
            else if (lane == 32) {
                o->x = (float)r.dx; o->y = (float)r.dy; o->scale = (float)r.scale; o->ori = (float)ori;
            }
        } else {
            sg_keypoint *o = reinterpret_cast<sg_keypoint *>(out) + (size_t)b * out_img + pos;
            o->descriptor[lane] = (double)d[lane];
            o->descriptor[lane + 64] = (double)d[lane + 64];
            if (lane == 0) {
                o->octave = r.octave; o->interval = r.interval; o->offset_interval = r.offi;
                o->x = r.x; o->y = r.y; o->scale = r.scale; o->dx = r.dx; o->dy = r.dy;
                o->offset_x = r.offx; o->offset_y = r.offy; o->octave_scale = r.oscale; o->ori = ori;
                // descr_length and the 4 padding bytes behind it in one store, so that records are byte-reproducible
                *reinterpret_cast<unsigned long long *>(&o->descr_length) = 128ull; o->val = r.val;
            }
        }
    }
}

void launch_pack(const KpRec *kp, size_t kp_img_stride, const unsigned int *kp_src, size_t src_img_stride,
                 const unsigned int *feat_src, size_t fsrc_img_stride, const unsigned int *feat_count,
                 const double *ori_tmp, size_t ori_img_stride, const uint8_t *desc, size_t desc_img_stride,
                 const unsigned int *rank, size_t rank_img_stride, void *out, size_t out_img_stride,
                 unsigned int out_cap, unsigned int *out_counts, unsigned int *overflow, int mode, int sorted,
                 unsigned int cap, int B, hipStream_t s)
{
    unsigned int blocks = (cap + 3) / 4;
    if (blocks > 2048)
        blocks = 2048;
    if (blocks == 0)
        blocks = 1;
    dim3 g(blocks, B);
    if (mode == 0)
        pack_k<0><<<g, 256, 0, s>>>(kp, kp_img_stride, kp_src, src_img_stride, feat_src, fsrc_img_stride, feat_count,
                                    ori_tmp, ori_img_stride, desc, desc_img_stride, rank, rank_img_stride, out,
                                    out_img_stride, out_cap, out_counts, overflow, sorted, cap);
    else
        pack_k<1><<<g, 256, 0, s>>>(kp, kp_img_stride, kp_src, src_img_stride, feat_src, fsrc_img_stride, feat_count,
                                    ori_tmp, ori_img_stride, desc, desc_img_stride, rank, rank_img_stride, out,
                                    out_img_stride, out_cap, out_counts, overflow, sorted, cap);
}

} // namespace sg
