// sg_wire.cpp -- the CPU step from wire records to reference-layout records (sg_expand_wire of include/siftgpu.h).
// Plain C++ with no HIP in it, so that it can also be built on its own (the host sanitizer builds link this file).
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <immintrin.h>

#include "siftgpu.h"

// Copy into a page-locked staging buffer that a DMA engine reads next and the CPU never reads again: streaming
// (non-temporal) stores, which write whole lines without first loading them into the cache.  dst must be 32-byte
// aligned (len need not be a multiple of anything); plain memcpy where the CPU has no AVX2.
namespace {
__attribute__((target("avx2"))) void copy_streaming_avx2(unsigned char *dst, const unsigned char *src, size_t len)
{
    size_t i = 0;
    for (; i + 128 <= len; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 96), d);
    }
    if (i < len)
        memcpy(dst + i, src + i, len - i);
    _mm_sfence(); // the streamed lines are globally visible before the transfer that reads them is enqueued
}
} // namespace

namespace sg {
void copy_to_staging(void *dst, const void *src, size_t len)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && (reinterpret_cast<uintptr_t>(dst) & 31) == 0 && len >= 4096)
        copy_streaming_avx2(static_cast<unsigned char *>(dst), static_cast<const unsigned char *>(src), len);
    else
        memcpy(dst, src, len);
}
} // namespace sg

extern "C" void sg_expand_wire(const sg_wire_feature *in, size_t n, sg_keypoint *out)
{
    if (!in || !out)
        return;
    for (size_t i = 0; i < n; i++) {
        const sg_wire_feature &r = in[i];
        sg_keypoint &k = out[i];
        memset(&k, 0, offsetof(sg_keypoint, descriptor)); // padding bytes too: records are byte-reproducible
        k.octave = r.octave;
        k.interval = r.interval;
        k.offset_interval = r.offset_interval;
        k.x = r.x;
        k.y = r.y;
        k.scale = r.scale;
        k.dx = r.dx;
        k.dy = r.dy;
        k.offset_x = r.offset_x;
        k.offset_y = r.offset_y;
        k.octave_scale = r.octave_scale;
        k.ori = r.ori;
        k.descr_length = 128;
        for (int q = 0; q < 128; q++)
            k.descriptor[q] = (double)r.desc[q];
        k.val = r.val;
    }
}
