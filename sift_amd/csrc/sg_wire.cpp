// sg_wire.cpp -- the CPU step from wire records to reference-layout records (sg_expand_wire of include/siftgpu.h).
// Plain C++ with no HIP in it, so that it can also be built on its own (the host sanitizer builds link this file).
#include <stddef.h>
#include <string.h>

#include "siftgpu.h"

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
