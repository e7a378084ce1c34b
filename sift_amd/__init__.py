"""sift_amd -- MI355X (gfx950) SIFT extraction path behind the entry points of Jia-Baos/SIFT's sift/sift.h.

The product is the HIP library sift_amd/libsiftgpu.so (C ABI in include/siftgpu.h); this package is
its Python binding plus small helpers (synthetic input, parity metrics).  There is no CPU fallback.
"""
from .api import (KEYPOINT_DTYPE, PACKED_DTYPE, SG_F32, SG_F64, SG_F64_FMA, STAGES, WIRE_DTYPE, SiftGPU, SiftGpuError,
                  load_library)  # noqa: F401
from . import synth  # noqa: F401
from .synth import vnoise5  # noqa: F401
