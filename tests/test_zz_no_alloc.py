"""The allocation-free matcher (sg_match_packed_device_noalloc) against sg_match_packed_device, through the C ABI:
two calls in a row on one context, the second with smaller inputs, so that what the first call left in the
static scratch array must not leak into the second result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _packed_sets(ctx, sizes, cap):
    """Packed features of vnoise5 frames of the given (w, h, seed), left on the device: [(d_ptr, n)]."""
    import sift_amd
    out = []
    for w, h, seed in sizes:
        d_img = ctx.malloc(w * h * 3)
        d_out = ctx.malloc(cap * 144)
        d_cnt = ctx.malloc(4)
        ctx.synth_vnoise5(d_img, w, h, seed)
        ctx.extract_batch_device(d_img, 1, w, h, 3, d_out, cap, d_cnt)
        ctx.sync()
        n = np.zeros(1, dtype=np.uint32)
        ctx.d2h(n, d_cnt)
        ctx.free(d_img)
        ctx.free(d_cnt)
        out.append((d_out, int(n[0])))
    return out


def _run(ctx, fn, f1, n1, f2, n2, ratio=0.8):
    d_idx = ctx.malloc(max(n1, 1) * 4)
    d_dst = ctx.malloc(max(n1, 1) * 8)
    try:
        fn(f1, n1, f2, n2, d_idx, d_dst, ratio)
        ctx.sync()
        idx = np.zeros(n1, dtype=np.int32)
        dst = np.zeros(2 * n1, dtype=np.uint32)
        if n1:
            ctx.d2h(idx, d_idx)
            ctx.d2h(dst, d_dst)
        return idx, dst
    finally:
        ctx.free(d_idx)
        ctx.free(d_dst)


def test_noalloc_matcher_equals_allocating_matcher():
    import sift_amd
    cap = 8192
    with sift_amd.SiftGPU(max_width=640, max_height=480) as ctx:
        sets = _packed_sets(ctx, [(640, 480, 11), (640, 480, 12), (200, 150, 13), (160, 120, 14)], cap)
        try:
            (a, na), (b, nb), (c, nc), (d, nd) = sets
            assert na > 2000 and nb > 2000 and 50 < nc < na and 50 < nd < nb
            # first call: two 640x480 sets; second call: smaller sets (fewer rows, fewer slices, other padding)
            for f1, n1, f2, n2 in ((a, na, b, nb), (c, nc, d, nd), (c, nc, b, nb), (a, 33, d, 1), (a, 5, d, 0)):
                want_idx, want_dst = _run(ctx, ctx.match_packed_device, f1, n1, f2, n2)
                got_idx, got_dst = _run(ctx, ctx.match_packed_device_noalloc, f1, n1, f2, n2)
                assert np.array_equal(got_idx, want_idx), (n1, n2)
                assert np.array_equal(got_dst, want_dst), (n1, n2)
            assert (want_idx == -1).all()  # nothing to match against in the last case
            # a set matched against itself finds itself at distance 0
            idx, dst = _run(ctx, ctx.match_packed_device_noalloc, c, nc, c, nc, ratio=0.9)
            assert (dst[0::2] == 0).all()
            # sizes beyond the static array are refused, not truncated
            with pytest.raises(sift_amd.SiftGpuError) as e:
                ctx.match_packed_device_noalloc(a, 32769, b, nb, a, None)
            assert e.value.status == sift_amd.api.SG_E_CAPACITY
        finally:
            for p, _ in sets:
                ctx.free(p)


def test_noalloc_matcher_allocates_nothing_on_a_fresh_context():
    """The property the entry point exists for: on a context that has never run the allocating matcher (so that none
    of its grow-on-demand buffers exists), the allocation-free matcher leaves the device's free memory unchanged --
    outputs allocated beforehand, free memory read before and after -- and its result equals the allocating matcher's,
    run afterwards."""
    import ctypes as C
    import sift_amd
    cap = 8192
    with sift_amd.SiftGPU(max_width=640, max_height=480) as ctx:
        # hipMemGetInfo of the HIP runtime the library itself is linked to (a symbol lookup on the library's handle
        # searches its dependencies; another copy of the runtime may be loaded in this process beside it)
        mem_get_info = ctx.lib.hipMemGetInfo

        def free_bytes():
            free, total = C.c_size_t(0), C.c_size_t(0)
            assert mem_get_info(C.byref(free), C.byref(total)) == 0
            return free.value

        sets = _packed_sets(ctx, [(640, 480, 21), (320, 240, 22)], cap)
        (a, na), (b, nb) = sets
        d_idx, d_dst = ctx.malloc(na * 4), ctx.malloc(na * 8)
        try:
            ctx.sync()
            free_before = free_bytes()
            assert ctx.lib.sg_match_scratch_bytes(ctx.ctx) == 0  # the allocating matcher has not run on this context
            for _ in range(2):
                ctx.match_packed_device_noalloc(a, na, b, nb, d_idx, d_dst, 0.8)
                ctx.sync()
            free_after = free_bytes()
            assert free_after == free_before, (free_before, free_after)  # (the runtime's pool hides small requests)
            assert ctx.lib.sg_match_scratch_bytes(ctx.ctx) == 0           # ... the context's own books do not
            got_idx, got_dst = np.zeros(na, dtype=np.int32), np.zeros(2 * na, dtype=np.uint32)
            ctx.d2h(got_idx, d_idx)
            ctx.d2h(got_dst, d_dst)
            # the allocating form on the same context allocates its scratch now (free memory drops) and agrees
            want_idx, want_dst = _run(ctx, ctx.match_packed_device, a, na, b, nb)
            assert ctx.lib.sg_match_scratch_bytes(ctx.ctx) >= (na + nb) * 132
            assert np.array_equal(got_idx, want_idx) and np.array_equal(got_dst, want_dst)
        finally:
            ctx.free(d_idx)
            ctx.free(d_dst)
            for p, _ in sets:
                ctx.free(p)
