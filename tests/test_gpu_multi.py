"""The multi-GPU path with real extraction output: frames sharded over ranks, each rank extracts its shard on the GPU,
the feature sets are all-gathered, and the gathered set must be byte-equal to one rank extracting all frames.
Two ranks share the one GPU of the test box, so the collective runs over gloo there (RCCL refuses two ranks on one
device); the C ABI's RCCL path is exercised with a communicator of one rank."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, CAP = 320, 240, 4096
N_FRAMES = 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _extract_shard(ctx, first, n):
    """Frames first .. first+n-1 (vnoise5 seeds 1000 + frame) on the device; returns (d_out, d_cnt) device pointers."""
    fb = W * H * 3
    d_img, d_out, d_cnt = ctx.malloc(n * fb), ctx.malloc(n * CAP * 144), ctx.malloc(4 * n)
    for i in range(n):
        ctx.synth_vnoise5(d_img + i * fb, W, H, 1000 + first + i)
    ctx.extract_batch_device(d_img, n, W, H, 3, d_out, CAP, d_cnt)
    ctx.sync()
    ctx.free(d_img)
    return d_out, d_cnt


def _single_rank_reference():
    import sift_amd
    with sift_amd.SiftGPU(max_width=W, max_height=H, max_batch=N_FRAMES) as ctx:
        d_out, d_cnt = _extract_shard(ctx, 0, N_FRAMES)
        cnt = np.zeros(N_FRAMES, dtype=np.uint32)
        ctx.d2h(cnt, d_cnt)
        out = np.zeros((N_FRAMES, CAP), dtype=sift_amd.PACKED_DTYPE)
        ctx.d2h(out, d_out)
        ctx.free(d_out)
        ctx.free(d_cnt)
    return [out[i, : cnt[i]].copy() for i in range(N_FRAMES)]


def _worker(rank, world, port, result_dir):
    import torch
    import torch.distributed as dist
    import sift_amd
    from sift_amd import dist as sdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, stop = sdist.shard_frames(N_FRAMES, world, rank)
        n = stop - first
        with sift_amd.SiftGPU(max_width=W, max_height=H, max_batch=n) as ctx:
            d_out, d_cnt = _extract_shard(ctx, first, n)
            cnt = np.zeros(n, dtype=np.uint32)
            ctx.d2h(cnt, d_cnt)
            out = np.zeros(n * CAP * 144, dtype=np.uint8)
            ctx.d2h(out, d_out)
            ctx.free(d_out)
            ctx.free(d_cnt)
        gv, gcv, rank_first = sdist.allgather_features_v(torch.from_numpy(out), torch.from_numpy(cnt.astype(np.int32)), CAP)
        frames = sdist.unpack_gathered_v(gv, gcv, CAP)
        np.save(os.path.join(result_dir, "counts%d.npy" % rank), np.array([len(f) for f in frames]))
        with open(os.path.join(result_dir, "feats%d.bin" % rank), "wb") as fh:
            for f in frames:
                fh.write(f.tobytes())
    finally:
        dist.destroy_process_group()


def test_two_ranks_extract_and_gather_equals_single_rank(tmp_path):
    import torch.multiprocessing as mp
    ref = _single_rank_reference()
    assert min(len(f) for f in ref) > 300
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = b"".join(f.tobytes() for f in ref)
    for rank in range(world):
        assert np.load(tmp_path / ("counts%d.npy" % rank)).tolist() == [len(f) for f in ref]
        assert open(tmp_path / ("feats%d.bin" % rank), "rb").read() == want, "rank %d" % rank


def _abi_gather(ctx, comm, d_out, d_cnt, n, n_ranks=1):
    import sift_amd
    cap_all = n * CAP
    d_all, d_all_cnt = ctx.malloc(cap_all * 144), ctx.malloc(4 * n * n_ranks)
    try:
        first = ctx.allgather_features(comm, d_out, d_cnt, n, CAP, d_all, cap_all, d_all_cnt, n_ranks)
        ctx.sync()
        cnt = np.zeros(n * n_ranks, dtype=np.uint32)
        ctx.d2h(cnt, d_all_cnt)
        feats = np.zeros(first[-1], dtype=sift_amd.PACKED_DTYPE)
        ctx.d2h(feats, d_all)
        return first, cnt, feats
    finally:
        ctx.free(d_all)
        ctx.free(d_all_cnt)


def test_c_abi_allgather_single_rank_and_rccl_world_of_one():
    """sg_allgather_features without a communicator (one rank, no RCCL call) and with an RCCL communicator of one
    rank (RCCL loaded at run time, counts through ncclAllGather): both give the frames back to back, unpadded."""
    import sift_amd
    ref = _single_rank_reference()
    want = np.concatenate(ref)
    with sift_amd.SiftGPU(max_width=W, max_height=H, max_batch=N_FRAMES) as ctx:
        d_out, d_cnt = _extract_shard(ctx, 0, N_FRAMES)
        try:
            first, cnt, feats = _abi_gather(ctx, None, d_out, d_cnt, N_FRAMES)
            assert first == [0, len(want)] and cnt.tolist() == [len(f) for f in ref]
            assert feats.tobytes() == want.tobytes()
            comm = ctx.comm_create(sift_amd.api.comm_unique_id(), 0, 1)
            try:
                first, cnt, feats = _abi_gather(ctx, comm, d_out, d_cnt, N_FRAMES)
                assert first == [0, len(want)] and cnt.tolist() == [len(f) for f in ref]
                assert feats.tobytes() == want.tobytes()
            finally:
                ctx.comm_destroy(comm)
            # an output too small for the gathered set is refused
            d_small, d_c2 = ctx.malloc(16 * 144), ctx.malloc(4 * N_FRAMES)
            with pytest.raises(sift_amd.SiftGpuError) as e:
                ctx.allgather_features(None, d_out, d_cnt, N_FRAMES, CAP, d_small, 16, d_c2, 1)
            assert e.value.status == sift_amd.api.SG_E_CAPACITY
            ctx.free(d_small)
            ctx.free(d_c2)
        finally:
            ctx.free(d_out)
            ctx.free(d_cnt)
