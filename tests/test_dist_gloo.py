"""The N > 1 path on CPU: two gloo ranks shard frames and all-gather their feature sets."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_frames, batch, cap, result_dir):
    import torch
    import torch.distributed as dist
    from sift_amd import dist as sdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, stop = sdist.shard_frames(n_frames, world, rank)
        assert stop - start == batch
        packed = np.stack([sdist.fake_features(f, 10 + 3 * f, cap) for f in range(start, stop)])
        counts = np.array([10 + 3 * f for f in range(start, stop)], dtype=np.int32)
        t_packed = torch.from_numpy(packed.view(np.uint8).reshape(-1))
        t_counts = torch.from_numpy(counts)
        g, gc, _ = sdist.allgather_features(t_packed, t_counts)
        frames = sdist.unpack_gathered(g, gc, batch, cap)
        assert len(frames) == n_frames
        for f, got in enumerate(frames):
            ref = sdist.fake_features(f, 10 + 3 * f, cap)[: 10 + 3 * f]
            assert got.tobytes() == ref.tobytes(), "frame %d on rank %d" % (f, rank)
        # all-gather-v: the layout of the C ABI's sg_allgather_features (frames back to back, rank after rank)
        gv, gcv, first = sdist.allgather_features_v(t_packed, t_counts, cap)
        assert first == [0] + list(np.cumsum([sum(10 + 3 * f for f in range(r * batch, (r + 1) * batch))
                                              for r in range(world)]))
        assert gcv.shape == (world, batch) and gv.numel() == first[-1] * 144
        for f, got in enumerate(sdist.unpack_gathered_v(gv, gcv, cap)):
            ref = sdist.fake_features(f, 10 + 3 * f, cap)[: 10 + 3 * f]
            assert got.tobytes() == ref.tobytes(), "all-gather-v frame %d on rank %d" % (f, rank)
        # asynchronous form of the padded exchange
        g2, gc2, handles = sdist.allgather_features(t_packed, t_counts, async_op=True)
        for h in handles:
            h.wait()
        assert torch.equal(g2, g) and torch.equal(gc2, gc)
        open(os.path.join(result_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_shard_frames_cover_everything():
    from sift_amd.dist import shard_frames
    for n, w in [(4096, 8), (256, 1), (10, 4), (3, 8)]:
        seen = []
        for r in range(w):
            a, b = shard_frames(n, w, r)
            seen.extend(range(a, b))
        assert seen == list(range(n))
    assert shard_frames(4096, 8, 3) == (1536, 2048)  # frame f -> GPU f // 512


def test_allgather_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp
    world, batch, cap = 2, 3, 64
    port = _free_port()
    mp.spawn(_worker, args=(world, port, world * batch, batch, cap, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]
