"""Multi-GPU plumbing: frames are independent units, sharded contiguously over ranks; the only exchange is an
all-gather of the per-rank packed feature sets (RCCL over xGMI on GPUs, gloo on CPU in tests).

torch.distributed is used here as plumbing only; extraction itself never communicates.
"""
import numpy as np

from .api import PACKED_DTYPE


def shard_frames(n_frames, world_size, rank):
    """Contiguous shard [start, stop) of rank `rank`: frame f belongs to rank f // ceil(n/world)."""
    per = (n_frames + world_size - 1) // world_size
    start = min(n_frames, rank * per)
    return start, min(n_frames, start + per)


def allgather_features(packed, counts, group=None, async_op=False, out=None, out_counts=None):
    """All-gather fixed-capacity feature buffers.

    packed: uint8 tensor of B*cap*144 bytes (any shape), counts: int32 tensor [B].
    Returns (gathered [world * packed.numel()], gathered_counts [world * B], handles).  Rank r's frames occupy
    slice r, so the global order is (rank, local frame) = global frame order for contiguous shards.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty(world * packed.numel(), dtype=packed.dtype, device=packed.device)
    if out_counts is None:
        out_counts = torch.empty(world * counts.numel(), dtype=counts.dtype, device=counts.device)
    h1 = dist.all_gather_into_tensor(out_counts, counts.reshape(-1), group=group, async_op=async_op)
    h2 = dist.all_gather_into_tensor(out, packed.reshape(-1), group=group, async_op=async_op)
    return out, out_counts, (h1, h2) if async_op else None


def unpack_gathered(gathered, gathered_counts, batch, cap):
    """Host view of a gathered buffer: list (global frame order) of PACKED_DTYPE arrays."""
    g = gathered.cpu().numpy().view(PACKED_DTYPE).reshape(-1, cap)
    c = gathered_counts.cpu().numpy().reshape(-1)
    assert g.shape[0] == c.shape[0] and g.shape[0] % batch == 0
    return [g[i, : c[i]].copy() for i in range(g.shape[0])]


def fake_features(frame_id, n, cap):
    """Deterministic stand-in feature set of frame `frame_id` (tests of the exchange step without a GPU)."""
    rng = np.random.default_rng(1000 + frame_id)
    a = np.zeros(cap, dtype=PACKED_DTYPE)
    a["x"][:n] = rng.random(n, dtype=np.float32) * 1920
    a["y"][:n] = rng.random(n, dtype=np.float32) * 1080
    a["scale"][:n] = np.sort(rng.random(n, dtype=np.float32) * 40 + 1)
    a["ori"][:n] = rng.random(n, dtype=np.float32) * 6.28 - 3.14
    a["desc"][:n] = rng.integers(0, 160, size=(n, 128), dtype=np.uint8)
    return a
