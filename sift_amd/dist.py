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


def allgather_features_v(packed, counts, cap, group=None):
    """All-gather-v through torch.distributed with the layout of sg_allgather_features (include/siftgpu.h): counts
    are all-gathered, every rank compacts its frames back to back into its slot of one output buffer, and slot r is
    broadcast from rank r.  Used where RCCL through the C ABI cannot be (gloo rehearsals on one GPU or on CPU).

    packed: uint8 tensor holding B slots of cap 144-byte features, counts: int32 tensor [B].
    Returns (all_feats uint8 [total * 144], all_counts int32 [world, B], rank_first list of world + 1 indices)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = counts.numel()
    all_counts = torch.empty(world * B, dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(all_counts, counts.reshape(-1).contiguous(), group=group)
    per = all_counts.reshape(world, B).clamp(max=cap).to(torch.int64).cpu()  # one transfer: the sizes are host-side from here
    rank_first = [0]
    for t in per.sum(dim=1).tolist():
        rank_first.append(rank_first[-1] + int(t))
    out = torch.empty(rank_first[-1] * 144, dtype=torch.uint8, device=packed.device)
    slots = packed.reshape(B, cap * 144)
    mine_counts = per[rank].tolist()
    mine = torch.cat([slots[i, : mine_counts[i] * 144] for i in range(B)]) if B else out[:0]
    out[rank_first[rank] * 144: rank_first[rank + 1] * 144] = mine
    for r in range(world):
        if rank_first[r + 1] > rank_first[r]:
            dist.broadcast(out[rank_first[r] * 144: rank_first[r + 1] * 144], src=dist.get_global_rank(group, r)
                           if group is not None else r, group=group)
    return out, all_counts.reshape(world, B), rank_first


def unpack_gathered_v(all_feats, all_counts, cap):
    """Host view of an all-gather-v result: list (global frame order) of PACKED_DTYPE arrays."""
    flat = all_feats.cpu().numpy().view(PACKED_DTYPE)
    c = np.minimum(all_counts.cpu().numpy().reshape(-1), cap)
    ends = np.cumsum(c)
    return [flat[e - n: e].copy() for e, n in zip(ends, c)]


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
