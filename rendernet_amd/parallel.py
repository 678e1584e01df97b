"""Batch sharding of the render path across the GPUs of one node.

Frames are independent units (no cross-sample op in the net: no batch-norm, dropout off at
inference), so the path shards by frame with NO data-path collective: one process
per GPU (torch.distributed, backend "nccl" = RCCL), weights replicated, every rank renders its
contiguous block of the batch.  The only optional collective is the gather of finished frames to
rank 0 (`gather=True`), which is outside the timed hot path of bench.py.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous block split; the first (n_items % world) ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world %d/%d" % (rank, world))
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def sharded_render(render_fn, voxels, poses, rank=None, world=None, gather=False, group=None):
    """Render this rank's block of (voxels, poses) with `render_fn(vox, pose) -> [b, ...]`.
    gather=False: returns (start, stop, frames) for the local block.
    gather=True:  all ranks return the full [B, ...] result (all_gather of zero-padded blocks)."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = voxels.shape[0]
    s, e = shard_range(B, rank, world)
    local = render_fn(voxels[s:e], poses[s:e]) if e > s else None
    if not gather or world == 1:
        return (s, e, local) if not gather else local
    sizes = [shard_range(B, r, world) for r in range(world)]
    mx = max(b - a for a, b in sizes)
    # frame shape from any non-empty rank (rank 0 always has one when B >= 1)
    shape = torch.tensor(list(local.shape[1:]) if local is not None else [0] * 3, dtype=torch.int64,
                         device=voxels.device if local is None else local.device)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape, group=group)
    fshape = [int(v) for v in max(shapes, key=lambda t: int(t.sum())).tolist()]
    dev = local.device if local is not None else voxels.device
    pad = torch.zeros([mx] + fshape, dtype=torch.float32, device=dev)
    if local is not None:
        pad[: e - s] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bufs[r][: b - a] for r, (a, b) in enumerate(sizes)], 0)
