"""world_size-2 `gloo` test of the batch sharding (rendernet_amd/parallel.py) on CPU."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    from rendernet_amd.parallel import shard_range
    for n in (0, 1, 5, 24, 25, 192):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [shard_range(24, r, 8) for r in range(8)][3] == (9, 12)
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rendernet_amd.parallel import sharded_render
    vox = torch.arange(n_frames * 8, dtype=torch.float32).reshape(n_frames, 2, 2, 2, 1)
    poses = torch.arange(n_frames * 3, dtype=torch.float32).reshape(n_frames, 3)
    calls = []

    def fake_render(v, p):            # stands in for Renderer.render (no GPU here): per-frame function
        calls.append(v.shape[0])
        return (v.sum(dim=(1, 2, 3, 4)) + p.sum(dim=1)).reshape(-1, 1, 1, 1) * torch.ones(1, 2, 2, 1)

    s, e, local = sharded_render(fake_render, vox, poses, gather=False)
    full = sharded_render(fake_render, vox, poses, gather=True)
    want = fake_render(vox, poses)
    ok = bool(torch.equal(full, want)) and (local is None or bool(torch.equal(local, want[s:e])))
    q.put((rank, s, e, ok, calls[0] if calls else 0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 24])
def test_two_rank_gloo_sharded_render(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_frames
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, e0, ok0, c0), (r1, s1, e1, ok1, c1) = res
    assert ok0 and ok1
    assert s0 == 0 and e0 == s1 and e1 == n_frames and c0 == e0 - s0 and c1 == e1 - s1


# ---------------------------------------------------------------------------------------------
# gradient all-reduce of the training step (rendernet_amd/train.py: GradBuckets), world size 2, gloo
# ---------------------------------------------------------------------------------------------
def _bucket_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rendernet_amd.train import GradBuckets
    sizes = [("w%d" % i, k) for i, k in enumerate([1000, 7, 33, 4096, 5, 2000, 64, 1])]
    layout, off = {}, 0
    for n, k in sizes:
        layout[n] = (off, k)
        off += (k + 3) // 4 * 4
    order = [n for n, _ in reversed(sizes)]                  # backward finishes the last layer first
    flat = torch.zeros(off)
    gb = GradBuckets(flat, layout, order, bucket_mb=4 * 2100 / 1e6)
    ok = len(gb.buckets) >= 3
    for step in range(2):                                     # two steps: reset() must re-arm the buckets
        flat.zero_()
        gb.reset()
        launched_before_finish = 0
        for n in order:
            o, k = layout[n]
            flat[o:o + k] = (rank + 1) * (step + 1) * torch.arange(1, k + 1, dtype=torch.float32)   # "wgrad" of n
            gb.ready(n)
            gb.ready(n)                                       # idempotent
            launched_before_finish = len(gb.launched)
        ok = ok and launched_before_finish == len(gb.buckets)  # every bucket went out during the "backward"
        gb.finish()
        for n in order:
            o, k = layout[n]
            want = 3.0 * (step + 1) * torch.arange(1, k + 1, dtype=torch.float32)   # (1 + 2) * ...
            ok = ok and bool(torch.equal(flat[o:o + k], want))
        # a parameter that never reports (no gradient this step) is flushed by finish()
    flat.zero_()
    gb.reset()
    for n in order[:-1]:
        gb.ready(n)
    n_before = len(gb.launched)
    gb.finish()
    ok = ok and n_before == len(gb.buckets) - 1 and len(gb.launched) == len(gb.buckets)
    q.put((rank, ok, len(gb.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_bucketed_gradient_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]
