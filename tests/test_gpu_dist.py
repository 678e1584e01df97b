"""Two ranks over RCCL (torch.distributed backend "nccl") on two MI355X: the data-parallel training step and the
frame-sharded render.  -m gpu; skipped on boxes with fewer than two HIP devices (the single-GPU boxes run the same
logic over gloo in tests/test_dist_cpu.py and through `bench.py --gpus 2` with RN_SHARE_GPU=1)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs():
    rng = np.random.default_rng(0)
    vox = (rng.random((4, 16, 16, 16, 1)) < 0.3).astype(np.float32)
    poses = np.array([[1.0, 0.6, 1.0], [4.0, 0.4, 0.9], [2.5, 0.9, 1.1], [0.3, 0.2, 1.0]], np.float32)
    tgt = rng.random((4, 128, 128, 1)).astype(np.float32)
    return vox, poses, tgt


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from rendernet_amd.parallel import shard_range, sharded_render
    from rendernet_amd.shader import Renderer, tiny_spec, init_shader_weights
    from rendernet_amd.train import Trainer
    spec = tiny_spec(1)
    w = init_shader_weights(spec, seed=3, perturb=True)
    vox, poses, tgt = _inputs()
    lo, hi = shard_range(4, rank, world)
    tr = Trainer(spec, w, device="cuda:%d" % rank, e_eta=1e-3, bucket_mb=0.5)
    loss = tr.step(vox[lo:hi], poses[lo:hi], tgt[lo:hi], patch_size=16, start_point=(3, 5), global_batch=4)
    r = Renderer(spec, w, device="cuda:%d" % rank)
    full = sharded_render(lambda v, p: r.render(v, p), torch.as_tensor(vox).cuda(), torch.as_tensor(poses).cuda(), gather=True)
    q.put((rank, float(loss.item()), tr.grad.cpu().numpy(), tr.param.cpu().numpy(), full.cpu().numpy(), len(tr.buckets.buckets)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two HIP devices")
def test_two_rank_rccl_training_step_and_sharded_render_match_one_rank():
    from rendernet_amd.shader import Renderer, tiny_spec, init_shader_weights
    from rendernet_amd.train import Trainer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    spec = tiny_spec(1)
    w = init_shader_weights(spec, seed=3, perturb=True)
    vox, poses, tgt = _inputs()
    tr = Trainer(spec, w, e_eta=1e-3)
    loss = tr.step(vox, poses, tgt, patch_size=16, start_point=(3, 5))
    g1, p1 = tr.grad.cpu().numpy(), tr.param.cpu().numpy()
    want = Renderer(spec, w).render(vox, poses).cpu().numpy()
    for rank, l2, g2, p2, full, nb in res:
        assert nb >= 2                                                  # several buckets went over RCCL
        assert abs(l2 - float(loss.item())) <= 1e-5 * abs(float(loss.item()))
        assert np.abs(g2 - g1).max() <= 1e-4 * np.abs(g1).max()        # summed shard gradients == the full-batch gradient
        # Adam divides by |g| + 1e-8: a parameter whose gradient is ~1e-8 moves by anything in [-lr, lr] on the last bits of
        # the summed gradient -- compare all but the 0.1 % least conditioned parameters
        assert np.quantile(np.abs(p2 - p1), 0.999) <= 1e-5 * np.abs(p1).max() + 1e-7
        assert np.abs(full - want).max() <= 1e-6                        # frames are independent: the shards reassemble
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])   # replicas stay identical
