#!/usr/bin/env python
"""Headline benchmark: rendered frames/s, 64^3 voxel -> 512x512 Phong-shader forward, batch 24 per
GPU (BASELINE.json configs[1]), on N MI355X of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path over one batch per rank: fused resampler
(pose -> 128^3 image-aligned grid) + the 75-conv RenderNet forward, inputs already resident in HBM.
Frames are independent, so ranks shard by batch with no data-path collective (weak scaling:
every rank renders its own batch of 24; value = all frames / max-over-ranks time).
Rank 0 prints ONE JSON line (see README/DESIGN.md for the fields `roofline`, `cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FIXTURES = ["chair", "bunny", "table", "suzanne", "teapot"]
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact fp32
GMAC_PER_FRAME = 1056.874         # SURVEY.md §8(d) / App. B (1-channel head)


def synthetic_batch(batch):
    """item i = fixture[i mod 5]; pose az=(250+15i) mod 360, el=60, r=3.3."""
    from rendernet_amd.tools import binvox_rw
    vox = []
    for n in FIXTURES:
        with open(os.path.join(ROOT, "binvox", n + ".binvox"), "rb") as f:
            vox.append(binvox_rw.read_as_3d_array(f).data.astype(np.float32)[..., None])
    vox = np.stack([vox[i % 5] for i in range(batch)])
    az = (250.0 + 15.0 * np.arange(batch)) % 360.0
    poses = np.stack([az * np.pi / 180.0, np.full(batch, (90 - 60) * np.pi / 180.0), np.full(batch, 3.3 / 3.3)], 1)
    return vox, poses.astype(np.float32)


def cpu_baseline(weights, frames=4):
    """The oracle (CPU restatement of the TF graph; the reference itself needs TensorFlow 1.x, which
    is not installable -- SURVEY.md F4) timed on this box's host cores on a bounded sample."""
    import torch
    from oracle import rendernet as ON
    from oracle import resample as OR
    vox, poses = synthetic_batch(frames)
    # pick the thread count that runs the dominant conv fastest on this box (all cores of a large
    # host oversubscribe oneDNN on a batch this small)
    import torch.nn.functional as F
    ncpu = os.cpu_count() or 1
    xx, ww = torch.randn(1, 1024, 64, 64), torch.randn(1024, 1024, 3, 3)
    best, cores = None, 1
    for th in sorted({min(ncpu, t) for t in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(th)
        F.conv2d(xx[:, :, :16], ww, padding=1)
        t0 = time.time()
        F.conv2d(xx, ww, padding=1)
        dt = time.time() - t0
        if best is None or dt < best:
            best, cores = dt, th
    torch.set_num_threads(cores)
    t0 = time.time()
    x = OR.net_input(vox, poses, 64, 128)
    out = ON.rendernet_forward(x, weights)
    dt = time.time() - t0
    assert out.shape == (frames, 512, 512, 1)
    return {"value": round(frames / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d frames (chair, bunny, table, suzanne) at the bench poses, one fp32 pass of the NumPy/torch-CPU oracle "
                      "(resampler + full 237M-parameter net), %.1f s" % (frames, dt)}


def train_main(args, world, rank, local_rank):
    """BASELINE configs[3]: Phong-shader training step, batch 24 per GPU (global batch 24*N), crop `--patch`,
    BCE loss, Adam; gradients summed across ranks with bucketed RCCL all-reduces overlapped with backward."""
    import torch
    import torch.distributed as dist
    from rendernet_amd.shader import ShaderSpec, init_shader_weights
    from rendernet_amd.train import Trainer
    spec = ShaderSpec().check()
    weights = init_shader_weights(spec, seed=1234, perturb=True)
    tr = Trainer(spec, weights, device="cuda:%d" % local_rank)
    B, p = args.batch, args.patch
    vox_np, poses_np = synthetic_batch(B)
    poses_np[:, 0] = (poses_np[:, 0] + rank * 0.1) % (2 * np.pi)
    vox, poses = torch.as_tensor(vox_np).cuda(), torch.as_tensor(poses_np).cuda()
    gen = torch.Generator(device="cuda").manual_seed(11 + rank)
    targets = torch.rand((B, 512, 512, spec.out_ch), device="cuda", generator=gen)
    starts = np.random.default_rng(3).integers(0, spec.new_size - p + 1, size=(args.warmup + args.steps, 2))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        loss = tr.step(vox, poses, targets, patch_size=p, start_point=starts[i])
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = tr.step(vox, poses, targets, patch_size=p, start_point=starts[args.warmup + i])
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    lossv = float(loss.item())
    assert np.isfinite(lossv)
    if rank == 0:
        # forward MACs scale with the crop area; backward = dgrad + wgrad ~ 2x forward
        fwd_tflop = 2e-3 * GMAC_PER_FRAME * (p / float(spec.new_size)) ** 2
        sps = B * world * args.steps / elapsed
        print(json.dumps({
            "metric": "training samples/sec, Phong shader forward+backward+Adam, crop %d of 128^3, batch 24 per GPU" % p,
            "value": round(sps, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Phong shader training step (resampler+crop, forward, BCE, dgrad+wgrad, bucketed "
                                   "gradient all-reduce, Adam), 237.3M params", "batch_per_gpu": B,
                       "global_batch": B * world, "patch": p, "parallelism": "data-parallel x%d, RCCL sum all-reduce" % world},
            "approx_tflops_per_gpu": round(3.0 * fwd_tflop * sps / world, 2),
            "final_loss": lossv}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=24, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["render", "train"], default="render",
                    help="render = the headline metric (BASELINE configs[1]); train = the training step of "
                         "configs[3] (forward + backward + gradient all-reduce + Adam), reported in samples/s")
    ap.add_argument("--patch", type=int, default=64, help="train mode: crop size on the 128^3 grid (RenderNet_Shader.py:204-207)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU render path)")
    # RN_SHARE_GPU=1 + RN_DIST_BACKEND=gloo: several ranks on one device, control collectives over gloo -- only for
    # smoke-testing the multi-rank launch path on a single-GPU box (RCCL refuses two ranks on one device)
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        if not os.environ.get("RN_SHARE_GPU"):
            raise SystemExit("LOCAL_RANK=%d but only %d HIP device(s) visible" % (local_rank, ndev))
        local_rank %= ndev
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes)
        backend = os.environ.get("RN_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from rendernet_amd import ops
    from rendernet_amd.shader import Renderer, ShaderSpec, init_shader_weights

    if args.mode == "train":
        return train_main(args, world, rank, local_rank)
    spec = ShaderSpec().check()
    weights = init_shader_weights(spec, seed=1234, perturb=True)
    renderer = Renderer(spec, weights, device="cuda:%d" % local_rank)
    B = args.batch
    vox_np, poses_np = synthetic_batch(B)
    # each rank renders a different pose set (rank-shifted azimuths): independent shards
    poses_np[:, 0] = (poses_np[:, 0] + rank * 0.1) % (2 * np.pi)
    vox = torch.as_tensor(vox_np).cuda()
    poses = torch.as_tensor(poses_np).cuda()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            out = renderer.render(vox, poses)
        # dominant kernel = the res2 3x3 1024->1024 conv (21 launches/step, 73 % + 3.7 % of FLOPs):
        # bracket each of its launches with HIP events on the launch stream during the timed region
        events, rs_events = [], []

        def hook(mode, xshape, pw):
            if mode == "resample":
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                rs_events.append(ev)
                return ev
            if mode == "conv2d" and pw.cin == spec.w_res2 and pw.cout == spec.w_res2 and pw.kdims[0] == 3:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                events.append(ev)
                return ev
            return None

        ops.LAUNCH_HOOK = hook
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = renderer.render(vox, poses)
        barrier()
        elapsed = time.perf_counter() - t0
        ops.LAUNCH_HOOK = None

    assert out.shape == (B, 512, 512, spec.out_ch)
    assert bool(torch.isfinite(out).all())
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        frames = B * world * args.steps
        fps = frames / elapsed
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in events])) if events else None
        flop_per_launch = 2.0 * (B * 64 * 64) * (9 * spec.w_res2) * spec.w_res2     # M*K*N*2
        achieved = flop_per_launch / (kern_ms * 1e-3) / 1e12 if kern_ms else None
        traffic = rs_traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("conv_igemm_res2_hbm_bytes_per_launch")
                rs_traffic = tj.get("resampler", {}).get("hbm_bytes_per_call")
            except Exception:
                traffic = rs_traffic = None
        res = {
            "metric": "rendered frames/sec, 64^3 voxel->512x512 Phong, batch 24",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Phong shader forward (resampler + RenderNet 1-ch head), 5 shipped binvox fixtures "
                                   "cycled, 64^3 -> 128^3 -> 512x512, seeded random weights (237.3M params)",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": "batch-sharded x%d, no collective" % world},
            "fraction_of_fp32_conv_roofline": round(fps / world * GMAC_PER_FRAME * 2e-3 / PEAK_FP32_MFMA_TFLOPS, 4),
            "roofline": {"kernel": "conv_igemm_glds_kernel (128x128x32 tile, LDS-DMA) on res2 3x3 1024->1024 @64x64xB",
                         "bound": "mfma", "achieved": round(achieved, 2) if achieved else None,
                         "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4) if achieved else None,
                         "avg_launch_ms": round(kern_ms, 4) if kern_ms else None,
                         "launches_timed": len(events), "flop_per_launch": flop_per_launch,
                         "traffic": traffic},
        }
        if rs_events:
            # second roofline of the path: the resampler is HBM-bound (9 437 184 algorithmic bytes per
            # frame = 1 MiB source read + 8 MiB grid written); its three launches are bracketed together
            rs_ms = float(np.mean([a.elapsed_time(b) for a, b in rs_events]))
            rs_bytes = B * 9437184.0
            res["roofline_resampler"] = {
                "kernel": "resample_prepare + resample_classify + resample_main (csrc/resample_tiled.hip)",
                "bound": "hbm", "achieved": round(rs_bytes / (rs_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(rs_bytes / (rs_ms * 1e-3) / 1e9 / 8000.0, 4), "avg_ms": round(rs_ms, 4),
                "launches_timed": len(rs_events), "bytes_per_call": rs_bytes, "traffic": rs_traffic if B == 24 else None}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(weights)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
