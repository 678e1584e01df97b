"""Generate the golden vectors under tests/golden/ with the CPU oracle (run in the build
container: `python tests/golden/make_golden.py`).  The reference itself cannot be executed
(TensorFlow 1.x is not installable here), so these vectors pin the HIP path to the
oracle restatement, not to TF outputs -- "parity unpinned" in the sense of oracle/__init__.py.

  tiny_shader.npz             tiny spec (16^3 -> 32^3 -> 128^2), 3 frames, perturbed weights seed 1234
  full_chair_demo_pose.npz    reference-size net, binvox/chair.binvox at the demo default pose
                              (RenderNet_demo.py:81-98), 128x128 centre crop of logits and output
  bench_frames.npz            (`make_golden.py bench_frames`) five frames of bench.py's batch, one per fixture
  stress_bench_frames.npz     (`make_golden.py stress8`) the 8 frames of `bench.py --mode stress`, one crop each
  texture_bench_frames.npz    (`make_golden.py texture_bench`) frames 0-3 of `bench.py --mode texture` (BASELINE configs[2]):
                              centre crops of both heads' images and logits
  train_step_golden.npz       (`make_golden.py train_step`) BASELINE configs[3] at full width: the first two samples of
                              `bench.py --mode train` (crop 64 at (31, 17), BCE): loss, a crop of the prediction, and four
                              sampled entries + the max of every one of the 166 parameter gradients (torch-CPU autograd
                              over the oracle graph, oracle/train.py), plus the resampled + cropped grid they were computed on
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import rendernet as ON                     # noqa: E402
from oracle import resample as OR                      # noqa: E402
from oracle.io_phong import read_binvox                # noqa: E402
from rendernet_amd.shader import ShaderSpec, tiny_spec, init_shader_weights   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def pose(az, el, r):
    return np.array([az * np.pi / 180.0, (90 - el) * np.pi / 180.0, 3.3 / r], np.float32)


def tiny():
    spec = tiny_spec(1)
    w = init_shader_weights(spec, seed=1234, perturb=True)
    rng = np.random.default_rng(2024)
    vox = (rng.random((3, 16, 16, 16, 1)) < 0.3).astype(np.float32)
    poses = np.stack([pose(250, 60, 3.3), pose(15, 35, 2.9), pose(200, 70, 4.1)])
    taps = {}
    x = OR.net_input(vox, poses, 16, 32, mode="tf")
    out = ON.rendernet_forward(x, w, taps, spec.n_res1, spec.n_res2, spec.n_res3)
    np.savez_compressed(os.path.join(HERE, "tiny_shader.npz"), seed=1234, vox=vox, poses=poses, output=out,
                        tap_enc3=taps["enc3"], tap_enc4=taps["enc4"], tap_enc6=taps["enc6"],
                        tap_logits=taps["logits"])


def full():
    spec = ShaderSpec().check()
    w = init_shader_weights(spec, seed=1234, perturb=True)
    vox = read_binvox(os.path.join(ROOT, "binvox", "chair.binvox")).astype(np.float32)[None, ..., None]
    p = pose(250, 60, 3.3)[None]
    x = OR.net_input(vox, p, 64, 128, mode="tf")
    taps = {}
    out = ON.rendernet_forward(x, w, taps)
    np.savez_compressed(os.path.join(HERE, "full_chair_demo_pose.npz"),
                        output_crop=out[0, 192:320, 192:320, 0], logits_crop=taps["logits"][0, 192:320, 192:320, 0],
                        net_in_sum=np.float64(x.sum()), enc3_skip_absmean=np.float64(np.abs(taps["enc3_skip"]).mean()),
                        enc4_absmean=np.float64(np.abs(taps["enc4"]).mean()))


def stress():
    """BASELINE config 5 (128^3 -> 256^3 -> 1024^2), one frame: chair upsampled x2 (nearest), demo pose."""
    from rendernet_amd.shader import stress_spec
    spec = stress_spec(1)
    w = init_shader_weights(spec, seed=1234, perturb=True)
    vox = read_binvox(os.path.join(ROOT, "binvox", "chair.binvox")).astype(np.float32)
    vox = vox.repeat(2, 0).repeat(2, 1).repeat(2, 2)[None, ..., None]
    p = pose(250, 60, 3.3)[None]
    x = OR.net_input(vox, p, 128, 256, mode="tf")
    taps = {}
    out = ON.rendernet_forward(x, w, taps)
    np.savez_compressed(os.path.join(HERE, "stress_chair_demo_pose.npz"),
                        output_crop=out[0, 448:576, 448:576, 0], logits_crop=taps["logits"][0, 448:576, 448:576, 0],
                        net_in_sum=np.float64(x.sum()), enc4_absmean=np.float64(np.abs(taps["enc4"]).mean()))


BENCH_FRAMES = [0, 6, 12, 18, 9]          # of the bench batch: chair, bunny, table, suzanne, teapot at az 250, 340, 70, 160, 25
CROPS = [(128, 128), (128, 256), (256, 128), (256, 256)]      # 128x128 crops tiling the central 256x256 (where the object is)


def _bench_inputs(n, upsample=1):
    """The synthetic batch of bench.py: item i = fixture[i mod 5], pose az = (250+15i) mod 360."""
    names = ["chair", "bunny", "table", "suzanne", "teapot"]
    vox = [read_binvox(os.path.join(ROOT, "binvox", m + ".binvox")).astype(np.float32) for m in names]
    if upsample > 1:
        vox = [v.repeat(upsample, 0).repeat(upsample, 1).repeat(upsample, 2) for v in vox]
    v = np.stack([vox[i % 5] for i in range(n)])[..., None]
    p = np.stack([pose((250.0 + 15.0 * i) % 360.0, 60.0, 3.3) for i in range(n)])
    return v, p


def bench_frames():
    """Five frames OF THE BENCHED BATCH (BASELINE configs[1]; one per shipped fixture, five different azimuths) through
    the full-size oracle: four 128x128 crops of the image and of the logits per frame, plus strided samples of the 3-D
    encoder output (enc3_skip) and of the projection unit's output (enc4)."""
    spec = ShaderSpec().check()
    w = init_shader_weights(spec, seed=1234, perturb=True)
    vox, poses = _bench_inputs(24)
    out = {"frames": np.array(BENCH_FRAMES), "crops": np.array(CROPS)}
    for k, i in enumerate(BENCH_FRAMES):
        x = OR.net_input(vox[i:i + 1], poses[i:i + 1], 64, 128, mode="tf")
        taps = {}
        img = ON.rendernet_forward(x, w, taps)
        out["output_%d" % k] = np.stack([img[0, r:r + 128, c:c + 128, 0] for r, c in CROPS])
        out["logits_%d" % k] = np.stack([taps["logits"][0, r:r + 128, c:c + 128, 0] for r, c in CROPS])
        out["enc3_skip_%d" % k] = taps["enc3_skip"][0, 3::8, 5::8, 1::4, :]
        out["enc4_%d" % k] = taps["enc4"][0, 3::8, 5::8, :]
        out["net_in_sum_%d" % k] = np.float64(x.sum())
        print("bench frame", i, "done", flush=True)
    np.savez_compressed(os.path.join(HERE, "bench_frames.npz"), **out)


def stress8():
    """BASELINE config 5 at its configured batch: the 8 frames of `bench.py --mode stress` (fixtures upsampled x2, bench
    poses) through the oracle, one 128x128 crop of image and logits per frame (the oracle needs minutes per frame)."""
    from rendernet_amd.shader import stress_spec
    spec = stress_spec(1)
    w = init_shader_weights(spec, seed=1234, perturb=True)
    vox, poses = _bench_inputs(8, 2)
    out = {}
    for i in range(8):
        x = OR.net_input(vox[i:i + 1], poses[i:i + 1], 128, 256, mode="tf")
        taps = {}
        img = ON.rendernet_forward(x, w, taps)
        out["output_%d" % i] = img[0, 448:576, 448:576, 0]
        out["logits_%d" % i] = taps["logits"][0, 448:576, 448:576, 0]
        print("stress frame", i, "done", flush=True)
        np.savez_compressed(os.path.join(HERE, "stress_bench_frames.npz"), **out)


TEXTURE_FRAMES = [0, 1, 2, 3]            # of the texture bench batch (strong scaling over 8 ranks still leaves rank 0 three of them)
TRAIN_START, TRAIN_PATCH, TRAIN_B = (31, 17), 64, 2


def texture_bench():
    """Frames 0-3 of `bench.py --mode texture` (geometry = the bench batch, texture codes ~N(0,1) seed 7, bench poses)
    through oracle/texture_net.py: the 128x128 centre crop of both heads' images and logits."""
    from oracle import texture_net as OT
    from rendernet_amd.texture import TextureSpec, init_texture_weights
    spec = TextureSpec().check()
    w = init_texture_weights(spec, seed=1234, perturb=True)
    vox, poses = _bench_inputs(24)
    z = np.random.default_rng(7).standard_normal((24, spec.z_dim)).astype(np.float32)      # bench.texture_codes
    out = {"frames": np.array(TEXTURE_FRAMES), "crop": np.array([192, 320])}
    for k, i in enumerate(TEXTURE_FRAMES):
        taps = {}
        img, nrm = OT.render_texture(vox[i:i + 1], z[i:i + 1], poses[i:i + 1], w, taps=taps)
        out["image_%d" % k] = img[0, 192:320, 192:320]
        out["normal_%d" % k] = nrm[0, 192:320, 192:320]
        out["image_logits_%d" % k] = taps["image_logits"][0, 192:320, 192:320]
        out["normal_logits_%d" % k] = taps["normal_logits"][0, 192:320, 192:320]
        print("texture frame", i, "done", flush=True)
    np.savez_compressed(os.path.join(HERE, "texture_bench_frames.npz"), **out)


def train_step():
    """The full-width Phong-shader net's training gradient (RenderNet_Shader.py:154-167) on the first TRAIN_B samples of
    `bench.py --mode train`'s batch: loss + sampled gradient entries of every variable (a full gradient is 949 MB)."""
    from oracle import train as OTR
    spec = ShaderSpec().check()
    w = init_shader_weights(spec, seed=1234, perturb=True)
    vox, poses = _bench_inputs(TRAIN_B)
    target = np.random.default_rng(11).uniform(0, 1, (TRAIN_B, 512, 512, 1)).astype(np.float32)
    full = OR.net_input(vox, poses, 64, 128, mode="tf")
    net_in, tgt = OTR.crop_voxel_image(full, target, TRAIN_START, TRAIN_PATCH)
    net_in, tgt = np.ascontiguousarray(net_in), np.ascontiguousarray(tgt)
    # float64: bias / PReLU-slope gradients are sums of ~10^5 mixed-sign terms -- in float32 the REFERENCE's own summation
    # noise (2e-3 of the largest entry, measured) would be what the comparison sees
    loss, grads, pred = OTR.loss_and_grads(net_in, tgt, w, dtype=np.float64)
    pred = pred.astype(np.float32)
    rng = np.random.default_rng(5)
    names = sorted(grads.keys())
    idx = np.zeros((len(names), 4), np.int64)
    val = np.zeros((len(names), 4), np.float32)
    gmax = np.zeros(len(names), np.float32)
    for k, n in enumerate(names):
        g = grads[n].reshape(-1)
        # two random entries and the two largest ones (random entries of a 9.4M-entry filter gradient are mostly tiny)
        top = np.resize(np.argsort(np.abs(g))[-2:], 2)          # (a 1-element bias: the same entry twice)
        idx[k] = np.concatenate([rng.integers(0, g.size, 2), top])
        val[k] = g[idx[k]].astype(np.float32)
        gmax[k] = np.abs(g).max()
    np.savez_compressed(os.path.join(HERE, "train_step_golden.npz"), names=np.array(names), idx=idx, val=val, gmax=gmax,
                        loss=np.float64(loss), pred_crop=pred[:, 64:192, 64:192, 0], net_in=net_in, target=tgt,
                        start=np.array(TRAIN_START), patch=np.int64(TRAIN_PATCH))
    print("train step golden: loss %.6f, %d variables" % (loss, len(names)))


if __name__ == "__main__":
    if "texture_bench" in sys.argv:
        texture_bench()
        sys.exit(0)
    if "train_step" in sys.argv:
        train_step()
        sys.exit(0)
    if "stress" in sys.argv:
        stress()
        sys.exit(0)
    if "bench_frames" in sys.argv:
        bench_frames()
        sys.exit(0)
    if "stress8" in sys.argv:
        stress8()
        sys.exit(0)
    tiny()
    full()
    print("golden vectors written to", HERE)
