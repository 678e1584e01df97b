"""tf.nn.dropout on the HIP path (rn_dropout) -- the op behind RenderNet_Shader.py:39,43,47,88,103,107-123 and
RenderNet_Texture_Face_Normal.py:55-142 (`keep_prob` of tools/layer_util.py:124-131; README default 0.75).  -m gpu.
TensorFlow's random stream cannot be reproduced; what is pinned: the formula x/kp*floor(kp+u) bit for bit against the
NumPy Philox restatement (oracle/dropout.py, itself pinned to the published Philox known answers), its statistics,
and that the backward pass regenerates the forward mask."""
import os

import numpy as np
import pytest
import torch

from oracle import dropout as OD

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,kp", [(4096 * 7 + 3, 0.75), (100003, 0.5), (16, 0.9)])
def test_dropout_matches_the_oracle_bit_for_bit(n, kp):
    from rendernet_amd import ops
    x = torch.randn(n, device="cuda")
    y = ops.dropout(x, kp, seed=0x1234567890ABCDEF, stream_id=5 + (1 << 33))
    want = OD.dropout(x.cpu().numpy(), kp, 0x1234567890ABCDEF, 5 + (1 << 33))
    assert np.array_equal(y.cpu().numpy(), want)


def test_dropout_statistics_and_backward_mask():
    from rendernet_amd import ops
    kp = 0.75
    gen = torch.Generator(device="cuda").manual_seed(5)                                 # fixed data: the test is deterministic
    x = torch.rand((8, 64, 64, 32), device="cuda", generator=gen) + 0.5
    x.requires_grad_(True)
    ops.seed_dropout(99)
    y = ops.dropout(x, kp)
    kept = (y != 0)
    rate = float(kept.float().mean())
    assert abs(rate - kp) < 4 * np.sqrt(kp * (1 - kp) / x.numel()) + 1e-4            # keep rate = keep_prob
    assert torch.allclose(y[kept], (x / kp)[kept])                                      # survivors scaled by 1/kp
    assert abs(float(y.detach().mean()) / float(x.detach().mean()) - 1.0) < 5e-3                          # E[y] = x
    g = torch.rand(y.shape, device="cuda", generator=gen) + 0.25                        # never zero: (grad != 0) is the mask
    y.backward(g)
    assert torch.equal(x.grad != 0, kept) and torch.allclose(x.grad[kept], (g / kp)[kept])   # same mask forward / backward
    z = ops.dropout(x.detach(), kp)                                                     # next call: a fresh stream
    assert float(((z != 0) != kept).float().mean()) > 0.2
    ops.seed_dropout(99)
    assert torch.equal(ops.dropout(x.detach(), kp), y.detach())                         # re-seeding replays the masks
    assert ops.dropout(x, 1.0) is x                                                     # keep_prob 1: identity (inference)


def test_training_step_with_dropout_runs_and_eval_is_deterministic():
    """keep_prob < 1 through the whole training graph (nine dropout sites, RenderNet_Shader.py:39-123): finite loss and
    gradients; is_training=False (the reference's validation feed) switches every site off."""
    from rendernet_amd.shader import tiny_spec, init_shader_weights
    from rendernet_amd.train import Trainer
    spec = tiny_spec(1)
    tr = Trainer(spec, init_shader_weights(spec, seed=3, perturb=True), keep_prob=0.75)
    rng = np.random.default_rng(0)
    vox = (rng.random((2, 16, 16, 16, 1)) < 0.3).astype(np.float32)
    poses = np.array([[1.0, 0.6, 1.0], [4.0, 0.4, 0.9]], np.float32)
    tgt = rng.random((2, 128, 128, 1)).astype(np.float32)
    a, _ = tr.forward(vox, poses, is_training=False)
    b, _ = tr.forward(vox, poses, is_training=False)
    assert torch.equal(a, b)
    # masks are a pure function of (trainer seed, rank, global step, dropout site): an extra forward between two steps (the
    # sample render every 600 steps) re-draws the coming step's masks instead of shifting every later one ...
    c, _ = tr.forward(vox, poses, is_training=True)
    d, _ = tr.forward(vox, poses, is_training=True)
    assert torch.equal(c, d) and not torch.equal(c, a)
    loss = tr.step(vox, poses, tgt, patch_size=16, start_point=(3, 5))
    assert np.isfinite(float(loss.item())) and bool(torch.isfinite(tr.grad).all()) and float(tr.grad.abs().max()) > 0
    # ... every step draws new ones (global_step went 0 -> 1) ...
    from rendernet_amd import ops
    assert ops.dropout_state() [1] >> 16 == 0                           # the step's forward drew streams (0 << 16) + k
    tr.forward(vox, poses, is_training=True)
    assert ops.dropout_state()[1] >> 16 == 1 and ops.dropout_state()[0] == tr.dropout_seed
    # ... and another rank of the same job draws different ones for its shard (one process-global constant seed
    # gave every rank the same mask sequence)
    assert tr.dropout_seed == ops.mix_seed(1234, 0) != ops.mix_seed(1234, 1)
    e1, _ = tr.forward(vox, poses, is_training=True)
    tr.dropout_seed = ops.mix_seed(1234, 1)
    e2, _ = tr.forward(vox, poses, is_training=True)
    assert not torch.equal(e1, e2)


def test_dropout_takes_unaligned_views_and_empty_tensors():
    """rn_dropout on a float-aligned (not 16-byte aligned) pointer -- the gradient of an offset view -- draws the same mask per
    ELEMENT as the aligned call, and an empty tensor is a no-op."""
    from rendernet_amd import ops
    base = torch.randn(4099, device="cuda")
    x = base[1:]                                                        # contiguous, 4 bytes off the 16-byte grid
    assert x.data_ptr() % 16 != 0
    y = ops.dropout(x, 0.6, seed=77, stream_id=3)
    assert np.array_equal(y.cpu().numpy(), OD.dropout(x.cpu().numpy(), 0.6, 77, 3))
    xg = x.clone().requires_grad_(True)
    g = torch.randn(4102, device="cuda")[3:3 + 4098]                     # an offset view as the incoming gradient
    ops.dropout(xg, 0.6, seed=77, stream_id=3).backward(g)
    assert np.array_equal(xg.grad.cpu().numpy(), OD.dropout(g.cpu().numpy(), 0.6, 77, 3))
    e = torch.empty(0, device="cuda")
    assert ops.dropout(e, 0.5).numel() == 0


def test_checkpoint_resume_continues_the_trajectory(tmp_path):
    """Trainer.save_checkpoint / load_checkpoint carry weights, Adam moments and global_step: a trainer restarted from the
    checkpoint and the trainer that wrote it take the SAME next step (to within the last bits the fp32-atomic filter gradients
    leave), while a weights-only restart -- Adam at t = 1 with zero moments, the learning-rate staircase back at step 0 -- does
    not.  The comparison is against the continuation of the very trainer that saved, not against a second run from scratch: Adam
    turns the rounding noise of a (near-)zero gradient into a full +-lr step, so two from-scratch runs of three steps land on one
    of two trajectories 1.5e-5 apart at the median, at random (measured in round 4: four trainers from the same weights, six
    processes) -- that says nothing about checkpoints.  Distances are medians / a fraction, for the same reason: a handful of such parameters may flip in the one step
    compared here too."""
    from rendernet_amd.shader import tiny_spec, init_shader_weights
    from rendernet_amd.train import Trainer
    spec = tiny_spec(1)
    w = init_shader_weights(spec, seed=3, perturb=True)
    rng = np.random.default_rng(0)
    vox = (rng.random((2, 16, 16, 16, 1)) < 0.3).astype(np.float32)
    poses = np.array([[1.0, 0.6, 1.0], [4.0, 0.4, 0.9]], np.float32)
    tgt = rng.random((2, 128, 128, 1)).astype(np.float32)
    kw = dict(e_eta=1e-3, decay_steps=2)

    def run(tr, steps):
        for i in steps:
            tr.step(vox, poses, tgt, patch_size=16, start_point=(i, 2 * i))
        return tr

    a = run(Trainer(spec, w, **kw), range(2))
    path = str(tmp_path / "ck.npz")
    a.save_checkpoint(path, epoch=7)
    assert not os.path.exists(path + ".tmp.npz")
    b = Trainer(spec, init_shader_weights(spec, seed=99), **kw)
    assert b.load_checkpoint(dict(np.load(path))) == 7 and b.global_step == 2
    assert torch.equal(b.param, a.param) and torch.equal(b.m, a.m) and torch.equal(b.v, a.v)
    c = Trainer(spec, {k: v for k, v in a.state_dict().items()}, **kw)     # weights only: the round-1 checkpoint
    before = a.param.clone()
    run(a, [2])
    run(b, [2])
    run(c, [2])
    assert b.global_step == 3 and a.global_step == 3
    step = float((a.param - before).abs().median())                 # what one step moves the median parameter by
    d_resumed, d_cold = (b.param - a.param).abs(), (c.param - a.param).abs()
    assert step > 1e-5
    assert float(d_resumed.median()) <= 1e-7 and float((d_resumed > 1e-6).float().mean()) <= 0.01, \
        (float(d_resumed.median()), float((d_resumed > 1e-6).float().mean()), float(d_resumed.max()))
    assert float(d_cold.median()) > 0.05 * step and float(d_cold.median()) > 100 * float(d_resumed.median()), (float(d_cold.median()), step)
