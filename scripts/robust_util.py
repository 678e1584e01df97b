"""Hostile input statistics and scheme-forcing harness for the Winograd robustness suite (scripts/wino_robustness.py,
tests/test_gpu_wino_robust.py).  Test / measurement infrastructure: nothing on the product path imports it.

The layer under test is the res2 conv of the reference (3x3, SAME, C -> C; tools/layer_util.py:91-105,
RenderNet_Shader.py:71-84).  `hostile_inputs` yields (name, x, w, b) with activations that look like what a TRAINED
net feeds that layer rather than like N(0,1): post-PReLU outputs have a large positive mean and heavy per-channel tails --
the known bad case for large-tile Winograd, whose transforms subtract neighbouring pixels of equal magnitude."""
import numpy as np
import torch

from rendernet_amd import ops

SCHEMES = ("direct", "f22", "f43", "f63", "f43s", "f63s", "f43h", "f63h")      # ...s: the split (bf16x3) GEMM stage of the same scheme; ...h: fp16x2


def xavier(rng, shape):
    rf = int(np.prod(shape[:-2]))
    lim = np.sqrt(6.0 / ((shape[-2] + shape[-1]) * rf))
    return rng.uniform(-lim, lim, shape).astype(np.float32)


def hostile_inputs(rng, B, H, W, Cin, Cout):
    w = xavier(rng, (3, 3, Cin, Cout))
    b = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    n = lambda: rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    yield "N(0,1) inputs, Xavier filter (round-2 statistics)", n(), w, b
    yield "abs(N(0,1)) + 3 inputs (post-PReLU-like positive mean)", (np.abs(n()) + 3.0).astype(np.float32), w, b
    yield "abs(N(0,1)) + 30 inputs (mean 30x the spread)", (np.abs(n()) + 30.0).astype(np.float32), w, b
    gains = np.exp(1.5 * rng.standard_normal(Cin)).astype(np.float32)
    yield "abs(N(0,1)) x log-normal(sigma 1.5) per-channel gains", (np.abs(n()) * gains).astype(np.float32), w, b
    sparse = n() * (rng.random((B, H, W, Cin)) < 0.02)
    yield "2 % sparse spikes x 100 on a +1 floor", (1.0 + 100.0 * np.abs(sparse)).astype(np.float32), w, b
    x = (np.abs(n()) + 3.0).astype(np.float32)
    y = np.abs(x).max() * np.abs(w).sum(axis=(0, 1, 2)).max()
    yield "abs(N)+3 inputs, filter scaled so that outputs reach +-8", x, (w * np.float32(64.0 / y)).astype(np.float32), b
    wp = (np.abs(w) * 0.5 + w * 0.5).astype(np.float32)      # mostly positive filter: no cancellation in y, outputs ~ mean * sum|w|
    yield "abs(N)+3 inputs, 75 % positive filter (large outputs)", x, wp, b


def conv_with_scheme(x, w, b, scheme, alpha=None, residual=None):
    """x [B,H,W,Cin], w [3,3,Cin,Cout] HIP tensors -> conv through the kernel family `scheme` forces."""
    pw = ops.pack_conv(w)
    split = {"s": "split", "h": "split16"}.get(scheme[-1]) if scheme[-1] in "sh" and scheme[:-1] in ("f43", "f63") else None
    if split:
        scheme = scheme[:-1]
    if scheme == "direct":
        pw.wino43 = None
        pw.wino = None
    elif scheme == "f22":
        pw.wino43 = None
        assert pw.wino is not None
    elif scheme == "f43":
        pw.wino63 = None
        assert pw._wino43_kind is not None
    elif scheme == "f63":
        assert pw._wino63_kind is not None
        pw.force_scheme = "f63"
    else:
        raise ValueError(scheme)
    with torch.no_grad(), ops.gemm_mode(split or "f32"):
        return ops.conv2d(x, pw, b, alpha, residual)


def res_stack_weights(rng, C, n_blocks=10):
    """10 res_block_2d + the res2_skip conv: 21 3x3 convs.  Biases positive and PReLU slopes small so that activations carry a
    positive mean from block to block (what trained nets do)."""
    blocks = []
    for _ in range(n_blocks):
        blk = []
        for _ in range(2):
            blk.append((xavier(rng, (3, 3, C, C)), (0.05 + 0.05 * rng.random(C)).astype(np.float32),
                        rng.uniform(0.0, 0.25, C).astype(np.float32)))
        blocks.append(blk)
    skip = (xavier(rng, (3, 3, C, C)), (0.05 * rng.random(C)).astype(np.float32))
    return blocks, skip


def _prelu64(x, a):
    return torch.clamp(x, min=0) + torch.as_tensor(a).double() * torch.clamp(x, max=0)


def res_stack_f64(x0, net, conv_f64):
    """x + conv(prelu(conv(x))) x 10, then conv + shortcut (tools/layer_util.py:91-105, RenderNet_Shader.py:71-84), float64."""
    blocks, skip = net
    x = torch.as_tensor(x0).double()
    short = x
    for (w1, b1, a1), (w2, b2, _a2) in blocks:
        x = x + conv_f64(_prelu64(conv_f64(x, w1, b1), a1), w2, b2)
    return conv_f64(x, skip[0], skip[1]) + short


def res_stack_gpu(x0, net, scheme):
    blocks, skip = net
    d = lambda a: torch.as_tensor(a).cuda()
    x = x0
    for (w1, b1, a1), (w2, b2, _a2) in blocks:
        h = conv_with_scheme(x, d(w1), d(b1), scheme, alpha=d(a1))
        x = conv_with_scheme(h, d(w2), d(b2), scheme, residual=x)
    return conv_with_scheme(x, d(skip[0]), d(skip[1]), scheme, residual=x0)
