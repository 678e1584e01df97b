"""The split (bf16x3, fp32-accumulate) multiply stage of the three-launch Winograd path (csrc/conv_wino_bf3.hip).  -m gpu.

The layers it serves are the wide stride-1 2-D convs of the reference -- res_block_2d / *_skip (slim.conv2d [3,3],
tools/layer_util.py:91-105, RenderNet_Shader.py:71-84, :91-99) and e_conv5 / e_conv6 (slim.conv2d [4,4], :86-88, :101-103).
Gates: the SAME bars as the exact-fp32 route -- every conv flavour <= 1e-4 * max|ref| against the oracle conv
(oracle/layers.py), the hostile-statistics suite against float64 (tests/test_gpu_wino_robust.py carries the split schemes
in its table), every sampled tap of the benched frames <= 2e-4 * max, image <= 1e-3 -- plus statements of the two layouts
(packed filter, transformed input) in NumPy: the three bf16 pieces must sum to the fp32 value they replace."""
import ctypes
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import layers as OL

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _xavier(rng, shape):
    rf = int(np.prod(shape[:-2]))
    lim = np.sqrt(6.0 / ((shape[-2] + shape[-1]) * rf))
    return rng.uniform(-lim, lim, shape).astype(np.float32)


def _close(got, want, what, rtol=1e-4):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = want.detach().cpu().numpy() if torch.is_tensor(want) else np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err, ref = float(np.abs(got - want).max()), float(np.abs(want).max())
    assert err <= rtol * ref + 1e-7, "%s: max err %g, max |ref| %g" % (what, err, ref)


def _bf16_to_f64(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def _rows_to_planes(rows, nrows_axis_bit3):
    """rows [..., R, 48] uint16 (96-byte rows: 3 planes x 2 chunks x 8 bf16, the chunks swapped where bit 3 of the row index
    is set) -> [..., R, 3, 16] uint16 in channel order."""
    r = rows.reshape(rows.shape[:-1] + (3, 2, 8)).copy()
    swap = nrows_axis_bit3.astype(bool)
    r[..., swap, :, :, :] = r[..., swap, :, ::-1, :]
    return r.reshape(rows.shape[:-1] + (3, 16))


SCHEMES = {"f43": (0, 36, 4, 3), "f44": (1, 49, 4, 4), "f63": (2, 64, 6, 3)}      # name -> (scheme id, planes, tile, filter)


@pytest.mark.parametrize("which,cin,cout,transposed", [("f43", 64, 256, 0), ("f63", 96, 512, 0), ("f44", 32, 256, 0),
                                                         ("f63", 256, 256, 1), ("f44", 256, 256, 1)])
def test_split_pack_is_the_fp32_pack_in_three_pieces(which, cin, cout, transposed):
    """rn_winograd_split_pack: Us [nxi][Cout/256][Cin/16][256][3][16]; the pieces of every element sum to the element of the
    fp32 pack (rn_pack_weights, same double-precision transform) EXACTLY, each piece is a bf16 of the running remainder."""
    from rendernet_amd import ops
    from rendernet_amd import _lib as L
    sid, nxi, _, R = SCHEMES[which]
    rng = np.random.default_rng(cin * 7 + cout)
    w = _xavier(rng, (R, R, cout, cin) if transposed else (R, R, cin, cout))
    pw = ops.pack_conv_transpose(_dev(w), 1) if transposed else ops.pack_conv(_dev(w))
    u32 = (pw.wino63 if which == "f63" else pw.wino43).cpu().numpy().reshape(nxi, cout // 256, cin // 4, 256, 4)
    us = pw.split(which).cpu().numpy().view(np.uint16).reshape(nxi, cout // 256, cin // 16, 256, 48)
    planes = _rows_to_planes(us, (np.arange(256) >> 3) & 1)                       # [nxi, nb, s, 256, 3, 16]
    want = u32.reshape(nxi, cout // 256, cin // 16, 4, 256, 4).transpose(0, 1, 2, 4, 3, 5).reshape(nxi, cout // 256, cin // 16, 256, 16)
    p = _bf16_to_f64(planes)
    assert np.array_equal(p[..., 0, :] + p[..., 1, :] + p[..., 2, :], want.astype(np.float64))
    # piece 0 is the nearest bf16 of the value: |x - p0| <= half a bf16 ulp = 2^-9 |x| (and so on down the remainders)
    assert np.all(np.abs(want - p[..., 0, :]) <= 2.0 ** -8 * np.abs(want) + 1e-45)
    assert np.all(np.abs(want - p[..., 0, :] - p[..., 1, :]) <= 2.0 ** -16 * np.abs(want) + 1e-45)
    assert L.lib().rn_winograd_split_packed_bytes(sid, cin, cout) == us.size * 2


@pytest.mark.parametrize("which,B,H,W,C", [("f63", 2, 13, 16, 64), ("f43", 3, 9, 11, 96), ("f44", 1, 16, 16, 32), ("f63", 5, 64, 64, 128)])
def test_split_input_transform_is_the_fp32_transform_in_three_pieces(which, B, H, W, C):
    """rn_winograd_split_input_transform: Vs [nxi][C/16][T][3][16] holds V = B^T d B of rn_winograd_input_transform as three bf16
    pieces per element (exact sum; the two kernels may contract their FMAs differently: V itself within 2 ulp-ish)."""
    from rendernet_amd import _lib as L
    sid, nxi, m, _ = SCHEMES[which]
    lib = L.lib()
    rng = np.random.default_rng(H * 31 + W)
    x = _dev(rng.standard_normal((B, H, W, C)).astype(np.float32))
    T = B * (-(-H // m)) * (-(-W // m))
    V = torch.empty(nxi * T * C, device="cuda")
    L.check(lib.rn_winograd_input_transform(sid, L.ptr(x), L.ptr(V), B, H, W, C, 1, L.stream_ptr()), "input")
    nb = lib.rn_winograd_split_v_bytes(sid, T, C)
    assert nb >= nxi * T * C * 6
    Vs = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    L.check(lib.rn_winograd_split_input_transform(sid, L.ptr(x), ctypes.c_void_p(Vs.data_ptr()), B, H, W, C, 1, L.stream_ptr()), "split input")
    rows = Vs.cpu().numpy()[:nxi * T * C * 6].view(np.uint16).reshape(nxi, C // 16, T, 48)
    p = _bf16_to_f64(_rows_to_planes(rows, (np.arange(T) >> 3) & 1))            # [nxi, s, T, 3, 16]
    got = (p[..., 0, :] + p[..., 1, :] + p[..., 2, :]).transpose(0, 2, 1, 3).reshape(nxi, T, C)
    want = V.cpu().numpy().reshape(nxi, T, C).astype(np.float64)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    assert np.array_equal(got.astype(np.float32).astype(np.float64), got)        # the sum of the pieces IS an fp32 number
    # the channel contract is the GEMM stage's (Cin >= 32, Cin % 32 == 0; include/rendernet_hip.h): a count the GEMM would refuse is refused here
    assert lib.rn_winograd_split_v_bytes(sid, T, 48) == 0 and lib.rn_winograd_split_v_bytes(sid, T, 16) == 0
    x48 = torch.zeros((B, H, W, 48), device="cuda")
    assert lib.rn_winograd_split_input_transform(sid, L.ptr(x48), ctypes.c_void_p(Vs.data_ptr()), B, H, W, 48, 1, L.stream_ptr()) != 0


CASES = [   # (which, B, H, W, Cin, Cout): ragged planes, 1 .. many tiles (below / across / above the 256-row block, the 128-row
            # item plan, partial rounds), one K step pair .. 64, 1 .. 4 channel blocks
    ("f43", 1, 4, 4, 32, 256), ("f43", 2, 16, 16, 256, 256), ("f43", 3, 33, 5, 96, 256), ("f43", 1, 16, 16, 1024, 512),
    ("f43", 5, 30, 34, 128, 1024), ("f43", 2, 64, 64, 64, 256),
    ("f63", 1, 6, 6, 32, 256), ("f63", 1, 1, 1, 32, 256), ("f63", 1, 13, 16, 1024, 512), ("f63", 5, 30, 34, 128, 1024),
    ("f63", 2, 64, 64, 64, 256), ("f63", 3, 64, 64, 256, 512), ("f63", 6, 64, 64, 32, 1024), ("f63", 9, 64, 64, 64, 256),
    ("f44", 1, 5, 7, 32, 256), ("f44", 2, 16, 16, 256, 256), ("f44", 1, 64, 64, 64, 512), ("f44", 3, 33, 9, 128, 256),
]


@pytest.mark.parametrize("smode", ["split", "split16"])
@pytest.mark.parametrize("case", CASES)
def test_conv2d_split_vs_oracle(case, smode):
    """ops.conv2d under ops.gemm_mode("split") (three bf16 pieces, six products) and "split16" (two fp16 pieces of the scaled value,
    three products) vs the oracle conv (bias / PReLU / residual / pre-activation / sigmoid epilogues), vs the exact-fp32 route on
    the same filter, and the input gradient through the transposed pack."""
    from rendernet_amd import ops
    from rendernet_amd import _lib as L
    which, B, H, W, Cin, Cout = case
    fmt = L.RN_SPLIT_FMT_H2 if smode == "split16" else 0
    sid, nxi, m, R = SCHEMES[which]
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = _xavier(rng, (R, R, Cin, Cout))
    b = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    alpha = rng.uniform(0, 0.25, Cout).astype(np.float32)
    old_min = ops.WINO43_MIN_PIXELS
    ops.WINO43_MIN_PIXELS = 1
    try:
        y0 = OL.conv2d(x, w, b, (1, 1))
        res = rng.standard_normal(y0.shape).astype(np.float32)
        outs = {}
        for mode in ("f32", smode):
            pw = ops.pack_conv(_dev(w))
            if which != "f44":
                pw.force_scheme = which
            with torch.no_grad(), ops.gemm_mode(mode):
                outs["split" if mode == smode else mode] = ops.conv2d(_dev(x), pw, _dev(b), _dev(alpha), _dev(res))
                if mode == smode:
                    _close(ops.conv2d(_dev(x), pw, _dev(b)), y0, "split %s" % which)
                    _close(ops.conv2d(_dev(x), pw, None, sigmoid=True), torch.sigmoid(OL.conv2d(x, w, None, (1, 1))), "split+sigmoid")
        _close(outs["split"], OL.prelu(y0, alpha) + torch.from_numpy(res), "split %s +prelu+res" % which)
        assert float((outs["split"] - outs["f32"]).abs().max()) <= 1e-4 * float(outs["f32"].abs().max())
        assert not torch.equal(outs["split"], outs["f32"]) or Cin * H * W < 64       # (a different summation, not the same bits)
        # the C entry with the pre-activation output
        lib = L.lib()
        ws = torch.empty(lib.rn_winograd_split_workspace_bytes(sid | fmt, B, H, W, Cin, Cout), dtype=torch.uint8, device="cuda")
        yy, zz = torch.empty(y0.shape, device="cuda"), torch.empty(y0.shape, device="cuda")
        xd, bd, ad = _dev(x), _dev(b), _dev(alpha)
        L.check(lib.rn_conv2d_winograd_split_fwd(sid | fmt, L.ptr(xd), ctypes.c_void_p(pw.split(which, fmt).data_ptr()), L.ptr(bd),
                                                 L.ptr(ad), None, L.ptr(yy), L.ptr(zz), ctypes.c_void_p(ws.data_ptr()),
                                                 B, H, W, Cin, Cout, 0, 1, L.stream_ptr()), "rn_conv2d_winograd_split_fwd")
        _close(zz, y0, "split preact")
        _close(yy, OL.prelu(y0, alpha), "split prelu")
        if Cin % 256 == 0:                               # the conv's input gradient: the transposed pack of the same filter
            dp = pw.dgrad_pack(True)
            dz = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
            dx = torch.empty((B, H, W, Cin), device="cuda")
            if which != "f44":
                dp.force_scheme = which
            dzd = _dev(dz)
            with ops.gemm_mode(smode):
                L.check(ops._wino43_fwd(dzd, dp, (None, None, None, L.ptr(dx), None), B, H, W, Cout, Cin, 0), "split dgrad")
            _close(dx, OL.conv2d_transpose(dz, w, None, (1, 1)), "split dgrad vs oracle")
    finally:
        ops.WINO43_MIN_PIXELS = old_min


@pytest.mark.parametrize("smode", ["split", "split16"])
def test_conv2d_transpose_s1_split(smode):
    """e_conv7_1-like stride-1 transposed 4x4 conv (slim.conv2d_transpose, RenderNet_Shader.py:109-111) through F(4x4,4x4) on
    the split stage (pad two before)."""
    from rendernet_amd import ops
    rng = np.random.default_rng(5)
    B, H, W, Cin, Cout = 2, 16, 18, 256, 256
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    wt = _xavier(rng, (4, 4, Cout, Cin))
    b = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    old_min = ops.WINO43_MIN_PIXELS
    ops.WINO43_MIN_PIXELS = 1
    try:
        pt = ops.pack_conv_transpose(_dev(wt), 1)
        assert pt.split("f44") is not None
        with torch.no_grad(), ops.gemm_mode(smode):
            got = ops.conv2d_transpose(_dev(x), pt, _dev(b), None, None, (1, 1))
        _close(got, OL.conv2d_transpose(x, wt, b, (1, 1)), "split convT s1")
    finally:
        ops.WINO43_MIN_PIXELS = old_min


@pytest.mark.parametrize("smode", ["split", "split16"])
def test_bench_frames_match_golden_split(fixtures_vox, smode):
    """The benched configuration with the split stage (either operand format): the five golden frames of bench.py's batch (tests/golden/bench_frames.npz,
    oracle output) at the bars of tests/test_gpu_net.py::test_bench_frames_match_golden -- sampled taps <= 2e-4 * max, image
    <= 1e-3, logits <= 5e-4 * max."""
    from rendernet_amd import ops
    from rendernet_amd.shader import Renderer, ShaderSpec, init_shader_weights
    from bench import synthetic_batch
    g = np.load(os.path.join(GOLDEN_DIR, "bench_frames.npz"))
    idx = [int(i) for i in g["frames"]]
    vox, poses = synthetic_batch(24)
    spec = ShaderSpec().check()
    r = Renderer(spec, init_shader_weights(spec, seed=1234, perturb=True), gemm=smode)       # the mode is the renderer's own
    taps = {}
    out = r.render(vox[idx], poses[idx], taps=taps).cpu().numpy()
    for k in range(5):
        e4 = taps["enc4"][k, 3::8, 5::8, :].cpu().numpy()
        assert np.abs(e4 - g["enc4_%d" % k]).max() <= 2e-4 * np.abs(g["enc4_%d" % k]).max() + 1e-6
        for c, (r0, c0) in enumerate(g["crops"]):
            crop = out[k, r0:r0 + 128, c0:c0 + 128, 0]
            assert np.abs(crop - g["output_%d" % k][c]).max() <= 1e-3, (k, c)
            lg = np.log(crop.astype(np.float64) / (1 - crop.astype(np.float64)))
            want = g["logits_%d" % k][c]
            assert np.abs(lg - want).max() <= 5e-4 * np.abs(want).max() + 1e-5, (k, c)


# ---------------------------------------------------------------------------------------------------------------------
# The fused 3x3x3 32 -> 32 kernel in bf16x3 (csrc/conv3d_wino_bf3.hip; on in the "split" modes of ops.gemm_mode, ops.CONV3D_SPLIT forces it).  The layers:
# res_block_3d's two slim.conv3d [3,3,3] 32 -> 32 (tools/layer_util.py:60-75, RenderNet_Shader.py:61-68).
C3_CASES = [
    (1, 4, 32, 2),       # one item, D = 2: both slices have a padded neighbour
    (1, 8, 32, 3),       # one full three-step turn of the accumulator ring
    (2, 16, 64, 5),      # two column blocks, D % 3 == 2
    (3, 10, 20, 4),      # ragged rows of tiles in W, D % 3 == 1
    (1, 5, 7, 1),        # a single depth slice, odd H, W < 32
    (1, 64, 64, 32),     # the res1 layer of the benched net
    (2, 3, 70, 7),       # three column blocks, the last one 6 columns wide
]


def test_conv3d_split_pack_is_the_fp32_filter_transform_in_three_pieces():
    """rn_conv3d_winograd_split_pack: [i][j][depth tap][piece][channel tile][lane = n % 16 + 16 (c / 8)][c % 8] holds
    U = G g G^T over (k1, k2) of each depth tap, the three bf16 pieces summing EXACTLY to the fp32 value."""
    from rendernet_amd import _lib as L, ops
    rng = np.random.default_rng(5)
    w = _xavier(rng, (3, 3, 3, 32, 32))
    pw = ops.pack_conv(_dev(w))
    us = pw.split3d()
    assert us is not None and us.numel() == L.lib().rn_conv3d_winograd_split_packed_bytes(32, 32)
    torch.cuda.synchronize()
    u16 = us.cpu().numpy().view(np.uint16).reshape(4, 4, 3, 3, 2, 4, 16, 8)           # [i][j][dz][piece][nt][c/8][n%16][c%8]
    p = _bf16_to_f64(u16)
    got = (p[:, :, :, 0] + p[:, :, :, 1] + p[:, :, :, 2])                               # [i][j][dz][nt][c/8][n%16][c%8]
    got = got.transpose(0, 1, 2, 4, 6, 3, 5).reshape(4, 4, 3, 32, 32)                   # [i][j][dz][c][n]
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    want = np.einsum("ip,jq,pqdcn->ijdcn", G, G, w.astype(np.float64)).astype(np.float32)
    assert np.abs(got - want).max() <= 2.0 ** -22 * np.abs(want).max()
    assert np.all(got == got.astype(np.float32))                                         # an fp32 value, exactly
    assert np.all(np.abs(got - p[:, :, :, 0].transpose(0, 1, 2, 4, 6, 3, 5).reshape(got.shape)) <= 2.0 ** -8 * np.abs(got) + 1e-45)


@pytest.mark.parametrize("smode", ["split", "split16"])
@pytest.mark.parametrize("case", C3_CASES)
def test_conv3d_split(case, smode, monkeypatch):
    """Forward with every epilogue, against the oracle conv (oracle/layers.py) at the bar of the fp32 kernel, and the input
    gradient (the flipped, channel-swapped filter through the same kernel) against the oracle's transposed conv; both operand
    formats (split16: max|x| by a pass here -- the chained hand-over is what the bench-frame and training tests exercise)."""
    from rendernet_amd import _lib as L, ops
    monkeypatch.setattr(ops, "CONV3D_SPLIT", True)
    monkeypatch.setattr(ops._MODE, "mode", smode, raising=False)
    fmt = 1 if smode == "split16" else 0
    B, H, W, D = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rng.standard_normal((B, H, W, D, 32)).astype(np.float32)
    w = _xavier(rng, (3, 3, 3, 32, 32))
    b = (0.1 * rng.standard_normal(32)).astype(np.float32)
    alpha = rng.uniform(0, 0.25, 32).astype(np.float32)
    pw = ops.pack_conv(_dev(w))
    assert pw.split3d(fmt) is not None
    y0 = OL.conv3d(x, w, b, (1, 1, 1))
    _close(ops.conv3d(_dev(x), pw, _dev(b)), y0, "split 3-D conv", rtol=2e-6)
    _close(ops.conv3d(_dev(x), pw, None), OL.conv3d(x, w, None, (1, 1, 1)), "split 3-D conv, no bias", rtol=2e-6)
    res = rng.standard_normal(y0.shape).astype(np.float32)
    _close(ops.conv3d(_dev(x), pw, _dev(b), _dev(alpha), _dev(res)), OL.prelu(y0, alpha) + torch.from_numpy(res), "split 3-D conv + PReLU + residual",
           rtol=2e-6)
    # the pre-activation tap of the training forward
    xd, bd, ad = _dev(x), _dev(b), _dev(alpha)
    y, z = torch.empty_like(xd), torch.empty_like(xd)
    words = torch.zeros(2, dtype=torch.int32, device="cuda")
    L.check(L.lib().rn_conv3d_winograd_split_fwd_ex(fmt, L.ptr(xd), ctypes.c_void_p(pw.split3d(fmt).data_ptr()), L.ptr(bd), L.ptr(ad), None, L.ptr(y), L.ptr(z),
                                                    B, H, W, D, 32, 32, L.RN_ACT_PRELU, None, ctypes.c_void_p(words.data_ptr() + 4),
                                                    ctypes.c_void_p(words.data_ptr()), L.stream_ptr()), "rn_conv3d_winograd_split_fwd_ex")
    # max|y| gathered by the launch (for the next layer's scale) and, in format H2, max|x| from the launcher's pass
    assert words[:1].view(torch.float32).item() == float(y.abs().max())
    if fmt:
        assert words[1:].view(torch.float32).item() == float(xd.abs().max())
    _close(z, y0, "split 3-D pre-activation", rtol=2e-6)
    _close(y, OL.prelu(y0, alpha), "split 3-D PReLU", rtol=2e-6)
    # input gradient
    dp = pw.dgrad_pack(True)
    assert dp.split3d(fmt) is not None
    dz = _dev(rng.standard_normal((B, H, W, D, 32)).astype(np.float32))
    dx = torch.empty_like(dz)
    L.check(L.lib().rn_conv3d_winograd_split_fwd_ex(fmt, L.ptr(dz), ctypes.c_void_p(dp.split3d(fmt).data_ptr()), None, None, None, L.ptr(dx), None,
                                                    B, H, W, D, 32, 32, 0, None, ctypes.c_void_p(words.data_ptr() + 4), None, L.stream_ptr()),
            "rn_conv3d_winograd_split_fwd_ex (dgrad)")
    _close(dx, OL.conv3d_transpose(dz.cpu().numpy(), w, None, (1, 1, 1)), "split 3-D dgrad", rtol=2e-6)


def test_conv3d_split_depth_segments_are_bit_identical(tmp_path):
    """csrc/conv3d_wino_bf3.hip: c3_depth_segments -- with few images a row of tiles is cut into depth segments so that the 256 persistent
    workgroups all have work (batch 1..3: 4 segments, batch 6: 2).  A segment re-walks one halo slice per end; every OUTPUT slice still
    sums the same taps in the same order, so any segment count gives the bits of the whole depth run -- forward with residual and PReLU
    and the pre-activation tap, depths that the count does not divide, a single slice.  RN_C3_DEPTH_SEGMENTS is read once per process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, ctypes, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from rendernet_amd import ops, _lib as L\n"
        "ops.WINO_GEMM = sys.argv[2]\n"
        "fmt = 1 if sys.argv[2] == 'split16' else 0\n"
        "rng = np.random.default_rng(11)\n"
        "outs = []\n"
        "for (B, H, W, D) in ((1, 8, 40, 13), (2, 5, 33, 7), (1, 4, 4, 1), (3, 16, 16, 64), (1, 64, 64, 64)):\n"
        "    x = torch.as_tensor(rng.standard_normal((B, H, W, D, 32)).astype(np.float32)).cuda()\n"
        "    w = torch.as_tensor((0.06 * rng.standard_normal((3, 3, 3, 32, 32))).astype(np.float32)).cuda()\n"
        "    b = torch.as_tensor((0.1 * rng.standard_normal(32)).astype(np.float32)).cuda()\n"
        "    a = torch.as_tensor(rng.uniform(0, 0.25, 32).astype(np.float32)).cuda()\n"
        "    r = torch.as_tensor(rng.standard_normal((B, H, W, D, 32)).astype(np.float32)).cuda()\n"
        "    pw = ops.pack_conv(w)\n"
        "    y, z = torch.empty_like(x), torch.empty_like(x)\n"
        "    words = torch.zeros(2, dtype=torch.int32, device='cuda')\n"
        "    L.check(L.lib().rn_conv3d_winograd_split_fwd_ex(fmt, L.ptr(x), ctypes.c_void_p(pw.split3d(fmt).data_ptr()), L.ptr(b), L.ptr(a), L.ptr(r), L.ptr(y), L.ptr(z),\n"
        "            B, H, W, D, 32, 32, L.RN_ACT_PRELU, None, ctypes.c_void_p(words.data_ptr() + 4), ctypes.c_void_p(words.data_ptr()), L.stream_ptr()), 'fwd')\n"
        "    torch.cuda.synchronize()\n"
        "    assert words[:1].view(torch.float32).item() == float(y.abs().max())\n"
        "    outs += [y.cpu().numpy(), z.cpu().numpy()]\n"
        "np.savez(sys.argv[1], *outs)\n" % root)
    for smode in ("split", "split16"):
        got = {}
        for nseg in ("1", "2", "3", "4", "8", ""):                      # "": the launcher's own choice
            out = str(tmp_path / ("seg_%s_%s.npz" % (smode, nseg or "auto")))
            env = dict(os.environ)
            env.pop("RN_C3_DEPTH_SEGMENTS", None)
            if nseg:
                env["RN_C3_DEPTH_SEGMENTS"] = nseg
            r = subprocess.run([sys.executable, "-c", code, out, smode], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            got[nseg] = np.load(out)
        for nseg in got:
            for k in got["1"].files:
                assert np.array_equal(got["1"][k], got[nseg][k]), (smode, nseg, k)
        assert all(np.isfinite(got["1"][k]).all() and np.abs(got["1"][k]).max() > 0 for k in got["1"].files)


class _TrainStub:
    """What ops._Conv needs of a training context (rendernet_amd/train.py: Trainer): an autograd anchor, gradient views, `ready`."""
    frozen = False

    def __init__(self, *params):
        self.anchor = torch.zeros(1, device="cuda", requires_grad=True)
        self.g = {id(p): torch.zeros_like(p) for p in params}

    def grad(self, t):
        return self.g.get(id(t))

    grad_if_param = grad

    def ready(self, *ts):
        pass


def test_four_wave_gemm_variant_is_bit_identical(tmp_path):
    """RN_WINO_BF3_W4=1: whole 256-row items of the bf16x3 GEMM stage on the four-wave kernel (128 x 128 wave tiles, one wave per SIMD;
    csrc/conv_wino_bf3.hip: wino_gemm_bf3_w4_kernel).  Every output element sums the same piece products over the same K steps in the same
    order as on the eight-wave kernel, so the two builds of the stage must agree BIT FOR BIT (the variant exists for the measurement that
    showed the stage to be power-bound: same time on both kernels, profiles/r05_gemm_w4_ab.txt).  The switch is read once per process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from rendernet_amd import ops\n"
        "ops.WINO_GEMM = 'split'\n"
        "rng = np.random.default_rng(3)\n"
        "outs = []\n"
        "for (B, hw, cin, cout, k) in ((5, 64, 64, 256, 3), (9, 64, 96, 512, 3), (4, 64, 64, 256, 4)):\n"      # T = 605 / 1089 / 1024: whole blocks + ragged rows
        "    x = torch.as_tensor(rng.standard_normal((B, hw, hw, cin)).astype(np.float32)).cuda()\n"
        "    w = torch.as_tensor((0.05 * rng.standard_normal((k, k, cin, cout))).astype(np.float32)).cuda()\n"
        "    with torch.no_grad():\n"
        "        outs.append(ops.conv2d(x, ops.pack_conv(w)).cpu().numpy())\n"
        "np.savez(sys.argv[1], *outs)\n" % root)
    got = {}
    for w4 in ("0", "1"):
        out = str(tmp_path / ("w4_%s.npz" % w4))
        # (both on v_mfma_f32_32x32x16_bf16: the four-wave variant exists in that form only)
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, RN_WINO_BF3_W4=w4, RN_WINO_BF3_P16="0"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[w4] = np.load(out)
    for k in got["0"].files:
        assert np.array_equal(got["0"][k], got["1"][k]), k
        assert np.isfinite(got["1"][k]).all() and np.abs(got["1"][k]).max() > 0
    # The product form (round 6): the same six piece products per element on v_mfma_f32_16x16x32_bf16, two per instruction (K = 16 channels x
    # two pieces).  Same terms, another grouping of the fp32 sums -> not the same bits; both sit within the split stage's bar of the oracle
    # (every other test of this file runs the product form), and of each other within a few fp32 roundings of the K-long sum:
    # |a - b| <= 8 * 2^-24 * sum_k |x_k u_k| per Winograd product, here bounded through max|y| with the measured ratio (< 3e-6 max|y|).
    out = str(tmp_path / "p16.npz")
    r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, RN_WINO_BF3_P16="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    p16 = np.load(out)
    for k in got["0"].files:
        ref = got["0"][k]
        assert np.abs(p16[k] - ref).max() <= 1e-5 * np.abs(ref).max(), (k, np.abs(p16[k] - ref).max(), np.abs(ref).max())
        assert not np.array_equal(p16[k], ref), "RN_WINO_BF3_P16 did not select another kernel"


def test_split_gemm_launch_plan_never_changes_a_bit(tmp_path):
    """The split GEMM stage picks whole items, half items, a merged or a separate ragged block and one or several launches by a cost model
    (csrc/conv_wino_bf3.hip: gemm_split_planes; RN_WINO_BF3_NOTAIL / _NOMERGE / _HALF_COST move its decisions).  Whatever it picks, every
    output element sums the same piece products over the same K steps in the same order: the plans must agree BIT FOR BIT -- on shapes
    with whole blocks + ragged rows, a partial last round, few tiles (the small-batch branch) and a 512-channel layer."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from rendernet_amd import ops\n"
        "rng = np.random.default_rng(5)\n"
        "outs = []\n"
        "for (B, hw, cin, cout, k) in ((3, 64, 64, 1024, 3), (4, 64, 32, 1024, 3), (5, 64, 64, 512, 3), (9, 64, 32, 256, 3), (2, 32, 64, 512, 4), (6, 64, 32, 1024, 3)):\n"
        "    x = torch.as_tensor(rng.standard_normal((B, hw, hw, cin)).astype(np.float32)).cuda()\n"
        "    w = torch.as_tensor((0.05 * rng.standard_normal((k, k, cin, cout))).astype(np.float32)).cuda()\n"
        "    with torch.no_grad(), ops.gemm_mode('split'):\n"
        "        outs.append(ops.conv2d(x, ops.pack_conv(w)).cpu().numpy())\n"
        "np.savez(sys.argv[1], *outs)\n" % root)
    plans = {"default": {}, "notail": {"RN_WINO_BF3_NOTAIL": "1"}, "nomerge": {"RN_WINO_BF3_NOMERGE": "1"},
             "cheap_halves": {"RN_WINO_BF3_HALF_COST": "1"}, "dear_halves": {"RN_WINO_BF3_HALF_COST": "40"}}
    got = {}
    for name, env in plans.items():
        out = str(tmp_path / ("%s.npz" % name))
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[name] = np.load(out)
    for name in plans:
        for k in got["default"].files:
            assert np.array_equal(got["default"][k], got[name][k]), (name, k)
    assert all(np.isfinite(got["default"][k]).all() and np.abs(got["default"][k]).max() > 0 for k in got["default"].files)


@pytest.mark.parametrize("cin,cout,transposed", [(64, 256, 0), (1024, 512, 0), (256, 256, 1)])
def test_1x1_split_pack_and_row_format_are_the_fp32_values_in_three_pieces(cin, cout, transposed):
    """RN_WINO_F11 layouts stated in NumPy: the pack Us [1][Cout/256][Cin/16][256][3][16] holds w[c][n] (transposed: w[n][c] of the TF
    conv_transpose layout) and the input "transform" Vs [1][C/16][T][3][16] holds x itself, each element as three bf16 pieces whose sum is
    EXACTLY the fp32 value (the identity scheme adds no arithmetic)."""
    from rendernet_amd import ops
    from rendernet_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(cin + cout)
    w = _xavier(rng, (1, 1, cout, cin) if transposed else (1, 1, cin, cout))
    pw = ops.pack_conv_transpose(_dev(w), 1) if transposed else ops.pack_conv(_dev(w))
    us = pw.split("f11").cpu().numpy().view(np.uint16).reshape(1, cout // 256, cin // 16, 256, 48)
    p = _bf16_to_f64(_rows_to_planes(us, (np.arange(256) >> 3) & 1))                      # [1, nb, s, 256, 3, 16]
    got = (p[..., 0, :] + p[..., 1, :] + p[..., 2, :])[0]                                  # [nb, s, 256 (n), 16 (c)]
    wm = (w[0, 0].T if transposed else w[0, 0]).astype(np.float64)                         # [cin, cout]
    want = wm.reshape(cin // 16, 16, cout // 256, 256).transpose(2, 0, 3, 1)
    assert np.array_equal(got, want)
    assert lib.rn_winograd_split_packed_bytes(L.RN_WINO_F11, cin, cout) == us.size * 2
    B, H, W = 2, 5, 7                                                                       # T = 70: below the 8-pixel group boundary at the end
    x = rng.standard_normal((B, H, W, cin)).astype(np.float32)
    T = B * H * W
    nb = lib.rn_winograd_split_v_bytes(L.RN_WINO_F11, T, cin)
    assert nb >= T * cin * 6
    Vs = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    L.check(lib.rn_winograd_split_input_transform(L.RN_WINO_F11, L.ptr(_dev(x)), ctypes.c_void_p(Vs.data_ptr()), B, H, W, cin, 0, L.stream_ptr()), "split input (1x1)")
    rows = Vs.cpu().numpy()[:T * cin * 6].view(np.uint16).reshape(1, cin // 16, T, 48)
    pv = _bf16_to_f64(_rows_to_planes(rows, (np.arange(T) >> 3) & 1))
    gotx = (pv[..., 0, :] + pv[..., 1, :] + pv[..., 2, :])[0].transpose(1, 0, 2).reshape(T, cin)
    assert np.array_equal(gotx, x.reshape(T, cin).astype(np.float64))


CASES_1X1 = [   # (B, H, W, Cin, Cout): pixel counts below / across / above the 256-row block and the 8-pixel transform group, 2 .. 64 K steps
    (1, 3, 5, 32, 256), (1, 16, 16, 64, 256), (2, 17, 15, 1024, 512), (3, 64, 64, 256, 256), (1, 64, 64, 1024, 1024), (5, 9, 7, 96, 768),
]


@pytest.mark.parametrize("smode", ["split", "split16"])
@pytest.mark.parametrize("case", CASES_1X1)
def test_conv2d_1x1_split_vs_oracle(case, smode, monkeypatch):
    """A 1x1 filter on the split multiply stage (scheme RN_WINO_F11: split x into the GEMM's rows / ONE T x Cin x Cout GEMM / epilogue) -- the
    projection unit's conv (tools/layer_util.py:8-22: slim.conv2d [1,1] + prelu) and its input gradient: the C entry with every epilogue
    flavour and the pre-activation output, ops.conv2d's routing, the exact-fp32 kernel on the same filter, all at the 1e-4 * max|ref| bar of
    every other conv flavour."""
    from rendernet_amd import ops
    from rendernet_amd import _lib as L
    B, H, W, Cin, Cout = case
    fmt = L.RN_SPLIT_FMT_H2 if smode == "split16" else 0
    sid = L.RN_WINO_F11
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = _xavier(rng, (1, 1, Cin, Cout))
    b = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    alpha = rng.uniform(0, 0.25, Cout).astype(np.float32)
    y0 = OL.conv2d(x, w, b, (1, 1))
    res = rng.standard_normal(y0.shape).astype(np.float32)
    lib = L.lib()
    assert lib.rn_winograd_split_supported(sid | fmt, Cin, Cout) == 1 and lib.rn_conv2d_wino43_supported(Cin, Cout) in (0, 1)
    monkeypatch.setattr(ops, "SPLIT11_MIN_PIXELS", 1)
    monkeypatch.setattr(ops._MODE, "mode", smode, raising=False)
    pw = ops.pack_conv(_dev(w))
    assert pw._split11 and pw.split("f11", fmt) is not None
    xd, bd, ad, rd = _dev(x), _dev(b), _dev(alpha), _dev(res)
    ws = torch.empty(lib.rn_winograd_split_workspace_bytes(sid | fmt, B, H, W, Cin, Cout), dtype=torch.uint8, device="cuda")
    yy, zz = torch.empty(y0.shape, device="cuda"), torch.empty(y0.shape, device="cuda")
    us = ctypes.c_void_p(pw.split("f11", fmt).data_ptr())
    L.check(lib.rn_conv2d_winograd_split_fwd(sid | fmt, L.ptr(xd), us, L.ptr(bd), L.ptr(ad), L.ptr(rd), L.ptr(yy), L.ptr(zz),
                                             ctypes.c_void_p(ws.data_ptr()), B, H, W, Cin, Cout, 0, 1, L.stream_ptr()), "rn_conv2d_winograd_split_fwd (1x1)")
    _close(zz, y0, "1x1 split preact")
    _close(yy, OL.prelu(y0, alpha) + torch.from_numpy(res), "1x1 split prelu + residual")
    with torch.no_grad():
        got = ops.conv2d(xd, pw, bd, ad, rd)                                  # the dispatcher takes the same route ...
        assert torch.equal(got, yy)
        _close(ops.conv2d(xd, pw, None, sigmoid=True), torch.sigmoid(OL.conv2d(x, w, None, (1, 1))), "1x1 split + sigmoid")
        monkeypatch.setattr(ops._MODE, "mode", "f32", raising=False)                          # ... and the exact mode the implicit-GEMM kernel
        exact = ops.conv2d(xd, pw, bd, ad, rd)
        monkeypatch.setattr(ops._MODE, "mode", smode, raising=False)
    assert float((got - exact).abs().max()) <= 1e-4 * float(exact.abs().max()) and not torch.equal(got, exact)
    # the input gradient: the same TF tensor packed the other way round (needs Cin % 256 == 0: the channel roles swap)
    if Cin % 256 == 0:
        dp = pw.dgrad_pack(True)
        assert dp._split11
        dz = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
        dx = torch.empty((B, H, W, Cin), device="cuda")
        L.check(ops._wino43_run(_dev(dz), dp, (None, None, None, L.ptr(dx), None), B, H, W, Cout, Cin, 0, "f11"), "1x1 split dgrad")
        _close(dx, OL.conv2d_transpose(dz, w, None, (1, 1)), "1x1 split dgrad vs oracle")
    # the exact-fp32 stage entries do not know the scheme
    assert lib.rn_winograd_input_transform(sid, L.ptr(xd), L.ptr(yy), B, H, W, Cin, 0, L.stream_ptr()) != 0


@pytest.mark.parametrize("smode", ["split", "split16"])
def test_projection_unit_on_the_split_stage(smode, monkeypatch):
    """ops.projection (depth-flatten + 1x1 conv + PReLU, tools/layer_util.py:8-22) at the benched width: in the split modes the GEMM runs on
    the split stage (forward, and under a training context forward + input gradient), against the oracle's projection unit and against the
    exact-fp32 kernel."""
    from rendernet_amd import ops
    rng = np.random.default_rng(17)
    B, H, W, D, C = 2, 64, 64, 32, 32
    x = rng.standard_normal((B, H, W, D, C)).astype(np.float32)
    w = _xavier(rng, (1, 1, D * C, D * C))
    b = (0.1 * rng.standard_normal(D * C)).astype(np.float32)
    al = rng.uniform(0, 0.25, D * C).astype(np.float32)
    want = OL.prelu(OL.conv2d(x.reshape(B, H, W, D * C), w, b, (1, 1)), al)
    pw = ops.pack_conv(_dev(w))
    out = {}
    for mode in ("f32", smode):
        monkeypatch.setattr(ops._MODE, "mode", mode, raising=False)
        with torch.no_grad():
            out[mode] = ops.projection(_dev(x), pw, _dev(b), _dev(al))
        _close(out[mode], want, "projection unit, %s" % mode)
    assert not torch.equal(out["f32"], out[smode])                           # two routes, not one
    # under autograd: forward + input gradient on the split stage (the filter gradient stays on the shared exact kernel).  The gradients are
    # compared on the LINEAR unit (no PReLU): among 8 M pre-activations a few lie within rounding of zero and take the other PReLU branch on the
    # two sides, which moves single entries of dx by a whole filter coefficient -- the epilogue backward has its own tests (test_gpu_train.py)
    monkeypatch.setattr(ops._MODE, "mode", smode, raising=False)
    xd, wd, ad, bd = _dev(x).requires_grad_(True), pw.w_tf, _dev(al), _dev(b)
    tc = _TrainStub(wd, ad, bd)
    monkeypatch.setattr(ops, "TRAIN", tc)
    g = rng.standard_normal((B, H, W, D * C)).astype(np.float32)
    _close(ops.projection(xd, pw, bd, ad), want, "projection under autograd (forward, with the saved pre-activation)")
    y = ops.conv2d(xd.view(B, H, W, D * C), pw, bd)
    y.backward(_dev(g))
    xr, wr = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(w).requires_grad_(True)
    yr = OL.conv2d(xr.reshape(B, H, W, D * C), wr, torch.from_numpy(b), (1, 1))
    yr.backward(torch.from_numpy(g))
    _close(y, yr, "1x1 conv under autograd")
    _close(xd.grad, xr.grad, "1x1 conv input gradient (split stage)", rtol=2e-4)
    _close(tc.g[id(wd)], wr.grad, "1x1 conv filter gradient", rtol=2e-4)


def _wgrad3d_split_partials(B, H, W, D):
    """Workgroups that add into one entry of dw in ONE rn_conv3d_wgrad_split launch -- the launchers' own split of the position list
    (csrc/conv_wgrad.hip launch_wgrad_k3d32_row: items of 4 columns x 16 depth positions, >= 16 items per workgroup, at most 1024 workgroups
    per filter row; launch_wgrad_k3d32 when W % 4 != 0: items of 4 columns x 32 positions, >= 4 per workgroup, at most 342 per tap pair)."""
    if W % 4 == 0:
        nitems, cap, least = (B * H * W // 4) * ((D + 15) // 16), (3072 + 2) // 3, 16
    else:
        nitems, cap, least = ((B * H * W + 3) // 4) * ((D + 31) // 32), (3072 + 8) // 9, 4
    ns = max(1, min(cap, (nitems + least - 1) // least))
    ipw = (nitems + ns - 1) // ns
    return (nitems + ipw - 1) // ipw


@pytest.mark.parametrize("shape", [(2, 12, 40, 6), (1, 8, 8, 33), (3, 16, 16, 16), (1, 5, 7, 1), (24, 32, 32, 16)])
def test_conv3d_wgrad_split(shape):
    """rn_conv3d_wgrad_split: the filter gradient of the 3-D encoder's 3x3x3 32 -> 32 convs (tf.nn.conv3d_backprop_filter_v2 of slim.conv3d,
    tools/layer_util.py:60-73) with the reduction over the positions on the bf16 pipe -- against torch-CPU autograd over the oracle conv
    (1e-4 * max|ref|, the bar of the exact kernel), against the exact-fp32 kernel (3e-5 * max: two fp32-class sums of the same terms), and
    ACCUMULATING into dw (two calls = twice the gradient).  Depths that are not multiples of the 16-position block / 32-position stage, one
    position, the benched shape (crop 64: B = 24, 32 x 32 x 16)."""
    from rendernet_amd import _lib as L
    B, H, W, D = shape
    rng = np.random.default_rng(B * 131 + D)
    x = rng.standard_normal((B, H, W, D, 32)).astype(np.float32)
    dz = rng.standard_normal((B, H, W, D, 32)).astype(np.float32)
    lib = L.lib()
    assert lib.rn_conv3d_wgrad_split_supported(32, 32) == 1 and lib.rn_conv3d_wgrad_split_supported(16, 16) == 0
    xd, dzd = _dev(x), _dev(dz)
    dws = torch.zeros((3, 3, 3, 32, 32), device="cuda")
    dwe = torch.zeros_like(dws)
    L.check(lib.rn_conv3d_wgrad_split(L.ptr(xd), L.ptr(dzd), L.ptr(dws), B, H, W, D, 32, 32, L.stream_ptr()), "rn_conv3d_wgrad_split")
    L.check(lib.rn_conv3d_wgrad(L.ptr(xd), L.ptr(dzd), L.ptr(dwe), B, H, W, D, 32, 32, L.ivec((3, 3, 3)), L.ivec((1, 1, 1)), L.stream_ptr()), "rn_conv3d_wgrad")
    ref = float(dwe.abs().max())
    assert float((dws - dwe).abs().max()) <= 3e-5 * ref, (float((dws - dwe).abs().max()), ref)
    # the oracle at EVERY shape, the benched one included (torch-CPU autograd over the oracle conv: 1.3 s at 393 216 positions)
    w = torch.zeros((3, 3, 3, 32, 32), requires_grad=True)
    OL.conv3d(torch.from_numpy(x), w, None, (1, 1, 1)).backward(torch.from_numpy(dz))
    _close(dws, w.grad, "split 3-D filter gradient vs oracle autograd")
    once = dws.clone()
    L.check(lib.rn_conv3d_wgrad_split(L.ptr(xd), L.ptr(dzd), L.ptr(dws), B, H, W, D, 32, 32, L.stream_ptr()), "rn_conv3d_wgrad_split")
    # Accumulation.  Every workgroup's partial sum is deterministic, but the P workgroups that share a filter entry add theirs with fp32
    # atomics in whatever order they finish, so the bound is DERIVED from the reduction instead of observed: the two launches perform 2 P
    # atomic adds per entry; an fp32 add rounds by at most 2^-24 of its result; the running value of an entry stays below R = 4 max|dw|
    # (twice the final 2 max|dw|: the partials of one entry are zero-mean sums of equally many products, their prefix sums wander by
    # ~|dw|, not by multiples of it).  Worst case, all roundings in one direction: |dw(2 calls) - 2 dw(1 call)| <= 2 P 2^-24 R
    # = 8 P 2^-24 max|dw| -- 1.8e-4 max|dw| at the benched shape (P = 384).  The observed value is the random walk of those roundings,
    # ~sqrt(2 P) 2^-25 R = 2e-6 max|dw|: round 5's 2e-6 sat INSIDE it (2.15e-6 on the driver box).  A launch that overwrote dw, or dropped
    # a workgroup's share, is wrong by ~max|dw| / sqrt(P) or more: two orders above the bound.
    P = _wgrad3d_split_partials(B, H, W, D)
    bound = 8.0 * P * 2.0 ** -24 * ref
    got = float((dws - 2 * once).abs().max())
    assert got <= bound, (got, bound, P)
    assert lib.rn_conv3d_wgrad_split(L.ptr(xd), L.ptr(dzd), L.ptr(dws), B, H, W, D, 16, 32, L.stream_ptr()) != 0


def test_conv3d_split_through_autograd_matches_the_fp32_kernel(monkeypatch):
    """ops.conv3d under autograd (forward with the saved pre-activation, input gradient through the split kernel, filter gradient
    through the shared wgrad kernel) with the opt-in on, against the same layer with it off."""
    from rendernet_amd import ops
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 12, 40, 6, 32)).astype(np.float32)
    w = _xavier(rng, (3, 3, 3, 32, 32))
    al = rng.uniform(0, 0.25, 32).astype(np.float32)
    g = rng.standard_normal(x.shape).astype(np.float32)
    out = {}
    for on in (False, True):
        monkeypatch.setattr(ops, "CONV3D_SPLIT", on)
        xd, wd, ad, bd = _dev(x).requires_grad_(True), _dev(w), _dev(al), torch.zeros(32, device="cuda")
        tc = _TrainStub(wd, ad, bd)
        monkeypatch.setattr(ops, "TRAIN", tc)
        y = ops.conv3d(xd, ops.pack_conv(wd), bd, ad)
        y.backward(_dev(g))
        torch.cuda.synchronize()
        out[on] = (y.detach(), xd.grad, tc.grad(wd), tc.grad(ad))
    monkeypatch.setattr(ops, "TRAIN", None)
    # y and dx are deterministic on both sides (3e-6: the split products' own error); dw and dalpha are sums of ~6 000 terms per entry added
    # with fp32 atomics in finishing order on BOTH sides (~sqrt(n) 2^-24 ~ 5e-7 of an entry per side: 1e-5 leaves an order of magnitude)
    for a_, b_, what, tol in zip(out[False], out[True], ("y", "dx", "dw", "dalpha"), (3e-6, 3e-6, 1e-5, 1e-5)):
        assert float(a_.abs().max()) > 0, what
        assert float((a_ - b_).abs().max()) <= tol * float(a_.abs().max()), what


def test_conv3d_split_is_refused_for_other_widths():
    from rendernet_amd import _lib as L
    lib = L.lib()
    assert lib.rn_conv3d_winograd_split_supported(32, 32) == 1
    assert lib.rn_conv3d_winograd_split_supported(16, 16) == 0 and lib.rn_conv3d_winograd_split_supported(32, 64) == 0
    x = torch.zeros((1, 4, 4, 4, 16), device="cuda")
    us = torch.zeros(64, dtype=torch.uint8, device="cuda")
    rc = lib.rn_conv3d_winograd_split_fwd(L.ptr(x), ctypes.c_void_p(us.data_ptr()), None, None, None, L.ptr(x), None, 1, 4, 4, 4, 16, 16, 0, L.stream_ptr())
    assert rc != 0


# ---------------------------------------------------------------------------------------------------------------------
# Filter gradient of the wide 2-D convs with the reduction over the tiles on the bf16 pipe (csrc/conv_wino_bf3_wgrad.hip):
# tf.nn.conv2d_backprop_filter of slim.conv2d [3,3] / [4,4] stride 1 (tools/layer_util.py:101-104, RenderNet_Shader.py:71-103).
@pytest.mark.parametrize("k,B,H,W,Cin,Cout", [(3, 1, 8, 8, 256, 256), (3, 2, 16, 16, 256, 512), (3, 3, 13, 27, 512, 256), (3, 1, 37, 5, 256, 256),
                                               (3, 6, 32, 32, 256, 256), (3, 1, 1, 1, 256, 256), (3, 24, 32, 32, 256, 256),
                                               (4, 1, 8, 8, 256, 256), (4, 2, 16, 16, 512, 256), (4, 3, 13, 27, 256, 256)])
def test_split_wgrad(k, B, H, W, Cin, Cout):
    """vs autograd over the oracle conv, and vs the exact-fp32 Winograd filter gradient of the same layer (3e-5 / 1e-4 of max: same formulas,
    fp32-class multiply); tile counts that are not multiples of 16, K splits, ragged planes, accumulation into dw."""
    from rendernet_amd import _lib as L
    lib = L.lib()
    sch = L.RN_WINO_F43 if k == 3 else L.RN_WINO_F44
    assert lib.rn_winograd_split_wgrad_supported(sch, Cin, Cout) == 1
    assert lib.rn_winograd_split_wgrad_supported(sch, 128, 256) == 0
    assert lib.rn_winograd_split_wgrad_supported(L.RN_WINO_F63, Cin, Cout) == 0
    rng = np.random.default_rng(B * 1000 + H * 31 + W + Cin + k)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    dz = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
    wt = torch.zeros(k, k, Cin, Cout, requires_grad=True)
    OL.conv2d(torch.from_numpy(x), wt).backward(torch.from_numpy(dz))
    xd, dzd = _dev(x), _dev(dz)
    dw = torch.zeros(k, k, Cin, Cout, device="cuda")
    ws = torch.empty(lib.rn_winograd_split_wgrad_workspace_bytes(sch, B, H, W, Cin, Cout), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        L.check(lib.rn_conv2d_winograd_split_wgrad(sch, L.ptr(xd), L.ptr(dzd), L.ptr(dw), ctypes.c_void_p(ws.data_ptr()), B, H, W, Cin, Cout,
                                                   L.stream_ptr()), "split wgrad")
    _close(dw, 2 * wt.grad, "accumulated split dw vs oracle autograd")
    ref = torch.zeros_like(dw)
    if k == 3:
        w2 = torch.empty(lib.rn_conv2d_wino43_wgrad_workspace_floats(B, H, W, Cin, Cout), device="cuda")
        L.check(lib.rn_conv2d_wino43_wgrad(L.ptr(xd), L.ptr(dzd), L.ptr(ref), L.ptr(w2), B, H, W, Cin, Cout, L.stream_ptr()), "wino43 wgrad")
    else:
        w2 = torch.empty(lib.rn_conv2d_wino44_wgrad_workspace_floats(B, H, W, Cin, Cout), device="cuda")
        L.check(lib.rn_conv2d_wino44_wgrad(L.ptr(xd), L.ptr(dzd), L.ptr(ref), L.ptr(w2), B, H, W, Cin, Cout, L.stream_ptr()), "wino44 wgrad")
    _close(dw, (2 * ref).cpu(), "split vs exact-fp32 Winograd wgrad", rtol=3e-5 if k == 3 else 1e-4)


def test_split_wgrad_plan():
    """The planner's choice for the res2 training shape: 36 x 16 blocks = 2.25 rounds of 256 workgroups, whose quarter round the GEMM
    launcher runs as half items (4.5 half rounds of 5: 0.9) -- no K split; a shape with few blocks is split along the tiles."""
    from rendernet_amd import _lib as L
    lib = L.lib()
    # res2 at crop 64: T = 24 * 8 * 8 = 1536 tiles: Vt + dMt + dUp
    assert lib.rn_winograd_split_wgrad_workspace_bytes(L.RN_WINO_F43, 24, 32, 32, 1024, 1024) == 36 * 1536 * 2048 * 6 + 36 * 1024 * 1024 * 4 + 256
    # 256 -> 256 at T = 1536: 36 blocks = 0.28 half rounds; KS = 3 fills 0.84 of one round (KS = 4: 0.56 of two), Tk = 512
    assert lib.rn_winograd_split_wgrad_workspace_bytes(L.RN_WINO_F43, 24, 32, 32, 256, 256) == 108 * 512 * 512 * 6 + 108 * 256 * 256 * 4 + 256


@pytest.mark.parametrize("mode", ["f32", "split", "split16"])
def test_full_width_training_step_against_the_float64_golden(mode, monkeypatch):
    """BASELINE configs[3] at full width (237M parameters, crop 64, two samples): loss, prediction and sampled gradient entries of all
    166 variables against tests/golden/train_step_golden.npz (float64 torch-CPU autograd over the oracle graph) -- the check bench.py
    runs on its train line, here for all three multiply routes.  In the split modes the forward, input-gradient, 3-D encoder and (>= 1024
    channels) filter-gradient stages all run on the 16-bit pipe; the bars are the same."""
    import bench
    from rendernet_amd import ops
    from rendernet_amd.shader import ShaderSpec, init_shader_weights
    from rendernet_amd.train import Trainer
    monkeypatch.setattr(ops._MODE, "mode", mode, raising=False)
    spec = ShaderSpec().check()
    tr = Trainer(spec, init_shader_weights(spec, seed=1234, perturb=True), device="cuda:0")
    got = bench.train_parity(tr, spec, 1)
    print(mode, {k: got[k] for k in ("loss_rel_err", "pred_max_abs_err", "filter_grad_max_rel_err", "bias_alpha_grad_max_rel_err", "worst_variables")})
    assert got["ok"], got
    del tr
    torch.cuda.empty_cache()


def test_split_routes_batch_chunks(monkeypatch):
    """The split launchers cut the batch like the exact ones when a transform plane would not fit one 2-GiB buffer window
    (RN_WINO43_MAX_PLANE lowers the limit for the test): forward through F(4x4,3x3) and F(6x6,3x3) bit for bit equal to the unchunked
    call, the filter gradient to rounding (dw accumulates chunk by chunk: another order)."""
    from rendernet_amd import _lib as L
    lib = L.lib()
    B, H, W, Cin, Cout = 7, 8, 12, 256, 256
    rng = np.random.default_rng(21)
    x = _dev(rng.standard_normal((B, H, W, Cin)).astype(np.float32))
    dz = _dev(rng.standard_normal((B, H, W, Cout)).astype(np.float32))
    w = _dev(_xavier(rng, (3, 3, Cin, Cout)))
    st = L.stream_ptr()

    def fwd(sch):
        us = torch.empty(lib.rn_winograd_split_packed_bytes(sch, Cin, Cout), dtype=torch.uint8, device="cuda")
        L.check(lib.rn_winograd_split_pack(sch, L.ptr(w), ctypes.c_void_p(us.data_ptr()), Cin, Cout, 0, st), "pack")
        ws = torch.empty(lib.rn_winograd_split_workspace_bytes(sch, B, H, W, Cin, Cout), dtype=torch.uint8, device="cuda")
        y = torch.empty((B, H, W, Cout), device="cuda")
        L.check(lib.rn_conv2d_winograd_split_fwd(sch, L.ptr(x), ctypes.c_void_p(us.data_ptr()), None, None, None, L.ptr(y), None,
                                                 ctypes.c_void_p(ws.data_ptr()), B, H, W, Cin, Cout, 0, 0, st), "split fwd")
        return y

    def wgrad():
        dw = torch.zeros(3, 3, Cin, Cout, device="cuda")
        ws = torch.empty(lib.rn_winograd_split_wgrad_workspace_bytes(L.RN_WINO_F43, B, H, W, Cin, Cout), dtype=torch.uint8, device="cuda")
        L.check(lib.rn_conv2d_winograd_split_wgrad(L.RN_WINO_F43, L.ptr(x), L.ptr(dz), L.ptr(dw), ctypes.c_void_p(ws.data_ptr()), B, H, W, Cin, Cout, st), "wgrad")
        return dw

    whole43, whole63, dw_whole = fwd(L.RN_WINO_F43), fwd(L.RN_WINO_F63), wgrad()
    plane = 2 * 3 * 256 * 4                                       # F(4x4): tiles per image * channels * 4 B
    monkeypatch.setenv("RN_WINO43_MAX_PLANE", str(2 * plane + plane // 2))          # forward: two images fit, three do not; filter gradient (6 B): one
    chunk43, chunk63, dw_chunk = fwd(L.RN_WINO_F43), fwd(L.RN_WINO_F63), wgrad()
    assert torch.equal(whole43, chunk43) and torch.equal(whole63, chunk63)
    assert float((dw_whole - dw_chunk).abs().max()) <= 1e-5 * float(dw_whole.abs().max())
    want = OL.conv2d(x.cpu().numpy(), w.cpu().numpy(), None, (1, 1))
    _close(chunk43, want, "chunked split F(4x4,3x3)")
    _close(chunk63, want, "chunked split F(6x6,3x3)")


@pytest.mark.parametrize("smode", ["split", "split16"])
def test_hipgraph_replay_of_the_split_routes(fixtures_vox, smode, monkeypatch):
    """Renderer.capture at full width in the split modes: the memsets, the 4-byte max|x| copies and the atomic maxima of the fp16x2
    route are stream operations like the kernels -- the captured graph replays to the bits of the eager launches, for new inputs too
    (the maxima are order-independent, so the route is deterministic)."""
    from rendernet_amd import ops
    from rendernet_amd.shader import Renderer, ShaderSpec, init_shader_weights
    from bench import synthetic_batch
    monkeypatch.setattr(ops._MODE, "mode", smode, raising=False)
    spec = ShaderSpec().check()
    r = Renderer(spec, init_shader_weights(spec, seed=1234, perturb=True), device="cuda:0")
    vox, poses = synthetic_batch(24)
    r.render(vox[:2], poses[:2])                                   # packs (and the F(6x6,3x3) self-check) happen outside the capture
    replay = r.capture(2)
    for sl in (slice(0, 2), slice(7, 9)):
        want = r.render(vox[sl], poses[sl])
        got = replay(vox[sl], poses[sl])
        torch.cuda.synchronize()
        assert torch.equal(got, want)
    del r, replay
    torch.cuda.empty_cache()


def test_split16_amax_handover_and_its_guard(monkeypatch):
    """fp16x2 route: a layer's launch leaves max|y| on its output (ops: y._rn_amax = (device word, tensor version)); the next layer takes
    its scale from it instead of a pass over x -- unless the tensor was changed in place since, which the version counter shows."""
    from rendernet_amd import ops
    monkeypatch.setattr(ops._MODE, "mode", "split16", raising=False)
    monkeypatch.setattr(ops, "WINO43_MIN_PIXELS", 1)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 24, 24, 256)).astype(np.float32)
    w1, w2 = _xavier(rng, (3, 3, 256, 256)), _xavier(rng, (3, 3, 256, 256))
    misses = []
    monkeypatch.setattr(ops, "AMAX_MISSES", misses)
    with torch.no_grad():
        h = ops.conv2d(_dev(x), ops.pack_conv(_dev(w1)))
        word, version = h._rn_amax
        assert word.view(torch.float32).item() == float(h.abs().max()) and version == h._version
        assert len(misses) == 1                                       # x came from nowhere: one pass
        y = ops.conv2d(h, ops.pack_conv(_dev(w2)))
        assert len(misses) == 1                                       # h brought its maximum along
        _close(y, OL.conv2d(h.cpu().numpy(), w2, None, (1, 1)), "chained split16 conv")
        h.mul_(1000.0)                                                # 1000 x the recorded maximum: the stale scale would overflow fp16
        y2 = ops.conv2d(h, ops.pack_conv(_dev(w2)))
        assert len(misses) == 2                                       # version changed: the launcher looked again
        assert bool(torch.isfinite(y2).all())
        _close(y2, OL.conv2d(h.cpu().numpy(), w2, None, (1, 1)), "split16 conv after an in-place change")


def test_two_renderers_of_one_process_run_different_modes():
    """Review item 10 (round 5): the multiply-stage mode is a Renderer attribute.  Two full-width renderers over the SAME weights, one
    exact fp32 and one bf16x3 split, rendered alternately in one process: each reproduces what a process running only that mode
    computes (the `with ops.gemm_mode(...)` render of a third, mode-less renderer), the two differ (different routes, not one), and
    both sit inside the 1e-3 bar of each other."""
    from rendernet_amd import ops
    from rendernet_amd.shader import Renderer, ShaderSpec, init_shader_weights
    from bench import synthetic_batch
    spec = ShaderSpec().check()
    w = init_shader_weights(spec, seed=1234, perturb=True)
    vox, poses = synthetic_batch(24)
    vox, poses = vox[:2], poses[:2]
    r32, rsp, rdef = Renderer(spec, w, gemm="f32"), Renderer(spec, w, gemm="split"), Renderer(spec, w)
    assert (r32.gemm, rsp.gemm, rdef.gemm) == ("f32", "split", None)
    a1, b1 = r32.render(vox, poses).clone(), rsp.render(vox, poses).clone()
    a2, b2 = r32.render(vox, poses).clone(), rsp.render(vox, poses).clone()              # interleaved again: nothing leaks between them
    assert torch.equal(a1, a2) and torch.equal(b1, b2)
    assert not torch.equal(a1, b1) and float((a1 - b1).abs().max()) <= 1e-3
    with ops.gemm_mode("f32"):
        want32 = rdef.render(vox, poses).clone()
    with ops.gemm_mode("split"):
        wantsp = rdef.render(vox, poses).clone()
    assert torch.equal(a1, want32) and torch.equal(b1, wantsp)
    with ops.gemm_mode("split16"):                                                       # the renderer's own mode wins over the caller's context
        assert torch.equal(r32.render(vox, poses), a1)
    del r32, rsp, rdef
    torch.cuda.empty_cache()

