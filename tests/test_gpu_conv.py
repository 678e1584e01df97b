"""Parity of every conv flavour (through the C ABI) against oracle/layers.py.  -m gpu.
fp32 tolerance: max|got-want| <= 1e-4 * max|want| (exact-fp32 MFMA = fmaf chain; the oracle is
oneDNN fp32 with a different summation order)."""
import numpy as np
import pytest
import torch

from oracle import layers as OL

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _dev(a):
    return None if a is None else torch.as_tensor(a).cuda()


def _rand(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


def _close(got, want, what):
    got = got.cpu().numpy()
    want = want.numpy() if isinstance(want, torch.Tensor) else want
    assert got.shape == tuple(want.shape), (what, got.shape, want.shape)
    err = np.abs(got - want).max()
    ref = np.abs(want).max()
    assert err <= RTOL * ref + 1e-6, "%s: max err %g vs max |ref| %g" % (what, err, ref)


def _xavier(rng, shape):
    rf = int(np.prod(shape[:-2]))
    lim = np.sqrt(6.0 / ((shape[-2] + shape[-1]) * rf))
    return rng.uniform(-lim, lim, shape).astype(np.float32)


# (B, H, W, D, Cin, Cout, k, stride) -- covers: direct stem (Cin 1/5/8), igemm BN=32/64/128 with
# BK=16/32, ragged M (not a multiple of 128), strided depth with (0,1) padding, k=4 (pad 1,2), k=5 s2
CONV3D_CASES = [
    (2, 16, 16, 16, 1, 8, 5, (2, 2, 2)),        # e_conv1
    (1, 12, 10, 16, 5, 8, 5, (2, 2, 2)),        # texture-net stem, ragged
    (2, 8, 8, 8, 8, 16, 3, (1, 1, 2)),          # e_conv2
    (2, 8, 8, 4, 16, 32, 3, (1, 1, 1)),         # e_conv3 (BK=16, BN=32)
    (2, 8, 8, 4, 32, 32, 3, (1, 1, 1)),         # res1 (BK=32, BN=32)
    (1, 5, 7, 3, 32, 32, 3, (1, 1, 1)),         # ragged M = 105
    (1, 6, 6, 6, 16, 16, 3, (1, 1, 1)),         # texture 16-ch trunk (Cout 16 -> Npad 32)
    (1, 8, 8, 8, 8, 4, 4, (1, 1, 1)),           # texture decoder conv3d k4 (pad 1,2), direct
    (1, 6, 6, 6, 32, 64, 3, (1, 1, 1)),         # BN=64
    (1, 4, 4, 4, 64, 128, 3, (2, 2, 2)),        # BN=128, stride 2
]


@pytest.mark.parametrize("cin", [1, 5])
def test_conv3d_stem_sparse_and_deep(cin, monkeypatch):
    """The row-staged 5^3 stride-2 stem kernel (conv_stem_kernel): a mostly empty volume (all-zero rows are skipped), more than
    64 output depths (two lane chunks), odd sizes (SAME pads 2 before on odd extents, 1 on even ones), ragged row groups."""
    from rendernet_amd import ops
    monkeypatch.setenv("RN_STEM_KERNEL_CIN1", "1")                   # read once per process: harmless if already latched
    rng = np.random.default_rng(77 + cin)
    for (B, H, W, D) in [(2, 10, 14, 136), (1, 9, 7, 21)]:
        x = _rand(rng, B, H, W, D, cin)
        x[:, :, : W // 2] = 0.0                                       # half of the rows empty
        x[:, H // 2:, :, D // 3:] = 0.0
        w = _xavier(rng, (5, 5, 5, cin, 8))
        b = _rand(rng, 8) * 0.1
        alpha = rng.uniform(0, 0.25, 8).astype(np.float32)
        pw = ops.pack_conv(_dev(w))
        _close(ops.conv3d(_dev(x), pw, _dev(b), _dev(alpha), stride=(2, 2, 2)), OL.prelu(OL.conv3d(x, w, b, (2, 2, 2)), alpha),
               "stem %s" % ((B, H, W, D, cin),))


@pytest.mark.parametrize("case", CONV3D_CASES)
def test_conv3d(case):
    from rendernet_amd import ops
    B, H, W, D, Cin, Cout, k, s = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = _rand(rng, B, H, W, D, Cin)
    w = _xavier(rng, (k, k, k, Cin, Cout))
    b = _rand(rng, Cout) * 0.1
    alpha = rng.uniform(0, 0.25, Cout).astype(np.float32)
    pw = ops.pack_conv(_dev(w))
    # plain
    _close(ops.conv3d(_dev(x), pw, _dev(b), stride=s), OL.conv3d(x, w, b, s), "conv3d")
    # bias + PReLU
    want = OL.prelu(OL.conv3d(x, w, b, s), alpha)
    _close(ops.conv3d(_dev(x), pw, _dev(b), _dev(alpha), stride=s), want, "conv3d+prelu")
    # bias + residual
    y0 = OL.conv3d(x, w, b, s)
    res = _rand(rng, *y0.shape)
    _close(ops.conv3d(_dev(x), pw, _dev(b), None, _dev(res), stride=s), y0 + torch.from_numpy(res), "conv3d+res")
    # no bias
    _close(ops.conv3d(_dev(x), pw, None, stride=s), OL.conv3d(x, w, None, s), "conv3d nobias")


CONV2D_CASES = [
    (2, 16, 16, 256, 256, 3, (1, 1)),      # res2-like (BN=128, BK=32)
    (1, 9, 11, 128, 64, 3, (1, 1)),        # ragged M, BN=64
    (2, 8, 8, 256, 128, 4, (1, 1)),        # e_conv5-like 4x4 (pad 1,2)
    (1, 8, 8, 64, 32, 4, (1, 1)),          # BN=32
    (1, 16, 16, 48, 96, 3, (1, 1)),        # Cin multiple of 16 only (BK=16), Cout 96 -> Npad 96 (BN=32)
    (1, 16, 16, 32, 32, 3, (2, 2)),        # strided 2-D conv
    (1, 12, 12, 3, 8, 3, (1, 1)),          # direct path, Cin=3
    (2, 8, 8, 1024, 128, 1, (1, 1)),       # 1x1, long K
]


@pytest.mark.parametrize("case", CONV2D_CASES)
def test_conv2d(case):
    from rendernet_amd import ops
    B, H, W, Cin, Cout, k, s = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = _rand(rng, B, H, W, Cin)
    w = _xavier(rng, (k, k, Cin, Cout))
    b = _rand(rng, Cout) * 0.1
    alpha = rng.uniform(0, 0.25, Cout).astype(np.float32)
    pw = ops.pack_conv(_dev(w))
    y0 = OL.conv2d(x, w, b, s)
    _close(ops.conv2d(_dev(x), pw, _dev(b), stride=s), y0, "conv2d")
    res = _rand(rng, *y0.shape)
    want = OL.prelu(y0, alpha) + torch.from_numpy(res)
    _close(ops.conv2d(_dev(x), pw, _dev(b), _dev(alpha), _dev(res), stride=s), want, "conv2d+prelu+res")
    _close(ops.conv2d(_dev(x), pw, _dev(b), stride=s, sigmoid=True), torch.sigmoid(y0), "conv2d+sigmoid")


CONVT2D_CASES = [
    (2, 8, 8, 256, 128, 2),    # e_conv7
    (1, 16, 16, 128, 128, 1),  # e_conv7_1
    (1, 16, 16, 128, 64, 2),   # e_conv8
    (1, 9, 7, 64, 32, 2),      # e_conv9, ragged
    (1, 16, 16, 32, 16, 1),    # e_conv10 (Cout 16 -> Npad 32)
    (2, 16, 16, 16, 1, 1),     # e_conv11 greyscale (direct)
    (1, 16, 16, 16, 3, 1),     # e_conv11 RGB (direct)
    (1, 8, 8, 16, 3, 2),       # texture heads: stride-2 to 3 channels (direct, phases)
    (1, 37, 21, 16, 1, 1),     # e_conv11 strip kernel (csrc/conv_tiled.hip: conv_tail_kernel): two row strips of 32, two column strips of 16, both ragged
    (2, 70, 50, 16, 3, 1),     # ... RGB, three row strips, four column strips (the last one 2 pixels wide), two images
    (1, 3, 5, 16, 1, 1),       # ... a map smaller than one strip and than the filter's reach
    (1, 37, 21, 32, 1, 1),     # the 32-channel tail of the 128^3 -> 1024^2 config on the same kernel (8 channel quads, 8 columns per wave): ragged strips
    (2, 70, 50, 32, 3, 1),     # ... RGB, seven column strips (the last one 2 pixels wide), two images
    (1, 3, 5, 32, 1, 1),       # ... smaller than one strip
]


@pytest.mark.parametrize("case", CONVT2D_CASES)
def test_conv2d_transpose(case):
    from rendernet_amd import ops
    B, H, W, Cin, Cout, s = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = _rand(rng, B, H, W, Cin)
    w = _xavier(rng, (4, 4, Cout, Cin))
    b = _rand(rng, Cout) * 0.1
    alpha = rng.uniform(0, 0.25, Cout).astype(np.float32)
    pw = ops.pack_conv_transpose(_dev(w), s)
    y0 = OL.conv2d_transpose(x, w, b, (s, s))
    _close(ops.conv2d_transpose(_dev(x), pw, _dev(b), stride=(s, s)), y0, "convT")
    _close(ops.conv2d_transpose(_dev(x), pw, _dev(b), _dev(alpha), stride=(s, s)), OL.prelu(y0, alpha), "convT+prelu")
    _close(ops.conv2d_transpose(_dev(x), pw, _dev(b), stride=(s, s), sigmoid=True), torch.sigmoid(y0), "convT+sigmoid")


@pytest.mark.parametrize("case", [(1, 6, 6, 6, 4, 4, 1), (1, 5, 6, 4, 4, 8, 2), (1, 4, 4, 4, 16, 16, 2)])
def test_conv3d_transpose(case):
    from rendernet_amd import ops
    B, H, W, D, Cin, Cout, s = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = _rand(rng, B, H, W, D, Cin)
    w = _xavier(rng, (4, 4, 4, Cout, Cin))
    b = _rand(rng, Cout) * 0.1
    pw = ops.pack_conv_transpose(_dev(w), s)
    _close(ops.conv3d_transpose(_dev(x), pw, _dev(b), stride=(s, s, s)), OL.conv3d_transpose(x, w, b, (s, s, s)), "convT3d")


@pytest.mark.parametrize("case", [(2, 8, 8, 8, 32), (1, 5, 7, 4, 16), (1, 4, 4, 32, 32)])
def test_projection_unit(case):
    """Depth-flatten + 1x1 conv + PReLU in one kernel, feature index f = d*C + c."""
    from rendernet_amd import ops
    B, H, W, D, C = case
    F = D * C
    rng = np.random.default_rng(hash(case) % 2**31)
    x = _rand(rng, B, H, W, D, C)
    w = _xavier(rng, (1, 1, F, F))
    b = _rand(rng, F) * 0.1
    alpha = rng.uniform(0, 0.25, F).astype(np.float32)
    pw = ops.pack_conv(_dev(w))
    _close(ops.projection(_dev(x), pw, _dev(b), _dev(alpha)), OL.projection_unit(x, w, b, alpha), "projection")


def test_fully_connected_and_prelu():
    from rendernet_amd import ops
    rng = np.random.default_rng(11)
    x = _rand(rng, 5, 199)
    w = _rand(rng, 199, 1000) * 0.02
    b = _rand(rng, 1000) * 0.1
    alpha = rng.uniform(0, 0.25, 1000).astype(np.float32)
    _close(ops.fully_connected(_dev(x), _dev(w), _dev(b), _dev(alpha)), OL.prelu(OL.fully_connected(x, w, b), alpha), "fc")
    t = _rand(rng, 3, 7, 5, 24)
    a = rng.uniform(0, 0.25, 24).astype(np.float32)
    _close(ops.prelu(_dev(t), _dev(a)), OL.prelu(t, a), "prelu")


def test_asymmetric_weights_catch_transposes():
    """A = I check with an asymmetric B: a 1x1 conv with an identity-like input must return the
    filter rows themselves (catches row/col swaps in the MFMA C/D mapping)."""
    from rendernet_amd import ops
    Cin, Cout = 64, 128
    x = np.zeros((1, 8, 8, Cin), np.float32)
    for p in range(64):
        x[0, p // 8, p % 8, p] = 1.0
    w = (np.arange(Cin)[:, None] * 1000 + np.arange(Cout)[None, :]).astype(np.float32).reshape(1, 1, Cin, Cout)
    pw = ops.pack_conv(_dev(w))
    got = ops.conv2d(_dev(x), pw).cpu().numpy().reshape(64, Cout)
    assert np.array_equal(got, w.reshape(Cin, Cout))


def test_bad_arguments_raise():
    from rendernet_amd import ops
    from rendernet_amd._lib import RenderNetHipError
    w = torch.zeros((3, 3, 16, 16), device="cuda")
    pw = ops.pack_conv(w)
    with pytest.raises(RenderNetHipError):
        ops.conv2d(torch.zeros((1, 4, 4, 8), device="cuda"), pw)       # channel mismatch
    with pytest.raises(RenderNetHipError):
        ops.conv2d(torch.zeros((1, 4, 4, 16)), pw)                     # CPU tensor: no fallback


# Winograd F(2x2,3x3) path (csrc/conv_wino.hip): ragged H/W (partial tile blocks, odd sizes), several column blocks
# (W > 32), Cin = 16 / 48 (1 and 3 steps), Cout = 32 / 96, the bench widths, batch spanning m-block groups.
WINO_CASES = [
    (2, 16, 16, 256, 256),
    (1, 9, 11, 128, 64),
    (1, 37, 70, 16, 32),
    (3, 33, 5, 48, 96),
    (2, 64, 64, 64, 64),
    (1, 16, 16, 1024, 512),
    (12, 32, 32, 32, 32),
    (2, 20, 40, 32, 16),                    # Cout % 32 != 0: one 16-channel n-tile per wave
    (1, 16, 16, 48, 48),
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv2d_winograd(case):
    """rn_conv2d_wino_fwd vs the oracle conv, vs the direct implicit-GEMM kernel on the same filter, and its packed
    filter vs the NumPy statement of the layout (scripts/wino_emulate.pack_wino)."""
    from rendernet_amd import ops
    from scripts.wino_emulate import pack_wino
    B, H, W, Cin, Cout = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = _rand(rng, B, H, W, Cin)
    w = _xavier(rng, (3, 3, Cin, Cout))
    b = _rand(rng, Cout) * 0.1
    alpha = rng.uniform(0, 0.25, Cout).astype(np.float32)
    pw = ops.pack_conv(_dev(w))
    pw.wino43 = None                                                        # this test is about the F(2x2,3x3) kernel
    assert pw.wino is not None, "the Winograd pack must exist for a 3x3 filter with Cin%16==0, Cout%16==0"
    want_u = pack_wino(w)
    assert np.abs(pw.wino.cpu().numpy() - want_u).max() <= 1e-6 * np.abs(want_u).max()
    y0 = OL.conv2d(x, w, b, (1, 1))
    _close(ops.conv2d(_dev(x), pw, _dev(b)), y0, "wino")
    res = _rand(rng, *y0.shape)
    want = OL.prelu(y0, alpha) + torch.from_numpy(res)
    got = ops.conv2d(_dev(x), pw, _dev(b), _dev(alpha), _dev(res))
    _close(got, want, "wino+prelu+res")
    _close(ops.conv2d(_dev(x), pw, None, sigmoid=True), torch.sigmoid(OL.conv2d(x, w, None, (1, 1))), "wino+sigmoid")
    # A/B against the direct kernel (same entry, Winograd pack dropped)
    pd = ops.pack_conv(_dev(w))
    pd.wino = None
    pd.wino43 = None
    direct = ops.conv2d(_dev(x), pd, _dev(b), _dev(alpha), _dev(res))
    assert float((got - direct).abs().max()) <= 2e-5 * float(direct.abs().max())
    # input gradient through the transposed Winograd pack == the direct dgrad kernel's result
    dp = pw.dgrad_pack(True)
    assert dp.wino is not None                                              # roles swap: K = Cout, N = Cin, both % 16
    dz = _dev(_rand(rng, B, H, W, Cout))
    from rendernet_amd import _lib as L
    dx_w = torch.empty((B, H, W, Cin), device="cuda")
    L.check(L.lib().rn_conv2d_wino_fwd(L.ptr(dz), L.ptr(dp.wino), None, None, None, L.ptr(dx_w), None, B, H, W, Cout, Cin, 0,
                                       L.stream_ptr()), "rn_conv2d_wino_fwd (dgrad)")
    dx_d = torch.empty_like(dx_w)
    L.check(L.lib().rn_conv2d_dgrad(L.ptr(dz), L.ptr(dp.data), L.ptr(dx_d), B, H, W, Cin, Cout, L.ivec([3, 3]), L.ivec([1, 1]),
                                    L.stream_ptr()), "rn_conv2d_dgrad")
    want_dx = OL.conv2d_transpose(dz.cpu().numpy(), w, None, (1, 1))       # w read as [k,k,Cout_T = Cin,Cin_T = Cout]
    _close(dx_w, want_dx, "wino dgrad vs oracle")
    assert float((dx_w - dx_d).abs().max()) <= 2e-5 * float(dx_d.abs().max())


# Winograd F(4x4,3x3) path (csrc/conv_wino43.hip; input transform, 36 GEMMs, output transform): planes that are not
# multiples of 4 (ragged tiles on both axes), tile counts below / across / above the 256-row GEMM block, Cin = 32 (one
# K step) .. 1024, Cout = 256 .. 1024 (1 .. 4 channel blocks), item counts that are not a multiple of the persistent grid.
WINO43_CASES = [
    (1, 4, 4, 32, 256),
    (2, 16, 16, 256, 256),
    (1, 9, 11, 64, 512),
    (3, 33, 5, 96, 256),
    (1, 16, 16, 1024, 512),
    (2, 64, 64, 64, 256),
    (5, 30, 34, 128, 1024),
    (1, 1, 1, 32, 256),
]


def _pack_wino43_numpy(w, scheme="F43", transposed=False):
    """NumPy statement of the RN_PACK_CONV_WINO43 / _WINO44 layouts (conv_wino43.hip, wino_pack_kernel): U = G g G^T with
    the generated matrices (scripts/gen_wino_mats.py), packed [nxi][Cout/256][Cin/4][256][4]."""
    from scripts.gen_wino_mats import SCHEMES, winograd_mats
    m, r, pts = SCHEMES[scheme]
    _, G, _ = winograd_mats(m, r, pts)
    G = np.array([[float(v) for v in row] for row in G], np.float64)
    if transposed:                                      # conv_transpose filter [R,R,Cout,Cin], taps flipped
        w = w[::-1, ::-1].transpose(0, 1, 3, 2)
    Cin, Cout = w.shape[2], w.shape[3]
    nxi = (m + r - 1) ** 2
    U = np.einsum("ia,abck,jb->ijck", G, w.astype(np.float64), G).reshape(nxi, Cin, Cout)
    out = np.empty((nxi, Cout // 256, Cin // 4, 256, 4), np.float32)
    for nb in range(Cout // 256):
        blk = U[:, :, nb * 256:(nb + 1) * 256]                               # [nxi, Cin, 256 channels of the block]
        out[:, nb] = blk.reshape(nxi, Cin // 4, 4, 256).transpose(0, 1, 3, 2)
    return out.reshape(-1)


@pytest.mark.parametrize("case", WINO43_CASES)
def test_conv2d_winograd43(case):
    """rn_conv2d_wino43_fwd vs the oracle conv (every epilogue flavour, the pre-activation output), vs the F(2x2,3x3) kernel
    on the same filter, its packed filter vs the NumPy statement, and the input gradient through the transposed pack."""
    from rendernet_amd import ops
    from rendernet_amd import _lib as L
    B, H, W, Cin, Cout = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = _rand(rng, B, H, W, Cin)
    w = _xavier(rng, (3, 3, Cin, Cout))
    b = _rand(rng, Cout) * 0.1
    alpha = rng.uniform(0, 0.25, Cout).astype(np.float32)
    old, ops.WINO43_MIN_PIXELS = ops.WINO43_MIN_PIXELS, 1
    try:
        pw = ops.pack_conv(_dev(w))
        pw.wino63 = None                                    # this test is about F(4x4,3x3): keep ops from picking F(6x6,3x3) on the larger maps
        assert pw.wino43 is not None
        want_u = _pack_wino43_numpy(w)
        assert np.abs(pw.wino43.cpu().numpy() - want_u).max() <= 1e-6 * np.abs(want_u).max()
        y0 = OL.conv2d(x, w, b, (1, 1))
        _close(ops.conv2d(_dev(x), pw, _dev(b)), y0, "wino43")
        res = _rand(rng, *y0.shape)
        want = OL.prelu(y0, alpha) + torch.from_numpy(res)
        got = ops.conv2d(_dev(x), pw, _dev(b), _dev(alpha), _dev(res))
        _close(got, want, "wino43+prelu+res")
        _close(ops.conv2d(_dev(x), pw, None, sigmoid=True), torch.sigmoid(OL.conv2d(x, w, None, (1, 1))), "wino43+sigmoid")
        # the pre-activation output of the training forward
        xd, wd, bd, ad = _dev(x), pw.wino43, _dev(b), _dev(alpha)
        yy, zz = torch.empty((B, H, W, Cout), device="cuda"), torch.empty((B, H, W, Cout), device="cuda")
        ws = torch.empty(L.lib().rn_conv2d_wino43_workspace_floats(B, H, W, Cin, Cout), device="cuda")
        L.check(L.lib().rn_conv2d_wino43_fwd(L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(ad), None, L.ptr(yy), L.ptr(zz), L.ptr(ws),
                                             B, H, W, Cin, Cout, 1, L.stream_ptr()), "rn_conv2d_wino43_fwd")
        _close(zz, y0, "wino43 preact")
        _close(yy, OL.prelu(y0, alpha), "wino43 prelu")
        # A/B against the F(2x2,3x3) kernel
        p2 = ops.pack_conv(_dev(w))
        p2.wino43 = None
        if p2.wino is not None:
            ref2 = ops.conv2d(_dev(x), p2, _dev(b), _dev(alpha), _dev(res))
            assert float((got - ref2).abs().max()) <= 1e-4 * float(ref2.abs().max())
        # input gradient through the transposed pack (roles swap: needs Cout % 32 == 0 -- always -- and Cin % 256 == 0)
        dp = pw.dgrad_pack(True)
        if Cin % 256 == 0:
            assert dp.wino43 is not None
            dz = _dev(_rand(rng, B, H, W, Cout))
            dx = torch.empty((B, H, W, Cin), device="cuda")
            ws = torch.empty(L.lib().rn_conv2d_wino43_workspace_floats(B, H, W, Cout, Cin), device="cuda")
            L.check(L.lib().rn_conv2d_wino43_fwd(L.ptr(dz), L.ptr(dp.wino43), None, None, None, L.ptr(dx), None, L.ptr(ws),
                                                 B, H, W, Cout, Cin, 0, L.stream_ptr()), "rn_conv2d_wino43_fwd (dgrad)")
            _close(dx, OL.conv2d_transpose(dz.cpu().numpy(), w, None, (1, 1)), "wino43 dgrad vs oracle")
        else:
            assert dp.wino43 is None
    finally:
        ops.WINO43_MIN_PIXELS = old


# Winograd F(6x6,3x3) (same three launches on 8x8 tiles, 64 planes): planes that are not multiples of 6, and tile counts T that
# walk every branch of the GEMM launcher -- no whole 256-row block (T = 1, 18, 100, 150, 242), whole blocks + a ragged last block
# run as quarter (T = 363: 107 rows), half (T = 100 with 256 (xi, n-block) pairs) and whole items (T = 726: 214 rows), a main
# launch that is itself a half-item tail (T = 363, Cout = 512), deep K (Cin = 1024).
WINO63_CASES = [
    (1, 6, 6, 32, 256),
    (1, 1, 1, 32, 256),
    (3, 33, 5, 96, 256),
    (1, 13, 16, 1024, 512),
    (5, 30, 34, 128, 1024),
    (1, 60, 60, 64, 1024),
    (2, 64, 64, 64, 256),
    (3, 64, 64, 256, 512),
    (6, 64, 64, 32, 1024),
]


@pytest.mark.parametrize("case", WINO63_CASES)
def test_conv2d_winograd63(case, monkeypatch):
    """rn_conv2d_wino63_fwd vs the oracle conv (epilogue flavours, pre-activation), vs F(4x4,3x3) on the same filter, its packed
    filter vs the NumPy statement, the input gradient through the transposed pack, and the routing rule of ops._wino_scheme.
    (The exact-fp32 entry: the dispatcher is pinned to RN_WINO_GEMM=f32 for the bit-equality with it; the split entries have the
    same test in tests/test_gpu_wino_split.py.)"""
    from rendernet_amd import ops
    from rendernet_amd import _lib as L
    monkeypatch.setattr(ops._MODE, "mode", "f32", raising=False)
    B, H, W, Cin, Cout = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = _rand(rng, B, H, W, Cin)
    w = _xavier(rng, (3, 3, Cin, Cout))
    b = _rand(rng, Cout) * 0.1
    alpha = rng.uniform(0, 0.25, Cout).astype(np.float32)
    lib = L.lib()
    pw = ops.pack_conv(_dev(w))
    assert pw.wino63 is not None
    want_u = _pack_wino43_numpy(w, "F63")
    assert np.abs(pw.wino63.cpu().numpy() - want_u).max() <= 1e-6 * np.abs(want_u).max()
    c63, c43 = -(-H // 6) * -(-W // 6) * 64, -(-H // 4) * -(-W // 4) * 36
    assert ops._wino_scheme(pw, H, W) == ("f63" if c63 <= (1.0 - ops.WINO63_MIN_GAIN) * c43 else "f43")
    assert ops._wino_scheme(pw, 64, 64) == "f63" and ops._wino_scheme(pw, 32, 32) == "f43"
    y0 = OL.conv2d(x, w, b, (1, 1))
    res = _rand(rng, *y0.shape)

    def run(xd, u, bias, al, rs, cin, cout, act, preact=False):
        yy = torch.empty((B, H, W, cout), device="cuda")
        zz = torch.empty((B, H, W, cout), device="cuda") if preact else None
        ws = torch.empty(lib.rn_conv2d_wino63_workspace_floats(B, H, W, cin, cout), device="cuda")
        L.check(lib.rn_conv2d_wino63_fwd(L.ptr(xd), L.ptr(u), L.ptr(bias) if bias is not None else None, L.ptr(al) if al is not None else None,
                                         L.ptr(rs) if rs is not None else None, L.ptr(yy), L.ptr(zz) if preact else None, L.ptr(ws),
                                         B, H, W, cin, cout, act, L.stream_ptr()), "rn_conv2d_wino63_fwd")
        return yy, zz
    xd, bd, ad, rd = _dev(x), _dev(b), _dev(alpha), _dev(res)
    yy, zz = run(xd, pw.wino63, bd, ad, None, Cin, Cout, 1, preact=True)
    _close(zz, y0, "wino63 preact")
    _close(yy, OL.prelu(y0, alpha), "wino63 prelu")
    got, _ = run(xd, pw.wino63, bd, ad, rd, Cin, Cout, 1)
    _close(got, OL.prelu(y0, alpha) + torch.from_numpy(res), "wino63+prelu+res")
    # A/B against F(4x4,3x3) on the same filter: the two differ by their fp32 rounding only
    p2 = ops.pack_conv(_dev(w))
    p2.wino63 = None
    old, ops.WINO43_MIN_PIXELS = ops.WINO43_MIN_PIXELS, 1
    try:
        ref2 = ops.conv2d(xd, p2, bd, ad, rd)
        if ops._wino_scheme(pw, H, W) == "f63":            # and through the dispatcher when it picks F(6x6,3x3) itself
            assert torch.equal(ops.conv2d(xd, pw, bd, ad, rd), got)
    finally:
        ops.WINO43_MIN_PIXELS = old
    assert float((got - ref2).abs().max()) <= 1e-4 * float(ref2.abs().max())
    # input gradient through the transposed pack (channel roles swap: needs Cin % 256 == 0)
    dp = pw.dgrad_pack(True)
    if Cin % 256 == 0:
        assert dp.wino63 is not None
        dz = _dev(_rand(rng, B, H, W, Cout))
        dx, _ = run(dz, dp.wino63, None, None, None, Cout, Cin, 0)
        _close(dx, OL.conv2d_transpose(dz.cpu().numpy(), w, None, (1, 1)), "wino63 dgrad vs oracle")
    else:
        assert dp.wino63 is None


# Winograd F(4x4,4x4) for the wide 4x4 stride-1 layers (same three launches on 7x7 tiles): SAME conv (pad 1 before, 2
# after) and stride-1 transposed conv (2 before, 1 after), ragged planes, 1..2 channel blocks, the e_conv5 / e_conv6 widths.
WINO44_CASES = [
    (1, 4, 4, 32, 256),
    (2, 16, 16, 256, 256),
    (1, 9, 11, 64, 512),
    (3, 33, 5, 96, 256),
    (1, 16, 16, 1024, 512),
    (2, 32, 32, 512, 256),
    (1, 1, 1, 32, 256),
]


def test_conv2d_winograd43_batch_chunks(monkeypatch):
    """The F(4x4) launchers split the batch when a transform plane would not fit one buffer resource (2 GiB).  Reaching that for
    real needs ~150 GB of workspace, so RN_WINO43_MAX_PLANE lowers the limit: 7 items in chunks of 2, 2, 2, 1 must equal the
    unchunked call bit for bit (forward) and to rounding (filter gradient: a different accumulation order)."""
    from rendernet_amd import ops, _lib as L
    B, H, W, Cin, Cout = 7, 8, 12, 256, 256
    rng = np.random.default_rng(21)
    x, w = _dev(_rand(rng, B, H, W, Cin)), _dev(_xavier(rng, (3, 3, Cin, Cout)))
    dz = _dev(_rand(rng, B, H, W, Cout))
    old, ops.WINO43_MIN_PIXELS = ops.WINO43_MIN_PIXELS, 1
    lib = L.lib()

    def wgrad():
        dw = torch.zeros(3, 3, Cin, Cout, device="cuda")
        ws = torch.empty(lib.rn_conv2d_wino43_wgrad_workspace_floats(B, H, W, Cin, Cout), device="cuda")
        L.check(lib.rn_conv2d_wino43_wgrad(L.ptr(x), L.ptr(dz), L.ptr(dw), L.ptr(ws), B, H, W, Cin, Cout, L.stream_ptr()), "wgrad")
        return dw
    def f63():                                                    # the F(6x6,3x3) entry: 2 x 2 tiles per image, chunks of 3, 3, 1
        y = torch.empty((B, H, W, Cout), device="cuda")
        ws = torch.empty(lib.rn_conv2d_wino63_workspace_floats(B, H, W, Cin, Cout), device="cuda")
        L.check(lib.rn_conv2d_wino63_fwd(L.ptr(x), L.ptr(pw.wino63), None, None, None, L.ptr(y), None, L.ptr(ws), B, H, W, Cin, Cout, 0,
                                         L.stream_ptr()), "rn_conv2d_wino63_fwd")
        return y
    try:
        pw = ops.pack_conv(w)
        whole, dw_whole, whole63 = ops.conv2d(x, pw), wgrad(), f63()
        plane = 2 * 3 * 256 * 4                                   # tiles per image * channels * 4 B
        monkeypatch.setenv("RN_WINO43_MAX_PLANE", str(2 * plane + plane // 2))      # two images fit, three do not
        chunked, dw_chunked, chunked63 = ops.conv2d(x, pw), wgrad(), f63()
        assert torch.equal(whole, chunked)
        assert torch.equal(whole63, chunked63)
        assert float((whole63 - whole).abs().max()) <= 1e-4 * float(whole.abs().max())
        assert float((dw_whole - dw_chunked).abs().max()) <= 1e-5 * float(dw_whole.abs().max())
    finally:
        ops.WINO43_MIN_PIXELS = old


@pytest.mark.parametrize("case", WINO44_CASES)
def test_conv2d_winograd44(case):
    """rn_conv2d_wino44_fwd (conv and stride-1 transposed conv) vs the oracle, vs the F(2x2,2x2)x4 kernel on the same filter,
    the packed filters vs their NumPy statement, and the input gradients through the opposite packs."""
    from rendernet_amd import ops
    from rendernet_amd import _lib as L
    B, H, W, Cin, Cout = case
    rng = np.random.default_rng(hash(case) % 2**31 + 7)
    x = _rand(rng, B, H, W, Cin)
    b = _rand(rng, Cout) * 0.1
    alpha = rng.uniform(0, 0.25, Cout).astype(np.float32)
    old, ops.WINO43_MIN_PIXELS = ops.WINO43_MIN_PIXELS, 1
    try:
        # conv
        w = _xavier(rng, (4, 4, Cin, Cout))
        pw = ops.pack_conv(_dev(w))
        assert pw.wino43 is not None
        want_u = _pack_wino43_numpy(w, "F44")
        assert np.abs(pw.wino43.cpu().numpy() - want_u).max() <= 1e-6 * np.abs(want_u).max()
        y0 = OL.conv2d(x, w, b, (1, 1))
        res = _rand(rng, *y0.shape)
        got = ops.conv2d(_dev(x), pw, _dev(b), _dev(alpha), _dev(res))
        _close(got, OL.prelu(y0, alpha) + torch.from_numpy(res), "wino44 conv+prelu+res")
        p2 = ops.pack_conv(_dev(w))
        p2.wino43 = None
        if p2.wino4 is not None:
            ref2 = ops.conv2d(_dev(x), p2, _dev(b), _dev(alpha), _dev(res))
            assert float((got - ref2).abs().max()) <= 1e-4 * float(ref2.abs().max())
        if Cin % 256 == 0:                               # the conv's input gradient = stride-1 transposed conv of dz
            dp = pw.dgrad_pack(True)
            assert dp.wino43 is not None
            dz = _dev(_rand(rng, B, H, W, Cout))
            dx = torch.empty((B, H, W, Cin), device="cuda")
            ws = torch.empty(L.lib().rn_conv2d_wino44_workspace_floats(B, H, W, Cout, Cin), device="cuda")
            L.check(L.lib().rn_conv2d_wino44_fwd(L.ptr(dz), L.ptr(dp.wino43), None, None, None, L.ptr(dx), None, L.ptr(ws),
                                                 B, H, W, Cout, Cin, 1, 0, L.stream_ptr()), "rn_conv2d_wino44_fwd (dgrad)")
            _close(dx, OL.conv2d_transpose(dz.cpu().numpy(), w, None, (1, 1)), "wino44 dgrad vs oracle")
        # stride-1 transposed conv (filter [4,4,Cout,Cin])
        wt = _xavier(rng, (4, 4, Cout, Cin))
        pt = ops.pack_conv_transpose(_dev(wt), 1)
        assert pt.wino43 is not None
        want_ut = _pack_wino43_numpy(wt, "F44", transposed=True)
        assert np.abs(pt.wino43.cpu().numpy() - want_ut).max() <= 1e-6 * np.abs(want_ut).max()
        yt = OL.conv2d_transpose(x, wt, b, (1, 1))
        _close(ops.conv2d_transpose(_dev(x), pt, _dev(b), _dev(alpha), None, (1, 1)), OL.prelu(yt, alpha), "wino44 convT+prelu")
    finally:
        ops.WINO43_MIN_PIXELS = old


# ---------------------------------------------------------------------------------------------------------------
# Inputs of 2 GiB and more: the kernels address their A operand with 32-bit byte offsets whose upper half is reserved
# for the hardware zero fill, so the launchers split the batch into chunks that fit the window (conv_igemm.hip,
# conv_wino.hip; conv3d_drun.hip hands such inputs to the implicit-GEMM kernel).  Each case: the big call vs the same
# layer on single items (no chunking) for items on both sides of every chunk boundary, and vs the oracle on one item.
# ---------------------------------------------------------------------------------------------------------------
def _chunk_items(B, per_item_bytes):
    chunk = int(0x7fffffff // per_item_bytes)
    edges = sorted({0, B - 1, chunk - 1, chunk, min(B - 1, 2 * chunk - 1), min(B - 1, 2 * chunk)})
    return chunk, [i for i in edges if 0 <= i < B]


def test_conv_transpose_input_over_2gib_batch_chunks():
    """e_conv10's shape (conv2d_transpose 4x4 s1 32->16 on 512x512 maps, RenderNet_Shader.py:121-123) at batch 64:
    2.1 GiB of input, what an 8-way-larger batch per GPU would hit."""
    from rendernet_amd import ops
    B, H, W, Cin, Cout = 64, 512, 512, 32, 16
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((B, H, W, Cin), device="cuda", generator=g)
    assert x.numel() * 4 >= 2 ** 31
    rng = np.random.default_rng(5)
    w = _xavier(rng, (4, 4, Cout, Cin))
    b = _rand(rng, Cout) * 0.1
    alpha = rng.uniform(0, 0.25, Cout).astype(np.float32)
    pw = ops.pack_conv_transpose(_dev(w), 1)
    y = ops.conv2d_transpose(x, pw, _dev(b), _dev(alpha))
    chunk, items = _chunk_items(B, H * W * Cin * 4)
    assert chunk < B
    for i in items:
        yi = ops.conv2d_transpose(x[i:i + 1].clone(), pw, _dev(b), _dev(alpha))
        assert torch.equal(y[i:i + 1], yi), "item %d differs between the chunked and the single-item launch" % i
    i = items[len(items) // 2]
    want = OL.prelu(OL.conv2d_transpose(x[i:i + 1].cpu().numpy(), w, b, (1, 1)), alpha)
    _close(y[i:i + 1], want, "chunked conv2d_transpose vs oracle")
    del x, y
    torch.cuda.empty_cache()


def test_winograd_and_conv3d_inputs_over_2gib_batch_chunks():
    from rendernet_amd import ops
    g = torch.Generator(device="cuda").manual_seed(6)
    rng = np.random.default_rng(6)
    # Winograd kernel: 129 items of 64x64x1024 (the res2 input shape), 32 output channels to keep it cheap
    B, H, W, Cin, Cout = 129, 64, 64, 1024, 32
    x = torch.randn((B, H, W, Cin), device="cuda", generator=g)
    assert x.numel() * 4 >= 2 ** 31
    w = _xavier(rng, (3, 3, Cin, Cout))
    b = _rand(rng, Cout) * 0.1
    pw = ops.pack_conv(_dev(w))
    assert pw.wino is not None
    y = ops.conv2d(x, pw, _dev(b))
    chunk, items = _chunk_items(B, H * W * Cin * 4)
    assert chunk < B
    for i in items:
        assert torch.equal(y[i:i + 1], ops.conv2d(x[i:i + 1].clone(), pw, _dev(b))), i
    _close(y[items[1]:items[1] + 1], OL.conv2d(x[items[1]:items[1] + 1].cpu().numpy(), w, b, (1, 1)), "chunked winograd vs oracle")
    del x, y
    torch.cuda.empty_cache()
    # 3-D encoder layer (3^3 32->32 on 64x64x32 maps): the depth-run kernel hands >= 2 GiB inputs to the implicit GEMM
    B, H, W, D, C = 129, 64, 64, 32, 32
    x = torch.randn((B, H, W, D, C), device="cuda", generator=g)
    assert x.numel() * 4 >= 2 ** 31
    w = _xavier(rng, (3, 3, 3, C, C))
    b = _rand(rng, C) * 0.1
    pw = ops.pack_conv(_dev(w))
    y = ops.conv3d(x, pw, _dev(b))
    chunk, items = _chunk_items(B, H * W * D * C * 4)
    for i in items[:4]:
        yi = ops.conv3d(x[i:i + 1].clone(), pw, _dev(b))                 # single item: the depth-run kernel
        assert float((y[i:i + 1] - yi).abs().max()) <= 2e-5 * float(yi.abs().max()), i
    _close(y[items[1]:items[1] + 1], OL.conv3d(x[items[1]:items[1] + 1].cpu().numpy(), w, b, (1, 1, 1)), "chunked conv3d vs oracle")
    del x, y
    torch.cuda.empty_cache()


# 3x3x3 stride-1 convs through the Winograd kernel (F(2x2,3x3) over H,W; depth taps as 3*Cin contiguous channels)
WINO3D_CASES = [
    (2, 8, 8, 4, 32, 32),        # res1
    (1, 5, 7, 3, 32, 32),        # ragged, D = 3
    (1, 20, 37, 1, 16, 32),      # e_conv3 widths, a single depth slice (both neighbours are padding)
    (1, 16, 16, 2, 32, 64),      # D = 2: every slice has one padded neighbour
    (2, 64, 64, 8, 32, 32),
    (2, 16, 16, 8, 16, 16),                 # the texture net's 16-wide 3-D encoder (RenderNet_Texture_Face_Normal.py:62)
]


@pytest.mark.parametrize("case", WINO3D_CASES)
def test_conv3d_winograd(case):
    from rendernet_amd import ops, _lib as L
    from scripts.wino_emulate import pack_wino
    B, H, W, D, Cin, Cout = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = _rand(rng, B, H, W, D, Cin)
    w = _xavier(rng, (3, 3, 3, Cin, Cout))
    b = _rand(rng, Cout) * 0.1
    alpha = rng.uniform(0, 0.25, Cout).astype(np.float32)
    pw = ops.pack_conv(_dev(w))
    assert pw.wino is not None
    want_u = pack_wino(w)
    assert np.abs(pw.wino.cpu().numpy() - want_u).max() <= 1e-6 * np.abs(want_u).max()
    y0 = OL.conv3d(x, w, b, (1, 1, 1))
    _close(ops.conv3d(_dev(x), pw, _dev(b)), y0, "wino3d")
    res = _rand(rng, *y0.shape)
    want = OL.prelu(y0, alpha) + torch.from_numpy(res)
    got = ops.conv3d(_dev(x), pw, _dev(b), _dev(alpha), _dev(res))
    _close(got, want, "wino3d+prelu+res")
    pd = ops.pack_conv(_dev(w))
    pd.wino = None                                          # the depth-run / implicit-GEMM kernels
    direct = ops.conv3d(_dev(x), pd, _dev(b), _dev(alpha), _dev(res))
    assert float((got - direct).abs().max()) <= 2e-5 * float(direct.abs().max())
    dp = pw.dgrad_pack(True)
    assert dp.wino is not None
    dz = _dev(_rand(rng, B, H, W, D, Cout))
    dx_w = torch.empty((B, H, W, D, Cin), device="cuda")
    L.check(L.lib().rn_conv3d_wino_fwd(L.ptr(dz), L.ptr(dp.wino), None, None, None, L.ptr(dx_w), None, B, H, W, D, Cout, Cin, 0,
                                       L.stream_ptr()), "rn_conv3d_wino_fwd (dgrad)")
    _close(dx_w, OL.conv3d_transpose(dz.cpu().numpy(), w, None, (1, 1, 1)), "wino3d dgrad vs oracle")


# 4x4 stride-1 filters through the Winograd kernel (four 2x2 sub-filters, F(2x2,2x2) each): e_conv5 / e_conv6 as convs
# (SAME padding (1,2)), e_conv7_1 as a stride-1 transposed conv (the flipped conv, padding (2,1)), and their input gradients.
WINO4_CASES = [
    (2, 16, 16, 256, 128),      # e_conv5-like, Cout % 64 == 0 (four n-tiles per wave)
    (1, 9, 11, 64, 32),         # ragged, two n-tiles
    (1, 37, 70, 16, 96),
    (3, 64, 64, 32, 64),
    (1, 16, 16, 1024, 512),
    (1, 40, 40, 32, 16),        # e_conv10's widths: one n-tile per wave
]


@pytest.mark.parametrize("case", WINO4_CASES)
def test_conv2d_winograd_4x4(case):
    from rendernet_amd import ops, _lib as L
    from scripts.wino_emulate import pack_wino4
    B, H, W, Cin, Cout = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = _rand(rng, B, H, W, Cin)
    w = _xavier(rng, (4, 4, Cin, Cout))
    b = _rand(rng, Cout) * 0.1
    alpha = rng.uniform(0, 0.25, Cout).astype(np.float32)
    pw = ops.pack_conv(_dev(w))
    assert pw.wino4 is not None
    want_u = pack_wino4(w)
    assert np.abs(pw.wino4.cpu().numpy() - want_u).max() <= 1e-6 * np.abs(want_u).max()
    y0 = OL.conv2d(x, w, b, (1, 1))
    _close(ops.conv2d(_dev(x), pw, _dev(b)), y0, "wino4")
    res = _rand(rng, *y0.shape)
    got = ops.conv2d(_dev(x), pw, _dev(b), _dev(alpha), _dev(res))
    _close(got, OL.prelu(y0, alpha) + torch.from_numpy(res), "wino4+prelu+res")
    pd = ops.pack_conv(_dev(w))
    pd.wino4 = None
    direct = ops.conv2d(_dev(x), pd, _dev(b), _dev(alpha), _dev(res))
    assert float((got - direct).abs().max()) <= 2e-5 * float(direct.abs().max())
    # the transposed form: conv2d_transpose 4x4 stride 1 with a [4,4,Cout_T,Cin_T] filter (e_conv7_1), PReLU epilogue
    wt = _xavier(rng, (4, 4, Cout, Cin))
    pt = ops.pack_conv_transpose(_dev(wt), 1)
    assert pt.wino4 is not None
    want_t = OL.prelu(OL.conv2d_transpose(x, wt, b, (1, 1)), alpha)
    _close(ops.conv2d_transpose(_dev(x), pt, _dev(b), _dev(alpha)), want_t, "wino4 transposed")
    # input gradients: a conv's is the transposed form of its own filter, and vice versa
    dz = _dev(_rand(rng, B, H, W, Cout))
    dp = pw.dgrad_pack(True)
    if dp.wino4 is not None:                                        # needs Cin % 32 == 0 (it is the output width there)
        dx = torch.empty((B, H, W, Cin), device="cuda")
        L.check(L.lib().rn_conv2d_wino4_fwd(L.ptr(dz), L.ptr(dp.wino4), None, None, None, L.ptr(dx), None, B, H, W, Cout, Cin, 1, 0,
                                            L.stream_ptr()), "rn_conv2d_wino4_fwd (dgrad)")
        _close(dx, OL.conv2d_transpose(dz.cpu().numpy(), w, None, (1, 1)), "wino4 dgrad vs oracle")


# Stride-2 4x4 transposed convs through Winograd F(2x2,2x2) per output phase (conv_wino.hip MODE 2): the decoder's e_conv7 /
# e_conv8 / e_conv9 widths (RenderNet_Shader.py:105-119), the texture heads' (RenderNet_Texture_Face_Normal.py:117-140), every
# n-tile count (Cout 16 / 32 / 64 / 128 -> NT 1 / 2 / 4 / 4 x 2 n-blocks), planes that are not multiples of the 32 x 16 block,
# more items than the persistent grid (B = 3 at 64 x 64 x 4 phases), 1 x 1 planes.
CONVT_S2_WINO_CASES = [
    (2, 8, 8, 256, 128),
    (1, 16, 16, 128, 64),
    (1, 9, 7, 64, 32),
    (1, 33, 17, 32, 16),
    (3, 64, 64, 64, 32),
    (1, 1, 1, 16, 16),
    (2, 5, 40, 48, 96),
]


@pytest.mark.parametrize("case", CONVT_S2_WINO_CASES)
def test_conv2d_transpose_s2_winograd(case):
    """rn_conv2d_transpose_s2_wino_fwd vs the oracle's tf.nn.conv2d_transpose restatement (every epilogue flavour, the
    pre-activation output), vs the direct phase kernels on the same filter, the packed filter vs its NumPy statement, and the
    dispatcher's routing."""
    from rendernet_amd import ops
    from rendernet_amd import _lib as L
    B, H, W, Cin, Cout = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = _rand(rng, B, H, W, Cin)
    w = _xavier(rng, (4, 4, Cout, Cin))
    b = _rand(rng, Cout) * 0.1
    alpha = rng.uniform(0, 0.25, Cout).astype(np.float32)
    lib = L.lib()
    pw = ops.pack_conv_transpose(_dev(w), 2)
    assert pw._wino4_kind == L.RN_PACK_CONVT_S2_WINO and pw.wino4 is not None
    # packed filter: [4 ph][Cout/NB][Cin/16][9 xi][4 kq][NB][4 r], U = G2 h G2^T, h[p][q] = w[3-pa-2p, 3-pb-2q]
    NB = 64 if Cout % 64 == 0 else 32 if Cout % 32 == 0 else 16
    G2 = np.array([[1, 0], [1, 1], [0, 1]], np.float64)
    want_u = np.zeros((4, Cout // NB, Cin // 16, 9, 4, NB, 4), np.float32)
    for ph in range(4):
        pa, pb = ph >> 1, ph & 1
        h = np.stack([np.stack([w[3 - pa - 2 * p, 3 - pb - 2 * q] for q in range(2)]) for p in range(2)]).astype(np.float64)   # [p,q,Cout,Cin]
        U = np.einsum("ip,pqoc,jq->ijoc", G2, h, G2).reshape(9, Cout // NB, NB, Cin // 16, 4, 4)    # [xi, nb, n, step, kq, r]
        want_u[ph] = U.transpose(1, 3, 0, 4, 2, 5)
    assert np.abs(pw.wino4.cpu().numpy().reshape(want_u.shape) - want_u).max() <= 1e-6 * np.abs(want_u).max()
    y0 = OL.conv2d_transpose(x, w, b, (2, 2))
    res = _rand(rng, *y0.shape)
    xd, bd, ad, rd = _dev(x), _dev(b), _dev(alpha), _dev(res)
    _close(ops.conv2d_transpose(xd, pw, bd, stride=(2, 2)), y0, "convT s2 wino")
    got = ops.conv2d_transpose(xd, pw, bd, ad, rd, stride=(2, 2))
    _close(got, OL.prelu(y0, alpha) + torch.from_numpy(res), "convT s2 wino+prelu+res")
    _close(ops.conv2d_transpose(xd, pw, None, stride=(2, 2), sigmoid=True), torch.sigmoid(OL.conv2d_transpose(x, w, None, (2, 2))), "convT s2 wino+sigmoid")
    yy, zz = torch.empty((B, 2 * H, 2 * W, Cout), device="cuda"), torch.empty((B, 2 * H, 2 * W, Cout), device="cuda")
    L.check(lib.rn_conv2d_transpose_s2_wino_fwd(L.ptr(xd), L.ptr(pw.wino4), L.ptr(bd), L.ptr(ad), None, L.ptr(yy), L.ptr(zz),
                                                B, H, W, Cin, Cout, 1, L.stream_ptr()), "rn_conv2d_transpose_s2_wino_fwd")
    _close(zz, y0, "convT s2 wino preact")
    _close(yy, OL.prelu(y0, alpha), "convT s2 wino prelu")
    # A/B against the direct phase kernels on the same filter
    p2 = ops.pack_conv_transpose(_dev(w), 2)
    p2.wino4 = None
    ref2 = ops.conv2d_transpose(xd, p2, bd, ad, rd, stride=(2, 2))
    assert float((got - ref2).abs().max()) <= 2e-5 * float(ref2.abs().max())


# The res_block_2d stack as one Winograd chain (ops.res_stack_2d; rn_winograd_output_input_transform): bit-equal to the per-layer
# launches.  (B, H, W, C, blocks, with_skip): F(6x6,3x3) on 64x64 and on a ragged map, F(4x4,3x3) on the training crops' 32x32
# and 16x16 maps, one block without a skip conv, wide tile rows (W = 100: 17 tiles -> 256-thread workgroups).
RES_STACK_CASES = [
    (2, 64, 64, 256, 2, True),
    (1, 50, 38, 256, 1, True),
    (3, 32, 32, 256, 2, True),
    (2, 16, 16, 512, 1, False),
    (1, 20, 100, 256, 1, True),
    (1, 7, 5, 256, 2, True),
]


@pytest.mark.parametrize("case", RES_STACK_CASES)
def test_res_stack_chain_is_bit_equal_to_the_layers(case, monkeypatch):
    from rendernet_amd import ops
    from rendernet_amd import _lib as L
    B, H, W, C, nb, with_skip = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = _dev(_rand(rng, B, H, W, C))
    blocks = []
    for _ in range(nb):
        blocks.append((ops.pack_conv(_dev(_xavier(rng, (3, 3, C, C)))), _dev(_rand(rng, C) * 0.1), _dev(rng.uniform(0, 0.25, C).astype(np.float32)),
                       ops.pack_conv(_dev(_xavier(rng, (3, 3, C, C)))), _dev(_rand(rng, C) * 0.1)))
    skip = (ops.pack_conv(_dev(_xavier(rng, (3, 3, C, C)))), _dev(_rand(rng, C) * 0.1), x) if with_skip else None
    monkeypatch.setattr(ops._MODE, "mode", "f32", raising=False)           # the chain exists for the exact-fp32 stage only (opt-in, measured slower)
    with torch.no_grad():
        monkeypatch.setattr(ops, "RES_STACK_FUSED", False)
        want = ops.res_stack_2d(x, blocks, skip)
        monkeypatch.setattr(ops, "RES_STACK_FUSED", True)
        n0 = ops.RES_STACK_STATS["fused"]
        got = ops.res_stack_2d(x, blocks, skip)
        assert ops.RES_STACK_STATS["fused"] == n0 + (1 if H * W >= ops.WINO43_MIN_PIXELS else 0)     # the chain really ran
    assert torch.equal(got, want), float((got - want).abs().max())
    # ... and against the oracle, so that "equal" is not "equally wrong"
    ref = torch.from_numpy(x.cpu().numpy())
    for pw1, b1, a1, pw2, b2 in blocks:
        h = OL.prelu(OL.conv2d(ref, pw1.w_tf.cpu(), b1.cpu()), a1.cpu())
        ref = ref + OL.conv2d(h, pw2.w_tf.cpu(), b2.cpu())
    if with_skip:
        ref = OL.conv2d(ref, skip[0].w_tf.cpu(), skip[1].cpu()) + torch.from_numpy(x.cpu().numpy())
    _close(got, ref, "res stack vs oracle")


def test_fused_transform_entry_matches_the_two_launches():
    """rn_winograd_output_input_transform vs rn_winograd_output_transform + rn_winograd_input_transform on the same M: V and y
    bit for bit (with / without PReLU, residual, y), and the `supported` predicate."""
    import ctypes
    from rendernet_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(5)
    for scheme, nxi, m in ((L.RN_WINO_F63, 64, 6), (L.RN_WINO_F43, 36, 4)):
        for (B, H, W, C) in ((2, 64, 64, 64), (1, 13, 29, 32)):
            th, tw = -(-H // m), -(-W // m)
            T = B * th * tw
            assert lib.rn_winograd_output_input_supported(scheme, H, W, C, 1) == 1
            Mp = _dev(_rand(rng, nxi, T, C))
            bias, alpha, res = _dev(_rand(rng, C) * 0.1), _dev(rng.uniform(0, 0.25, C).astype(np.float32)), _dev(_rand(rng, B, H, W, C))
            for act, rs, want_y in ((1, None, False), (0, res, True), (1, res, True), (0, None, False)):
                y1 = torch.empty((B, H, W, C), device="cuda")
                V1 = torch.empty((nxi, T, C), device="cuda")
                L.check(lib.rn_winograd_output_transform(scheme, L.ptr(Mp), L.ptr(bias), L.ptr(alpha), L.ptr(rs), L.ptr(y1), None,
                                                         B, H, W, C, act, L.stream_ptr()), "out")
                L.check(lib.rn_winograd_input_transform(scheme, L.ptr(y1), L.ptr(V1), B, H, W, C, 1, L.stream_ptr()), "in")
                y2 = torch.full((B, H, W, C), 7.0, device="cuda")
                V2 = torch.full((nxi, T, C), 7.0, device="cuda")
                L.check(lib.rn_winograd_output_input_transform(scheme, L.ptr(Mp), L.ptr(bias), L.ptr(alpha), L.ptr(rs),
                                                               L.ptr(y2) if want_y else None, L.ptr(V2), B, H, W, C, act, L.stream_ptr()), "outin")
                assert torch.equal(V1, V2), (scheme, B, H, W, C, act, float((V1 - V2).abs().max()))
                if want_y:
                    assert torch.equal(y1, y2)
    assert lib.rn_winograd_output_input_supported(L.RN_WINO_F63, 64, 64, 24, 1) == 0          # channels not a multiple of 16
    assert lib.rn_winograd_output_input_supported(L.RN_WINO_F44, 64, 64, 256, 1) == 0
    assert lib.rn_winograd_output_input_supported(L.RN_WINO_F63, 64, 64, 256, 2) == 0          # sigmoid: not a res-block epilogue
