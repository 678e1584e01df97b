"""CPU checks of the Winograd F(4x4,3x3) / F(4x4,4x4) / F(6x6,3x3) path's host-side statements (no GPU):
the generated matrices (scripts/gen_wino_mats.py -> rendernet_amd/csrc/wino_mats.h) satisfy the minimal-filtering identity
exactly, the committed header is what the script prints, and a NumPy emulation of the three launches -- input transform,
one GEMM per xi over the PACKED filter layout, output transform -- reproduces the oracle conv (forward, stride-1 transposed
conv through the transposed pack) and its filter gradient (the wgrad identities of conv_wino43_wgrad.hip)."""
import os
import subprocess
import sys
from fractions import Fraction as Fr

import numpy as np
import pytest
import torch

from oracle import layers as OL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.gen_wino_mats import SCHEMES, winograd_mats  # noqa: E402


def _mats(scheme):
    m, r, pts = SCHEMES[scheme]
    AT, G, BT = winograd_mats(m, r, pts)
    f = lambda M: np.array([[float(v) for v in row] for row in M], np.float64)
    return m, r, f(AT), f(G), f(BT), (AT, G, BT)


@pytest.mark.parametrize("scheme", ["F43", "F44", "F63"])
def test_identity_is_exact_in_rational_arithmetic(scheme):
    m, r, _, _, _, (AT, G, BT) = _mats(scheme)
    a = m + r - 1
    rng = np.random.default_rng(0)
    d = [Fr(int(v), 7) for v in rng.integers(-50, 50, a)]
    g = [Fr(int(v), 3) for v in rng.integers(-50, 50, r)]
    Gg = [sum(G[i][j] * g[j] for j in range(r)) for i in range(a)]
    Bd = [sum(BT[i][j] * d[j] for j in range(a)) for i in range(a)]
    y = [sum(AT[i][k] * Gg[k] * Bd[k] for k in range(a)) for i in range(m)]
    want = [sum(d[i + j] * g[j] for j in range(r)) for i in range(m)]
    assert y == want                                       # correlation y[i] = sum_j d[i+j] g[j], exactly


def test_committed_header_is_the_generators_output():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_wino_mats.py")], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(ROOT, "rendernet_amd", "csrc", "wino_mats.h")).read()


def _pack(w, G, transposed=False):
    """wino_pack_kernel's layout: [nxi][Cout/256][Cin/4][256][4]; transposed: conv_transpose filter [R,R,Cout,Cin], taps flipped."""
    if transposed:
        w = w[::-1, ::-1].transpose(0, 1, 3, 2)
    Cin, Cout = w.shape[2], w.shape[3]
    nxi = G.shape[0] ** 2
    U = np.einsum("ia,abck,jb->ijck", G, w.astype(np.float64), G).reshape(nxi, Cin, Cout)
    out = np.empty((nxi, Cout // 256, Cin // 4, 256, 4))
    for nb in range(Cout // 256):
        out[:, nb] = U[:, :, nb * 256:(nb + 1) * 256].reshape(nxi, Cin // 4, 4, 256).transpose(0, 1, 3, 2)
    return out


def _three_launches(x, packed, AT, BT, pad_lo, Cout):
    """NumPy statement of conv_wino43.hip: V = B^T d B per m x m-output tile (m = A^T's rows), M[xi] = V[xi] . U[xi] read from the packed panels,
    Y = A^T m A.  x [B,H,W,Cin] -> [B,H,W,Cout] (float64)."""
    B, H, W, Cin = x.shape
    a, m = BT.shape[0], AT.shape[0]
    th, tw = (H + m - 1) // m, (W + m - 1) // m
    xp = np.zeros((B, m * th + a - m + 8, m * tw + a - m + 8, Cin))
    xp[:, pad_lo:pad_lo + H, pad_lo:pad_lo + W] = x
    tiles = np.stack([xp[b, m * ty:m * ty + a, m * tx:m * tx + a] for b in range(B) for ty in range(th) for tx in range(tw)])
    V = np.einsum("ia,tabc,jb->ijtc", BT, tiles, BT).reshape(a * a, -1, Cin)                 # [nxi][T][Cin]
    nxi, T = V.shape[0], V.shape[1]
    M = np.empty((nxi, T, Cout))
    for nb in range(Cout // 256):
        panel = packed[:, nb]                                                                  # [nxi][Cin/4][256][4]
        Ublk = panel.transpose(0, 1, 3, 2).reshape(nxi, Cin, 256)                              # k = 4*kg + r
        M[:, :, nb * 256:(nb + 1) * 256] = np.einsum("xtc,xcn->xtn", V, Ublk)
    Y = np.einsum("pi,ijtn,qj->tpqn", AT, M.reshape(a, a, T, Cout), AT)                        # [T][m][m][Cout]
    y = Y.reshape(B, th, tw, m, m, Cout).transpose(0, 1, 3, 2, 4, 5).reshape(B, m * th, m * tw, Cout)
    return y[:, :H, :W]


@pytest.mark.parametrize("scheme,shape", [("F43", (2, 9, 6, 8, 256)), ("F44", (1, 7, 10, 4, 256)), ("F63", (2, 13, 7, 8, 256))])
def test_three_launch_emulation_matches_the_oracle_conv(scheme, shape):
    m, r, AT, G, BT, _ = _mats(scheme)
    B, H, W, Cin, Cout = shape
    rng = np.random.default_rng(5)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((r, r, Cin, Cout)) * 0.1).astype(np.float32)
    got = _three_launches(x.astype(np.float64), _pack(w, G), AT, BT, 1, Cout)                 # SAME conv: one row/col before
    want = OL.conv2d(x, w, None, (1, 1)).numpy()
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
    # stride-1 transposed conv (= the conv's input gradient) through the transposed pack: pad_lo = R - 2
    wt = (rng.standard_normal((r, r, Cout, Cin)) * 0.1).astype(np.float32)                     # conv_transpose filter [R,R,Cout_T,Cin_T]
    xt = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    if Cout % 256 == 0:
        got_t = _three_launches(xt.astype(np.float64), _pack(wt, G, transposed=True), AT, BT, r - 2, Cout)
        want_t = OL.conv2d_transpose(xt, wt, None, (1, 1)).numpy()
        assert np.abs(got_t - want_t).max() <= 2e-5 * np.abs(want_t).max()


@pytest.mark.parametrize("scheme", ["F43", "F44", "F63"])
def test_wgrad_identity_matches_autograd(scheme):
    """dg = G^T [ sum_tiles (B^T d B) .* (A dY A^T) ] G  (conv_wino43_wgrad.hip) vs torch autograd over the oracle conv."""
    m, r, AT, G, BT, _ = _mats(scheme)
    B, H, W, Cin, Cout = 2, 6, 9, 3, 5
    rng = np.random.default_rng(8)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    dz = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
    wt = torch.zeros(r, r, Cin, Cout, requires_grad=True)
    OL.conv2d(torch.from_numpy(x), wt).backward(torch.from_numpy(dz))
    a = BT.shape[0]
    th, tw = (H + m - 1) // m, (W + m - 1) // m
    xp = np.zeros((B, m * th + a, m * tw + a, Cin)); xp[:, 1:1 + H, 1:1 + W] = x
    zp = np.zeros((B, m * th, m * tw, Cout)); zp[:, :H, :W] = dz
    dU = np.zeros((a, a, Cin, Cout))
    for b in range(B):
        for ty in range(th):
            for tx in range(tw):
                V = np.einsum("ia,abc,jb->ijc", BT, xp[b, m * ty:m * ty + a, m * tx:m * tx + a], BT)
                dM = np.einsum("pi,pqn,qj->ijn", AT, zp[b, m * ty:m * ty + m, m * tx:m * tx + m], AT)
                dU += V[:, :, :, None] * dM[:, :, None, :]
    dw = np.einsum("ia,ijcn,jb->abcn", G, dU, G)
    want = wt.grad.numpy()
    assert np.abs(dw - want).max() <= 2e-5 * np.abs(want).max()
