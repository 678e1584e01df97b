"""CPU check of the Winograd F(2x2,3x3) kernel's layout arithmetic (csrc/conv_wino.hip) through its lane-level NumPy
restatement, scripts/wino_emulate.py: block map, LDS-DMA lane map + swizzle, fragment addresses, MFMA lane maps, the
RN_PACK_CONV_WINO / RN_PACK_CONVT_S1_WINO layouts and the epilogue addressing, against oracle/layers.py."""
import numpy as np
import pytest

from oracle import layers as OL
from scripts.wino_emulate import block_map, conv_wino4_emulated, conv_wino_emulated, pack_wino, pack_wino4


@pytest.mark.parametrize("shape", [(2, 20, 37, 32, 64), (1, 16, 16, 16, 32), (1, 5, 33, 16, 32)])
def test_emulated_kernel_matches_the_oracle_conv(shape):
    B, H, W, Cin, Cout = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) * 0.1).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    got = conv_wino_emulated(x, pack_wino(w), Cout, b)
    want = OL.conv2d(x, w, b, (1, 1)).numpy()
    assert not np.isnan(got).any()                       # every output element written exactly by some lane
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    # the transposed flavour: a stride-1 transposed conv = the input gradient of a stride-1 3x3 conv
    wt = (rng.standard_normal((3, 3, Cout, Cin)) * 0.1).astype(np.float32)       # conv_transpose layout [k,k,Cout,Cin]
    got = conv_wino_emulated(x, pack_wino(wt, transposed=True), Cout)
    want = OL.conv2d_transpose(x, wt, None, (1, 1)).numpy()
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()


@pytest.mark.parametrize("mn", [(192, 32), (192, 16), (64, 64), (7, 3), (8, 32), (9, 5), (24, 8), (300, 1)])
def test_block_map_is_a_bijection(mn):
    m, n = mn
    seen = {block_map(i, m, n) for i in range(m * n)}
    assert len(seen) == m * n and all(0 <= a < m and 0 <= b < n for a, b in seen)


def test_block_map_groups_an_xcd_round_on_few_slabs():
    # res2 at batch 24: 192 m-blocks x 32 n-blocks; the 32 workgroups one XCD holds in a round of 256 touch
    # 4 filter slabs and 8 patches, and the whole round touches only those 8 patches
    m, n = 192, 32
    for rnd in (0, 5):
        patches = set()
        for xcd in range(8):
            ids = [rnd * 256 + 8 * k + xcd for k in range(32)]
            mbs = {block_map(i, m, n)[0] for i in ids}
            nbs = {block_map(i, m, n)[1] for i in ids}
            assert len(mbs) == 8 and len(nbs) == 4
            patches |= mbs
        assert len(patches) == 8


@pytest.mark.parametrize("shape", [(1, 9, 20, 4, 16, 32), (2, 16, 16, 1, 32, 32), (1, 5, 5, 3, 16, 64), (1, 6, 33, 2, 16, 16), (1, 4, 4, 3, 32, 48)])
def test_emulated_kernel_matches_the_oracle_conv3d(shape):
    """The 3x3x3 flavour: Winograd over (H,W), the three depth taps as 3*Cin contiguous channels per depth slice, whole
    steps skipped where a depth tap is SAME padding."""
    B, H, W, D, Cin, Cout = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal((B, H, W, D, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, Cin, Cout)) * 0.1).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    got = conv_wino_emulated(x, pack_wino(w), Cout, b)
    want = OL.conv3d(x, w, b, (1, 1, 1)).numpy()
    assert not np.isnan(got).any()
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    wt = (rng.standard_normal((3, 3, 3, Cout, Cin)) * 0.1).astype(np.float32)     # conv3d_transpose layout [k,k,k,Cout,Cin]
    got = conv_wino_emulated(x, pack_wino(wt, transposed=True), Cout)
    want = OL.conv3d_transpose(x, wt, None, (1, 1, 1)).numpy()
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()


@pytest.mark.parametrize("shape", [(1, 20, 37, 32, 64), (2, 9, 5, 16, 32), (1, 7, 40, 32, 16)])
def test_emulated_4x4_variant_matches_the_oracle(shape):
    """MODE 1 of the kernel: a 4x4 filter as four 2x2 sub-filters (consecutive K steps reading the patch shifted by (2a, 2b)
    pixels), each a Winograd F(2x2,2x2); SAME conv (pad 1,2) and the stride-1 transposed conv (flipped, pad 2,1)."""
    B, H, W, Cin, Cout = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((4, 4, Cin, Cout)) * 0.1).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    got = conv_wino4_emulated(x, pack_wino4(w), Cout, 1, b)
    want = OL.conv2d(x, w, b, (1, 1)).numpy()
    assert not np.isnan(got).any() and np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    wt = (rng.standard_normal((4, 4, Cout, Cin)) * 0.1).astype(np.float32)
    got = conv_wino4_emulated(x, pack_wino4(wt, True), Cout, 2)
    want = OL.conv2d_transpose(x, wt, None, (1, 1)).numpy()
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
