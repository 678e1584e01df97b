"""Pins oracle/layers.py to TensorFlow's documented SAME / conv_transpose definitions with a
loop-level restatement on small cases."""
import numpy as np
import pytest

from oracle import layers as OL


def _same(in_size, k, s):
    out = -(-in_size // s)
    tot = max((out - 1) * s + k - in_size, 0)
    return out, tot // 2


def loop_conv_nd(x, w, stride):
    """TF conv, SAME: y[b,o,n] = sum_{t,c} x[b, o*s - pad_before + t, c] * w[t, c, n]."""
    nd = x.ndim - 2
    B, Cin, Cout = x.shape[0], x.shape[-1], w.shape[-1]
    geo = [_same(x.shape[1 + d], w.shape[d], stride[d]) for d in range(nd)]
    y = np.zeros((B,) + tuple(g[0] for g in geo) + (Cout,), np.float64)
    for o in np.ndindex(*[g[0] for g in geo]):
        for t in np.ndindex(*w.shape[:nd]):
            i = tuple(o[d] * stride[d] - geo[d][1] + t[d] for d in range(nd))
            if any(i[d] < 0 or i[d] >= x.shape[1 + d] for d in range(nd)):
                continue
            y[(slice(None),) + o] += x[(slice(None),) + i].astype(np.float64) @ w[t].astype(np.float64)
    return y


def loop_convT_nd(x, w, stride):
    """TF conv_transpose, SAME, output = in*s: the input-gradient of the SAME forward conv
    F: [out=in*s] -> [in].  y[b,o,n] += x[b,i,c] * w[t,n,c] for o = i*s - pad_before + t."""
    nd = x.ndim - 2
    B, Cin, Cout = x.shape[0], x.shape[-1], w.shape[-2]
    osz = [x.shape[1 + d] * stride[d] for d in range(nd)]
    pb = [_same(osz[d], w.shape[d], stride[d])[1] for d in range(nd)]
    y = np.zeros((B,) + tuple(osz) + (Cout,), np.float64)
    for i in np.ndindex(*x.shape[1:1 + nd]):
        for t in np.ndindex(*w.shape[:nd]):
            o = tuple(i[d] * stride[d] - pb[d] + t[d] for d in range(nd))
            if any(o[d] < 0 or o[d] >= osz[d] for d in range(nd)):
                continue
            y[(slice(None),) + o] += x[(slice(None),) + i].astype(np.float64) @ w[t].astype(np.float64).T
    return y


def test_same_pads_table():
    # k3 s1 (1,1); k5 s2 on 128 (1,2); k3 s2 on 64 (0,1); k4 s1 (1,2)
    assert OL.same_pads(64, 3, 1) == (1, 1)
    assert OL.same_pads(128, 5, 2) == (1, 2)
    assert OL.same_pads(64, 3, 2) == (0, 1)
    assert OL.same_pads(64, 4, 1) == (1, 2)


@pytest.mark.parametrize("shape,k,s", [((1, 6, 6, 6, 2), 5, (2, 2, 2)), ((2, 4, 4, 6, 3), 3, (1, 1, 2)),
                                       ((1, 5, 4, 3, 2), 3, (1, 1, 1)), ((1, 4, 4, 4, 2), 4, (1, 1, 1))])
def test_conv3d_matches_loop_definition(shape, k, s):
    rng = np.random.default_rng(1)
    x = rng.standard_normal(shape).astype(np.float32)
    w = rng.standard_normal((k, k, k, shape[-1], 3)).astype(np.float32)
    b = rng.standard_normal(3).astype(np.float32)
    got = OL.conv3d(x, w, b, s).numpy()
    want = loop_conv_nd(x, w, s) + b
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 1e-4


@pytest.mark.parametrize("shape,k,s", [((1, 6, 5, 3), 3, (1, 1)), ((2, 5, 5, 2), 4, (1, 1)), ((1, 6, 6, 2), 3, (2, 2))])
def test_conv2d_matches_loop_definition(shape, k, s):
    rng = np.random.default_rng(2)
    x = rng.standard_normal(shape).astype(np.float32)
    w = rng.standard_normal((k, k, shape[-1], 4)).astype(np.float32)
    got = OL.conv2d(x, w, None, s).numpy()
    assert np.abs(got - loop_conv_nd(x, w, s)).max() < 1e-4


@pytest.mark.parametrize("shape,s", [((1, 4, 5, 3), 1), ((2, 4, 4, 2), 2), ((1, 3, 5, 2), 2)])
def test_conv2d_transpose_matches_loop_definition(shape, s):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape).astype(np.float32)
    w = rng.standard_normal((4, 4, 3, shape[-1])).astype(np.float32)      # [kh,kw,Cout,Cin]
    got = OL.conv2d_transpose(x, w, None, (s, s)).numpy()
    want = loop_convT_nd(x, w, (s, s))
    assert got.shape == want.shape == (shape[0], shape[1] * s, shape[2] * s, 3)
    assert np.abs(got - want).max() < 1e-4


@pytest.mark.parametrize("s", [1, 2])
def test_conv3d_transpose_matches_loop_definition(s):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((1, 3, 2, 3, 2)).astype(np.float32)
    w = rng.standard_normal((4, 4, 4, 3, 2)).astype(np.float32)
    got = OL.conv3d_transpose(x, w, None, (s, s, s)).numpy()
    assert np.abs(got - loop_convT_nd(x, w, (s, s, s))).max() < 1e-4


def test_conv_transpose_is_adjoint_of_same_conv():
    """<conv(u), v> == <u, conv_transpose(v)> with the same filter (TF defines it as the gradient)."""
    rng = np.random.default_rng(5)
    for s in (1, 2):
        u = rng.standard_normal((1, 8, 8, 3)).astype(np.float32)          # forward-conv input [out size]
        w = rng.standard_normal((4, 4, 3, 5)).astype(np.float32)          # fwd filter [kh,kw,Cin=3,Cout=5]
        fu = OL.conv2d(u, w, None, (s, s)).numpy()
        v = rng.standard_normal(fu.shape).astype(np.float32)
        # the transposed conv of v uses the SAME array read as [kh,kw,Cout_T=3,Cin_T=5]
        tv = OL.conv2d_transpose(v, w, None, (s, s)).numpy()
        assert abs(float((fu * v).sum()) - float((u * tv).sum())) < 1e-2


def test_prelu_projection_sigmoid_bce():
    rng = np.random.default_rng(6)
    x = rng.standard_normal((2, 3, 3, 4, 5)).astype(np.float32)
    a = rng.uniform(0, 0.3, 5).astype(np.float32)
    want = np.maximum(x, 0) + a * np.minimum(x, 0)
    assert np.allclose(OL.prelu(x, a).numpy(), want)
    F = 20
    w = rng.standard_normal((1, 1, F, F)).astype(np.float32)
    b = rng.standard_normal(F).astype(np.float32)
    al = rng.uniform(0, 0.3, F).astype(np.float32)
    flat = x.reshape(2, 3, 3, F)                  # f = d*C + c  (tools/layer_util.py:19-20)
    y = flat @ w[0, 0] + b
    assert np.allclose(OL.projection_unit(x, w, b, al).numpy(), np.maximum(y, 0) + al * np.minimum(y, 0), atol=1e-5)
    p = rng.uniform(0.01, 0.99, (2, 4, 4, 1)).astype(np.float32)
    t = rng.uniform(0, 1, (2, 4, 4, 1)).astype(np.float32)
    want = np.mean(-np.sum(t * np.log(1e-6 + p) + (1 - t) * np.log(1e-6 + 1 - p), axis=(1, 2, 3)))
    assert abs(float(OL.bce_loss(p, t)) - want) < 1e-4


@pytest.mark.parametrize("shape,k,s", [((1, 9, 7, 3), 3, (1, 1)), ((2, 8, 8, 2), 4, (1, 1)), ((1, 10, 9, 2), 4, (2, 2)),
                                       ((1, 6, 7, 5, 2), 3, (1, 1, 2)), ((1, 8, 8, 8, 1), 5, (2, 2, 2))])
def test_conv_matches_scipy_correlate(shape, k, s):
    """A third, independent statement of the SAME conv: zero-pad by TF's (pad_before, pad_after) table, full 'valid'
    cross-correlation with scipy.signal.correlate per channel pair, then keep every s-th output."""
    from scipy.signal import correlate
    rng = np.random.default_rng(sum(shape) + k)
    nd = len(shape) - 2
    cin, cout = shape[-1], 3
    x = rng.standard_normal(shape).astype(np.float32)
    w = rng.standard_normal((k,) * nd + (cin, cout)).astype(np.float32)
    got = (OL.conv2d if nd == 2 else OL.conv3d)(x, w, None, s).numpy()
    pads = [OL.same_pads(shape[1 + d], k, s[d]) for d in range(nd)]
    xp = np.pad(x.astype(np.float64), [(0, 0)] + [tuple(p) for p in pads] + [(0, 0)])
    want = np.zeros(got.shape, np.float64)
    for b in range(shape[0]):
        for n in range(cout):
            acc = 0.0
            for c in range(cin):
                acc = acc + correlate(xp[b, ..., c], w[..., c, n].astype(np.float64), mode="valid")
            want[b, ..., n] = acc[tuple(slice(None, None, s[d]) for d in range(nd))]
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()


def test_tf_conv2d_documentation_example():
    """The worked example of the `tf.nn.conv2d` API documentation (TF 2.x docstring: a 5x5x1 image, a [2,2,1,2] filter, VALID
    padding, 4x4x2 result).  TensorFlow cannot be run or fetched in this container, so the numbers below are typed in from
    memory of that page -- NOT a pin of the oracle against TensorFlow, only a check of the two conventions the example
    exercises: tf.nn.conv2d is a cross-correlation (no filter flip) and the filter layout is [kh, kw, Cin, Cout].  Every value
    is also a sum one can do by hand, e.g. out[3,1,0] = 2*2 + 3*0 + 0*0 + 1*3 = 7.  The oracle's conv is SAME-padded; for a
    2x2 filter SAME pads only after the last row / column, so its top-left 4x4 block is the VALID result."""
    x = np.array([[[[2], [1], [2], [0], [1]], [[1], [3], [2], [2], [3]], [[1], [1], [3], [3], [0]],
                   [[2], [2], [0], [1], [1]], [[0], [0], [3], [1], [2]]]], np.float32)
    k = np.array([[[[2, 0.1]], [[3, 0.2]]], [[[0, 0.3]], [[1, 0.4]]]], np.float32)
    want = np.array([[[10, 1.9], [10, 2.2], [6, 1.6], [6, 2.0]], [[12, 1.4], [15, 2.2], [13, 2.7], [13, 1.7]],
                     [[7, 1.7], [11, 1.3], [16, 1.3], [7, 1.0]], [[10, 0.6], [7, 1.4], [4, 1.5], [7, 1.4]]], np.float32)
    got = OL.conv2d(x, k, None, (1, 1)).numpy()
    assert got.shape == (1, 5, 5, 2)
    assert np.abs(got[0, :4, :4] - want).max() <= 1e-6
