"""Pins the training-step oracle (oracle/train.py) on CPU: its gradients against central finite
differences of its own loss in float64, its Adam against a scalar hand computation of TF's update rule,
the crop against tools/model_util.py:95-99, and the host-side schedule helpers of rendernet_amd/train.py."""
import math

import numpy as np
import torch

from oracle import rendernet as ON
from oracle import train as OT
from oracle import layers as OL


def _tiny():
    from rendernet_amd.shader import tiny_spec, init_shader_weights
    spec = tiny_spec(1)
    w = init_shader_weights(spec, seed=1234, perturb=True)
    rng = np.random.default_rng(0)
    x = (rng.random((1, 8, 8, 32, 1)) < 0.3).astype(np.float32)
    t = rng.random((1, 32, 32, 1)).astype(np.float32)
    return spec, w, x, t


def test_gradients_match_finite_differences():
    spec, w, x, t = _tiny()
    n = (spec.n_res1, spec.n_res2, spec.n_res3)
    loss, grads, _ = OT.loss_and_grads(x, t, w, *n, dtype=np.float64)

    def f(wd):
        with torch.no_grad():
            wt = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)) for k, v in wd.items()}
            pred = ON.rendernet_forward_torch(torch.from_numpy(x.astype(np.float64)), wt, None, *n)
            return float(OL.bce_loss(pred, torch.from_numpy(t.astype(np.float64))).item())

    assert abs(f(w) - loss) < 1e-9 * abs(loss)
    rng = np.random.default_rng(1)
    # one entry of every kind of variable along the depth of the net (3-D conv, alpha, bias, 1x1, 2-D conv,
    # transposed convs incl. the head)
    for name in ["encoder/e_conv1/e_conv1/weights", "encoder/e_conv2/alpha", "encoder/res1_1/con1_3X3/biases",
                 "encoder/projection_unit/Conv/weights", "encoder/res2_1/conv2_3x3/weights",
                 "encoder/e_conv5/e_conv5/weights", "encoder/e_conv7/e_conv7/weights", "encoder/e_conv9/alpha",
                 "encoder/e_conv11/weights", "encoder/e_conv11/biases"]:
        g = grads[name]
        idx = np.unravel_index(int(np.argmax(np.abs(g))), g.shape)        # a well-conditioned entry
        h = 1e-5 * max(1.0, abs(float(w[name][idx])))
        wp = {k: (v.astype(np.float64).copy() if k == name else v) for k, v in w.items()}
        wm = {k: (v.astype(np.float64).copy() if k == name else v) for k, v in w.items()}
        wp[name][idx] += h
        wm[name][idx] -= h
        fd = (f(wp) - f(wm)) / (2 * h)
        assert abs(fd - g[idx]) <= 1e-5 * max(abs(fd), abs(g[idx])) + 1e-9, (name, fd, g[idx])


def test_float32_gradients_close_to_float64():
    spec, w, x, t = _tiny()
    n = (spec.n_res1, spec.n_res2, spec.n_res3)
    l32, g32, _ = OT.loss_and_grads(x, t, w, *n)
    l64, g64, _ = OT.loss_and_grads(x, t, w, *n, dtype=np.float64)
    assert abs(l32 - l64) <= 1e-5 * abs(l64)
    for k in g64:
        assert np.abs(g32[k] - g64[k]).max() <= 1e-4 * np.abs(g64[k]).max() + 1e-9, k


def test_adam_is_tf_update_rule():
    """Hand computation of tf.train.AdamOptimizer(lr, beta1=0.5) with staircase decay for two steps."""
    opt = OT.Adam(e_eta=0.1, decay_steps=1, beta1=0.5, beta2=0.999, epsilon=1e-8)
    p = {"x": np.array([1.0, -2.0], np.float32)}
    g1 = {"x": np.array([0.5, -4.0], np.float32)}
    p1 = opt.apply(p, g1)["x"]
    # t=1: lr=0.1, m=0.5g, v=0.001g^2, lr_t = 0.1*sqrt(0.001)/0.5 -> step = lr_t*m/(sqrt(v)+eps) = 0.1*sign(g) (to eps)
    assert np.allclose(p1, [1.0 - 0.1, -2.0 + 0.1], atol=1e-6)
    g2 = {"x": np.array([0.5, 4.0], np.float32)}
    p2 = opt.apply({"x": p1}, g2)["x"]
    lr = 0.1 * 0.96                                                     # staircase: floor(1/1) = 1
    lr_t = lr * math.sqrt(1 - 0.999 ** 2) / (1 - 0.5 ** 2)
    m = np.array([0.5 * 0.25 + 0.5 * 0.5, 0.5 * -2.0 + 0.5 * 4.0])
    v = np.array([0.999 * 0.001 * 0.25 + 0.001 * 0.25, 0.999 * 0.001 * 16 + 0.001 * 16])
    assert np.allclose(p2, p1 - lr_t * m / (np.sqrt(v) + 1e-8), atol=1e-6)


def test_crop_matches_reference_indexing():
    vox = np.arange(2 * 8 * 8 * 4, dtype=np.float32).reshape(2, 8, 8, 4, 1)
    img = np.arange(2 * 32 * 32, dtype=np.float32).reshape(2, 32, 32, 1)
    v, i = OT.crop_voxel_image(vox, img, (2, 5), 3)
    assert v.shape == (2, 3, 3, 4, 1) and i.shape == (2, 12, 12, 1)
    assert np.array_equal(v, vox[:, 2:5, 5:8]) and np.array_equal(i, img[:, 8:20, 20:32])


def test_host_schedule_helpers():
    from rendernet_amd.train import exponential_decay, adam_lr_t, plan_buckets
    assert exponential_decay(1e-5, 0, 100000) == 1e-5
    assert exponential_decay(1e-5, 99999, 100000) == 1e-5
    assert abs(exponential_decay(1e-5, 250000, 100000) - 1e-5 * 0.96 ** 2) < 1e-18
    assert abs(exponential_decay(1.0, 150, 100, staircase=False) - 0.96 ** 1.5) < 1e-12
    assert abs(adam_lr_t(0.1, 1, 0.5, 0.999) - 0.1 * math.sqrt(0.001) / 0.5) < 1e-12
    assert OT.exponential_decay(3e-4, 250000, 100000) == exponential_decay(3e-4, 250000, 100000)
    # buckets: contiguous, cover every parameter once, follow the completion order, respect the size cap
    sizes = {"a": 10, "b": 7, "c": 100, "d": 3, "e": 50}
    layout, off = {}, 0
    for n, k in sizes.items():
        layout[n] = (off, k)
        off += (k + 3) // 4 * 4
    order = list(reversed(list(sizes)))
    b = plan_buckets(layout, order, 64)
    assert [n for _, _, ns in b for n in ns] == order
    assert all(lo == min(layout[n][0] for n in ns) for lo, _, ns in b)
    assert all(hi - lo <= 64 or len(ns) == 1 for lo, hi, ns in b)
    spans = sorted((lo, hi) for lo, hi, _ in b)
    assert spans[0][0] == 0 and spans[-1][1] == off and all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))


def test_torch_resampler_equals_numpy_oracle_and_is_differentiable():
    """oracle/texture_train.py::resample_torch (the differentiable restatement used for the texture net's training
    oracle) reproduces oracle/resample.py bit for bit, and its gradient is the scatter of oracle/resample.py's backward."""
    from oracle import resample as OR
    from oracle import texture_train as TT
    rng = np.random.default_rng(2)
    S, N, C = 8, 16, 3
    vox = rng.standard_normal((2, S, S, S, C)).astype(np.float32)
    M = OR.inverse_affine(np.array([[1.0, 0.6, 0.9], [2.5, 0.2, 1.2]], np.float32), S, N)
    for mode in ("tf", "ordered"):
        got = TT.resample_torch(torch.from_numpy(vox), M, N, mode).numpy()
        assert np.array_equal(got, OR.resampling_affine(vox, M, N, mode))
    vt = torch.from_numpy(vox).requires_grad_(True)
    dout = rng.standard_normal((2, N, N, N, C)).astype(np.float32)
    TT.resample_torch(vt, M, N, "ordered").backward(torch.from_numpy(dout))
    dv, _ = OR.resampling_affine_bwd(vox, M, dout, N)
    assert np.abs(vt.grad.numpy() - dv).max() <= 1e-4 * np.abs(dv).max()
    img = TT.to_image_layout(torch.from_numpy(vox)).numpy()
    assert np.array_equal(img, OR.transform_voxel_to_match_image(vox))
