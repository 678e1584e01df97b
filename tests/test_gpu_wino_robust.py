"""Rounding robustness of the Winograd routes on hostile activation statistics.  -m gpu.

Round 2 showed the F(6x6,3x3) error budget on N(0,1) inputs and Xavier filters only.  Here the res2 layer shape (3x3,
1024 -> 1024, 64x64 map; tools/layer_util.py:91-105, RenderNet_Shader.py:71-84) is fed what a TRAINED net feeds it -- large
positive means, log-normal per-channel gains, sparse spikes, filters that push the outputs to +-8 and beyond -- plus 21 stacked
convs whose activations the scheme under test produced itself, and every route (direct implicit GEMM, fused F(2x2,3x3),
F(4x4,3x3), F(6x6,3x3)) is measured against a FLOAT64 CPU conv of the same fp32 operands.
Bars: max|got - f64| <= 1e-4 * max|y| for F(6x6,3x3) (the per-tap bar of the net tests is 2e-4; measured worst 4.0e-5,
profiles/r03a_wino_robustness.md), 3e-5 for the other routes.  Then the routing guard (ops.WINO63_CHECK_TOL)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from scripts import robust_util as RU

pytestmark = pytest.mark.gpu
BAR = {"direct": 3e-5, "f22": 3e-5, "f43": 3e-5, "f63": 1e-4, "f43s": 3e-5, "f63s": 1e-4, "f43h": 3e-5, "f63h": 1e-4}      # the split stages: the SAME bars
C = 1024


def conv_f64(x, w, b=None):
    xn = torch.as_tensor(x).double().permute(0, 3, 1, 2)
    y = F.conv2d(xn, torch.as_tensor(w).double().permute(3, 2, 0, 1), None, 1, 1).permute(0, 2, 3, 1)
    return (y + torch.as_tensor(b).double() if b is not None else y).contiguous()


CASES = list(range(7))


@pytest.mark.parametrize("k", CASES)
def test_single_layer_hostile_statistics(k):
    rng = np.random.default_rng(20260926)
    name, x, w, b = list(RU.hostile_inputs(rng, 1, 64, 64, C, C))[k]
    want = conv_f64(x, w, b)
    ymax = float(want.abs().max())
    xd, wd, bd = (torch.as_tensor(a).cuda() for a in (x, w, b))
    errs = {}
    for scheme in RU.SCHEMES:
        got = RU.conv_with_scheme(xd, wd, bd, scheme)
        errs[scheme] = float((got.cpu().double() - want).abs().max()) / ymax
    print("%s: max|y| %.3g  " % (name, ymax) + "  ".join("%s %.2e" % kv for kv in errs.items()))
    for scheme, e in errs.items():
        assert e <= BAR[scheme], "%s, %s: %.3g * max|y| > %.3g" % (name, scheme, e, BAR[scheme])


def test_21_stacked_convs():
    """10 res_block_2d + res2_skip with positive biases / small PReLU slopes: the error of a scheme compounds through its own
    activations.  24x24 map (16 F(6x6) tiles per image, ragged: 24 = 4 x 6) so that the float64 stack stays a few seconds."""
    rng = np.random.default_rng(7)
    net = RU.res_stack_weights(rng, C, n_blocks=10)
    x0 = (np.abs(rng.standard_normal((1, 24, 24, C))) + 0.5).astype(np.float32)
    want = RU.res_stack_f64(x0, net, conv_f64)
    ymax = float(want.abs().max())
    for scheme in RU.SCHEMES:
        got = RU.res_stack_gpu(torch.as_tensor(x0).cuda(), net, scheme)
        e = float((got.cpu().double() - want).abs().max()) / ymax
        print("21 stacked convs, %s: %.2e * max|y| (max|y| %.3g)" % (scheme, e, ymax))
        assert e <= BAR[scheme], (scheme, e)


def test_wino63_self_check_demotes_and_passes(monkeypatch, gemm_mode):
    """ops.WINO63_CHECK_TOL, in every multiply-stage mode (the default mode's check is what RenderNet_demo.py --weights runs): the first
    F(6x6,3x3) launch of a filter -- in the ACTIVE mode -- is repeated with F(4x4,3x3) on the exact-fp32 stage; an impossible tolerance
    demotes the filter (the result IS the exact F(4x4,3x3) one, later launches take that route), a sane one keeps the mode's F(6x6,3x3)
    route and checks only once."""
    from rendernet_amd import ops
    rng = np.random.default_rng(3)
    x = torch.as_tensor((np.abs(rng.standard_normal((2, 64, 64, 256))) + 3.0).astype(np.float32)).cuda()
    w = torch.as_tensor(RU.xavier(rng, (3, 3, 256, 256))).cuda()
    b = torch.as_tensor((0.1 * rng.standard_normal(256)).astype(np.float32)).cuda()
    y43 = RU.conv_with_scheme(x, w, b, "f43")                                   # exact fp32: what a demoted filter computes
    y63 = RU.conv_with_scheme(x, w, b, "f63" + {"f32": "", "split": "s", "split16": "h"}[gemm_mode])
    assert ops.gemm_mode_now() == gemm_mode                                     # (conv_with_scheme leaves the mode alone)
    assert not torch.equal(y43, y63)
    with torch.no_grad():
        # impossible tolerance -> demoted
        monkeypatch.setattr(ops, "WINO63_CHECK_TOL", 1e-12)
        n0 = len(ops.WINO63_DEMOTED)
        pw = ops.pack_conv(w)
        assert ops._wino_scheme(pw, 64, 64) == "f63"
        got = ops.conv2d(x, pw, b)
        assert torch.equal(got, y43)
        assert len(ops.WINO63_DEMOTED) == n0 + 1 and ops.WINO63_DEMOTED[-1]["cin"] == 256 and ops.WINO63_DEMOTED[-1]["mode"] == gemm_mode
        assert ops._wino_scheme(pw, 64, 64) == "f43" and torch.equal(ops.conv2d(x, pw, b), y43)
        # the documented bar -> kept, verdict cached
        monkeypatch.setattr(ops, "WINO63_CHECK_TOL", 2e-4)
        pw2 = ops.pack_conv(w)
        assert torch.equal(ops.conv2d(x, pw2, b), y63) and pw2._wino63_verdict is True
        assert len(ops.WINO63_DEMOTED) == n0 + 1
        monkeypatch.setattr(ops, "WINO63_CHECK_TOL", 1e-12)          # already checked: not checked again
        assert torch.equal(ops.conv2d(x, pw2, b), y63)
        # off (the default): no verdict is formed
        monkeypatch.setattr(ops, "WINO63_CHECK_TOL", None)
        pw3 = ops.pack_conv(w)
        assert torch.equal(ops.conv2d(x, pw3, b), y63) and getattr(pw3, "_wino63_verdict", None) is None


def test_renderer_validate_winograd(fixtures_vox):
    """Renderer.validate_winograd on the bench weights: no filter of the seeded net is demoted at the 2e-4 bar, and the
    render after the check equals the render before it."""
    from conftest import demo_pose
    from rendernet_amd.shader import Renderer, ShaderSpec, init_shader_weights
    spec = ShaderSpec().check()
    r = Renderer(spec, init_shader_weights(spec, seed=1234, perturb=True))
    vox, pose = fixtures_vox[1:2], demo_pose()[None]
    before = r.render(vox, pose).clone()
    r2 = Renderer(spec, init_shader_weights(spec, seed=1234, perturb=True))
    assert r2.validate_winograd(vox, pose, tol=2e-4) == []
    assert torch.equal(r2.render(vox, pose), before)


@pytest.mark.parametrize("k", [0, 1, 4])
def test_f44_layers_on_hostile_statistics(k):
    """The 4x4 layers (e_conv5: 1024 -> 512, RenderNet_Shader.py:86-88) run F(4x4,4x4) by default: the same hostile inputs, the
    route against a float64 conv with TF's SAME padding for an even filter (1 before, 2 after), next to the F(2x2,2x2)x4 kernel
    and the direct one."""
    from rendernet_amd import ops
    rng = np.random.default_rng(99)
    name, x, _, _ = list(RU.hostile_inputs(rng, 1, 64, 64, C, 512))[k]
    w = RU.xavier(rng, (4, 4, C, 512))
    b = (0.1 * rng.standard_normal(512)).astype(np.float32)
    xn = F.pad(torch.as_tensor(x).double().permute(0, 3, 1, 2), (1, 2, 1, 2))
    want = (F.conv2d(xn, torch.as_tensor(w).double().permute(3, 2, 0, 1)).permute(0, 2, 3, 1) + torch.as_tensor(b).double()).contiguous()
    ymax = float(want.abs().max())
    xd, wd, bd = (torch.as_tensor(a).cuda() for a in (x, w, b))
    errs = {}
    for scheme in ("f44", "f44s", "f22x4", "direct"):
        pw = ops.pack_conv(wd)
        if scheme not in ("f44", "f44s"):
            pw.wino43 = None
        if scheme == "direct":
            pw.wino4 = None
        with torch.no_grad(), ops.gemm_mode("split" if scheme == "f44s" else "f32"):
            got = ops.conv2d(xd, pw, bd)
        errs[scheme] = float((got.cpu().double() - want).abs().max()) / ymax
    print("%s (4x4 filter): max|y| %.3g  " % (name, ymax) + "  ".join("%s %.2e" % kv for kv in errs.items()))
    assert errs["f44"] <= 1e-4 and errs["f44s"] <= 1e-4 and errs["f22x4"] <= 3e-5 and errs["direct"] <= 3e-5, errs
