"""RenderNet Phong-shader graph and its runner on the MI355X path.

Mirrors RenderNet_Shader.py:32-131 (`RenderNet`) and :135-156 (graph: resample -> transform ->
crop -> net), plus the `sess.run` contract of RenderNet_demo.py:47-51 (`Renderer.run` accepts
the reference tensor names "real_model_in:0", "view_name:0", "patch_size:0", "is_training:0" and
fetches "encoder/output:0").  Weights are keyed by the TF variable names.
"""
from dataclasses import dataclass, field

import numpy as np
import torch

from . import variables as V
from .tools import layer_util as LU
from .tools.layer_util import conv3d, res_block_2d, res_block_3d, projection_unit, keep_prob
from .tools.resampling_voxel_grid import rotation_resampling_to_image
from .variables import xavier_initializer, constant_initializer


@dataclass
class ShaderSpec:
    """Channel plan of the Phong-shader net.  Defaults = the reference (RenderNet_Shader.py:36-129).
    The projection width is depth/4 * c3 and must equal `w_res2`."""
    size: int = 64                 # source voxel grid (model_in placeholder :140)
    new_size: int = 128            # resampled grid (new_res :136)
    in_ch: int = 1
    c1: int = 8
    c2: int = 16
    c3: int = 32
    n_res1: int = 10
    w_res2: int = 32 * 32
    n_res2: int = 10
    w5: int = 32 * 16
    n_res3: int = 5
    w6: int = 32 * 8
    w7: int = 32 * 4
    w7_1: int = 32 * 4
    w8: int = 32 * 2
    w9: int = 32
    w10: int = 16
    out_ch: int = 1                # is_greyscale "True" -> 1, else 3 (:125-130)

    def check(self):
        if (self.new_size // 4) * self.c3 != self.w_res2:
            raise ValueError("projection width %d != res2 width %d" % ((self.new_size // 4) * self.c3, self.w_res2))
        return self


def tiny_spec(out_ch=1):
    """A reduced net with the same structure for fast parity tests (16^3 -> 32^3 -> 128^2)."""
    return ShaderSpec(size=16, new_size=32, c1=8, c2=16, c3=32, n_res1=2, w_res2=256, n_res2=2, w5=128, n_res3=1,
                      w6=64, w7=32, w7_1=32, w8=32, w9=16, w10=16, out_ch=out_ch).check()


def stress_spec(out_ch=1):
    """BASELINE config 5: 128^3 voxels -> 256^3 -> 1024x1024.  The reference net hard-codes its widths for a
    depth-128 input (RenderNet_Shader.py:71-129); at depth 256 the projection unit emits 64*32 = 2048
    features, so every 2-D width doubles (16 280 GMAC/frame, 3.79 GB of weights)."""
    return ShaderSpec(size=128, new_size=256, w_res2=2048, w5=1024, w6=512, w7=256, w7_1=256, w8=128, w9=64,
                      w10=32, out_ch=out_ch).check()


def shader_variable_shapes(spec):
    """[(tf_name, shape, kind)] for every variable of the graph, in creation order.
    kind: 'w3' hand-rolled conv3d filter (xavier), 'b3' its bias (0.001), 'ws' slim filter (xavier),
    'bs' slim bias (0), 'a' PReLU alpha (0)."""
    s = spec
    out = []
    e = "encoder/"

    def c3d(scope, name, k, cin, cout, alpha=True):
        out.append((e + "%s/%s/weights" % (scope, name), [k, k, k, cin, cout], 'w3'))
        out.append((e + "%s/%s/biases" % (scope, name), [cout], 'b3'))
        if alpha:
            out.append((e + "%s/alpha" % scope, [cout], 'a'))

    c3d("e_conv1", "e_conv1", 5, s.in_ch, s.c1)
    c3d("e_conv2", "e_conv2", 3, s.c1, s.c2)
    c3d("e_conv3", "e_conv3", 3, s.c2, s.c3)
    for k in range(1, s.n_res1 + 1):
        sc = "res1_%d" % k
        out.append((e + sc + "/alpha", [s.c3], 'a'))
        for n in ("con1_3X3", "conv2_3x3"):
            out.append((e + "%s/%s/weights" % (sc, n), [3, 3, 3, s.c3, s.c3], 'w3'))
            out.append((e + "%s/%s/biases" % (sc, n), [s.c3], 'b3'))
    c3d("res1_skip", "con1_3X3", 3, s.c3, s.c3, alpha=False)
    F = s.w_res2
    out.append((e + "projection_unit/Conv/weights", [1, 1, F, F], 'ws'))
    out.append((e + "projection_unit/Conv/biases", [F], 'bs'))
    out.append((e + "projection_unit/alpha", [F], 'a'))

    def res2d(prefix, n, width):
        for k in range(1, n + 1):
            sc = "%s_%d" % (prefix, k)
            out.append((e + sc + "/alpha", [width], 'a'))
            for nm in ("con1_3X3", "conv2_3x3"):
                out.append((e + "%s/%s/weights" % (sc, nm), [3, 3, width, width], 'ws'))
                out.append((e + "%s/%s/biases" % (sc, nm), [width], 'bs'))
        out.append((e + "%s_skip/con1_3X3/weights" % prefix, [3, 3, width, width], 'ws'))
        out.append((e + "%s_skip/con1_3X3/biases" % prefix, [width], 'bs'))

    res2d("res2", s.n_res2, F)
    out.append((e + "e_conv5/e_conv5/weights", [4, 4, F, s.w5], 'ws'))
    out.append((e + "e_conv5/e_conv5/biases", [s.w5], 'bs'))
    out.append((e + "e_conv5/alpha", [s.w5], 'a'))
    res2d("res3", s.n_res3, s.w5)
    out.append((e + "e_conv6/e_conv6/weights", [4, 4, s.w5, s.w6], 'ws'))
    out.append((e + "e_conv6/e_conv6/biases", [s.w6], 'bs'))
    out.append((e + "e_conv6/alpha", [s.w6], 'a'))
    cin = s.w6
    for name, cout in (("e_conv7", s.w7), ("e_conv7_1", s.w7_1), ("e_conv8", s.w8), ("e_conv9", s.w9),
                       ("e_conv10", s.w10)):
        out.append((e + "%s/%s/weights" % (name, name), [4, 4, cout, cin], 'ws'))     # [kh,kw,Cout,Cin]
        out.append((e + "%s/%s/biases" % (name, name), [cout], 'bs'))
        out.append((e + "%s/alpha" % name, [cout], 'a'))
        cin = cout
    out.append((e + "e_conv11/weights", [4, 4, s.out_ch, cin], 'ws'))
    out.append((e + "e_conv11/biases", [s.out_ch], 'bs'))
    return out


def init_shader_weights(spec, seed=1234, perturb=False):
    """Seeded synthetic weights with the reference initialisers (no trained weights ship with the
    reference).  perturb=True additionally draws biases ~N(0,0.01) and PReLU alpha
    ~U(0,0.25) so that parity runs exercise the negative PReLU branch.
    Returns {tf_name: float32 ndarray in TF layout}."""
    rng = np.random.default_rng(seed)
    xav = xavier_initializer()
    w = {}
    for name, shape, kind in shader_variable_shapes(spec):
        if kind in ('w3', 'ws'):
            w[name] = xav(shape, rng)
        elif kind == 'b3':
            w[name] = np.full(shape, 0.001, np.float32)
        elif kind == 'bs':
            w[name] = np.zeros(shape, np.float32)
        else:
            w[name] = np.zeros(shape, np.float32)
        if perturb and kind in ('b3', 'bs'):
            w[name] = (rng.standard_normal(shape) * 0.01).astype(np.float32)
        if perturb and kind == 'a':
            w[name] = rng.uniform(0.0, 0.25, shape).astype(np.float32)
    return w


def _dropout(x, kp):
    """tf.nn.dropout(x, kp) = x/kp * floor(kp + U[0,1)); identity at kp == 1 (inference).  One HIP launch
    (rn_dropout, counter-based mask regenerated in the backward pass)."""
    from . import ops
    return ops.dropout(x, kp)


def RenderNet(models_in, is_training, prob=0.75, reuse=False, spec=None, taps=None):
    """RenderNet_Shader.py:32-131.  models_in [B,H,W,D,in_ch] (resampled, image-aligned voxels);
    returns the sigmoid image [B,4H,4W,out_ch].  PReLU / residual adds / the sigmoid run in the
    conv epilogues.  `taps` (dict) collects named intermediates for the parity tests."""
    s = spec or ShaderSpec()
    st = V.get_default_store()
    kp = keep_prob(prob, is_training)
    xav = xavier_initializer

    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    def alpha_in(scope, ch):
        with st.variable_scope(scope):
            a, _ = st.get_variable('alpha', shape=[ch], initializer=constant_initializer(0.0))
        return a

    with st.variable_scope("encoder"):
        a1 = alpha_in('e_conv1', s.c1)
        with st.variable_scope('e_conv1'):
            enc1 = conv3d(models_in, s.c1, kernel_size=[5, 5, 5], stride=[2, 2, 2], reuse=reuse, pad="SAME",
                          scope='e_conv1', weight_initializer_type=xav(), activation_alpha=a1)      # :36-39
            enc1 = _dropout(enc1, kp)
        tap("enc1", enc1)
        a2 = alpha_in('e_conv2', s.c2)
        with st.variable_scope('e_conv2'):
            enc2 = conv3d(enc1, s.c2, kernel_size=[3, 3, 3], stride=[1, 1, 2], reuse=reuse, pad="SAME",
                          scope='e_conv2', weight_initializer_type=xav(), activation_alpha=a2)      # :40-43
            enc2 = _dropout(enc2, kp)
        tap("enc2", enc2)
        a3 = alpha_in('e_conv3', s.c3)
        with st.variable_scope('e_conv3'):
            enc3 = conv3d(enc2, s.c3, kernel_size=[3, 3, 3], stride=[1, 1, 1], reuse=reuse, pad="SAME",
                          scope='e_conv3', weight_initializer_type=xav(), activation_alpha=a3)      # :44-47
            enc3 = _dropout(enc3, kp)
        tap("enc3", enc3)

        net = enc3
        for k in range(1, s.n_res1 + 1):                                                            # :51-60
            net = res_block_3d(net, s.c3, scope='res1_%d' % k)
        tap("res1", net)
        with st.variable_scope('res1_skip'):                                                        # :62-64
            enc3_skip = conv3d(net, s.c3, kernel_size=[3, 3, 3], stride=[1, 1, 1], pad="SAME", scope="con1_3X3",
                               weight_initializer_type=xav(), residual=enc3)
        tap("enc3_skip", enc3_skip)

        enc4 = tap("enc4", projection_unit(enc3_skip))                                              # :67

        # :71-84 -- ten res_block_2d and the res2_skip conv + shortcut, as one Winograd chain at inference (ops.res_stack_2d)
        enc4_skip = LU.res_stack_2d(enc4, s.w_res2, s.n_res2, 'res2_%d', skip_scope='res2_skip', skip_residual=enc4)
        tap("enc4_skip", enc4_skip)

        a5 = alpha_in('e_conv5', s.w5)
        with st.variable_scope('e_conv5'):                                                          # :86-88
            enc5 = LU.conv2d(enc4_skip, s.w5, kernel_size=[4, 4], stride=[1, 1], scope='e_conv5',
                             weight_initializer_type=xav(), activation_alpha=a5, default_bias=0.0)
            enc5 = _dropout(enc5, kp)
        tap("enc5", enc5)

        enc5_skip = LU.res_stack_2d(enc5, s.w5, s.n_res3, 'res3_%d', skip_scope='res3_skip', skip_residual=enc5)   # :91-99
        tap("enc5_skip", enc5_skip)

        a6 = alpha_in('e_conv6', s.w6)
        with st.variable_scope('e_conv6'):                                                          # :101-103
            enc6 = LU.conv2d(enc5_skip, s.w6, kernel_size=[4, 4], stride=[1, 1], scope='e_conv6',
                             weight_initializer_type=xav(), activation_alpha=a6, default_bias=0.0)
            enc6 = _dropout(enc6, kp)
        tap("enc6", enc6)

        net = enc6
        for name, width, stride in (("e_conv7", s.w7, 2), ("e_conv7_1", s.w7_1, 1), ("e_conv8", s.w8, 2),
                                    ("e_conv9", s.w9, 2), ("e_conv10", s.w10, 1)):                 # :105-123
            an = alpha_in(name, width)
            with st.variable_scope(name):
                net = LU.conv2d_transpose(net, width, kernel_size=[4, 4], stride=[stride, stride], scope=name,
                                          weight_initializer_type=xav(), activation_alpha=an, default_bias=0.0)
                net = _dropout(net, kp)
            tap("enc" + name[6:], net)

        # :125-131 -- e_conv11 sits directly under "encoder"; sigmoid fused into its epilogue
        output = LU.conv2d_transpose(net, s.out_ch, kernel_size=[4, 4], stride=[1, 1], scope='e_conv11',
                                     weight_initializer_type=xav(), sigmoid=True, default_bias=0.0)
        tap("output", output)
        return output


class Renderer:
    """Stands in for the TF1 Session of the reference (RenderNet_demo.py:23-30,113; the graph of
    RenderNet_Shader.py:135-156).  Holds the weights on one GPU and executes
    resample -> (crop) -> RenderNet for a batch."""

    def __init__(self, spec=None, weights=None, device="cuda", seed=1234, gemm=None):
        """gemm: the multiply-stage mode of THIS renderer ("f32" exact fp32 | "split" bf16x3 | "split16" fp16x2; rendernet_amd.ops);
        None = the process default (env RN_WINO_GEMM, "split").  Two renderers of one process may differ."""
        from . import ops
        if gemm is not None and gemm not in ops.GEMM_MODES:
            raise ValueError("gemm=%r: expected one of %s" % (gemm, ", ".join(ops.GEMM_MODES)))
        self.gemm = gemm
        self.spec = (spec or ShaderSpec()).check()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("rendernet_amd.Renderer needs a HIP device; there is no CPU render path")
        self.store = V.VariableStore(self.device, seed)
        if weights is None:
            weights = init_shader_weights(self.spec, seed)
        self.store.load_state_dict(weights)

    def render(self, voxels, poses, is_training=False, patch_size=None, start_point=None, prob=1.0, taps=None):
        """voxels [B,S,S,S,C] and poses [B,3] (radians; azimuth, elevation, 3.3/radius) as numpy or
        tensors.  Returns a HIP tensor [B,4p,4p,out_ch]."""
        s = self.spec
        vox = torch.as_tensor(voxels, dtype=torch.float32).to(self.device)
        pose = torch.as_tensor(np.asarray(poses, np.float32) if not isinstance(poses, torch.Tensor) else poses,
                               dtype=torch.float32).to(self.device)
        window = None
        if is_training and patch_size is not None and patch_size != s.new_size:
            # tf_random_crop_voxel_image (tools/model_util.py:77-100): one start for the batch; the
            # crop is folded into the resampler so only the patch is ever produced.
            if start_point is None:
                start_point = torch.randint(0, s.new_size - patch_size + 1, (2,)).tolist()
            window = (int(start_point[0]), int(start_point[1]), int(patch_size), int(patch_size))
        from . import ops
        old = V._default
        V.set_default_store(self.store)
        try:
            with torch.no_grad(), ops.gemm_mode(self.gemm):   # the Renderer is the inference runner (the trainers own the differentiable graph)
                net_in = rotation_resampling_to_image(vox, pose, size=s.size, new_size=s.new_size, window=window)
                if taps is not None:
                    taps["net_in"] = net_in
                return RenderNet(net_in, is_training, prob=prob, spec=s, taps=taps)
        finally:
            V._default = old

    def validate_winograd(self, voxels, poses, tol=2e-4):
        """One render with the F(6x6,3x3) self-check on (ops.WINO63_CHECK_TOL): every filter that takes that route is also run
        through F(4x4,3x3) on the activations this input produces, and demoted if the two differ by more than tol * max|y|.
        Call it once after loading weights of unknown provenance, on a representative input; returns the demotions."""
        from . import ops
        old, ops.WINO63_CHECK_TOL = ops.WINO63_CHECK_TOL, float(tol)
        n0 = len(ops.WINO63_DEMOTED)
        try:
            self.render(voxels, poses)
        finally:
            ops.WINO63_CHECK_TOL = old
        return ops.WINO63_DEMOTED[n0:]

    # -- hipGraph replay for small batches ----------------------------------------------------
    def capture(self, batch, in_ch=1):
        """Capture one inference render of `batch` frames into a hipGraph (torch.cuda.CUDAGraph over the HIP stream
        the kernels are launched on) and return `replay(voxels, poses) -> image tensor`.  At batch 1-2 the 81 launches
        of a render are host-bound when issued one by one from Python (ctypes + allocator per launch); the graph
        replays them with one call.  Inputs are copied into static buffers; the returned tensor is the graph's static
        output buffer (overwritten by the next replay)."""
        s = self.spec
        vox_buf = torch.zeros((batch, s.size, s.size, s.size, in_ch), dtype=torch.float32, device=self.device)
        pose_buf = torch.zeros((batch, 3), dtype=torch.float32, device=self.device)
        pose_buf[:, 2] = 1.0
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):                         # warm-up: packs the filters, sets kernel attributes
                self.render(vox_buf, pose_buf)
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            out_buf = self.render(vox_buf, pose_buf)

        def replay(voxels, poses):
            vox_buf.copy_(torch.as_tensor(voxels, dtype=torch.float32), non_blocking=True)
            pose_buf.copy_(torch.as_tensor(np.asarray(poses, np.float32) if not isinstance(poses, torch.Tensor) else poses,
                                           dtype=torch.float32), non_blocking=True)
            graph.replay()
            return out_buf

        replay.graph = graph
        return replay

    # -- the reference's Session contract ---------------------------------------------------
    def run(self, fetches, feed_dict):
        """sess.run("encoder/output:0", {"real_model_in:0": vox, "view_name:0": pose,
        "patch_size:0": 128, "is_training:0": False})  (RenderNet_demo.py:47-51).  Returns numpy."""
        single = isinstance(fetches, str)
        names = [fetches] if single else list(fetches)
        for n in names:
            if n != "encoder/output:0":
                raise KeyError("unknown fetch %r (only 'encoder/output:0' is exported)" % n)
        try:
            vox = feed_dict["real_model_in:0"]
            pose = feed_dict["view_name:0"]
        except KeyError as e:
            raise KeyError("feed_dict is missing %s" % e)
        training = bool(feed_dict.get("is_training:0", False))
        patch = feed_dict.get("patch_size:0", None)
        out = self.render(vox, pose, is_training=training, patch_size=patch).cpu().numpy()
        return out if single else [out for _ in names]
