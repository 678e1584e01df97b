"""Texture + normal face renderer (BASELINE config 3) on the MI355X path.

Mirrors RenderNet_Texture_Face_Normal.py: `decoder_texture` (:34-46), the two-head `RenderNet`
(:48-147) and the graph (:152-179): geometry voxels and the decoded 4-channel texture volume are
resampled with the same pose, concatenated to 5 channels and rendered to an albedo image and a
normal map.  Variable names follow the TF scopes of the reference, including its irregular ones
(`Image/e_conv7_1/e_conv7_2`, default scopes `conv2d_transpose` / `conv3d_transpose` / `conv3d` /
`fully_connected` where the reference passes none).
"""
from dataclasses import dataclass

import numpy as np
import torch

from . import variables as V
from .tools import layer_util as LU
from .tools.layer_util import res_block_2d, res_block_3d, projection_unit
from .tools.resampling_voxel_grid import rotation_resampling_concat_to_image, rotation_resampling_to_image  # noqa: F401
from .variables import xavier_initializer, constant_initializer, random_normal_initializer


@dataclass
class TextureSpec:
    """Channel plan; defaults = the reference (RenderNet_Texture_Face_Normal.py:34-147)."""
    size: int = 64
    new_size: int = 128
    z_dim: int = 199          # texture code (:159)
    tex_res: int = 32         # FC output grid 32^3 x 4 (:38-39)
    tex_c0: int = 4
    tex_c1: int = 8
    tex_c2: int = 4
    c1: int = 8
    c2: int = 16
    c3: int = 16
    n_res1: int = 10
    w_res2: int = 32 * 16
    n_res2: int = 10
    w5: int = 32 * 8
    n_res3: int = 5
    w6: int = 32 * 4
    w7: int = 32 * 2
    w8: int = 32
    w9: int = 16

    def check(self):
        if (self.new_size // 4) * self.c3 != self.w_res2:
            raise ValueError("projection width mismatch")
        if self.tex_res * 2 != self.size:
            raise ValueError("texture decoder output %d^3 != voxel size %d^3" % (self.tex_res * 2, self.size))
        return self


def tiny_texture_spec():
    return TextureSpec(size=16, new_size=32, z_dim=19, tex_res=8, c1=8, c2=16, c3=16, n_res1=2, w_res2=128, n_res2=2,
                       w5=64, n_res3=1, w6=32, w7=32, w8=16, w9=16).check()


HEADS = (("Image", "_1"), ("Normal", "_2"))


def _head_scopes(head, sfx):
    """(variable scope, conv scope) per head layer, as spelled in the reference (:113-145)."""
    if head == "Image":
        return [("e_conv6_1", "e_conv6_1"), ("e_conv7_1", "e_conv7_2"), ("e_conv8_1", "conv2d_transpose"),
                ("e_conv9_1", "conv2d_transpose"), ("e_conv10_1", "conv2d_transpose")]
    return [("e_conv6_2", "e_conv6_2"), ("e_conv7_2", "e_conv7_2"), ("e_conv8_2", "e_conv8_2"),
            ("e_conv9_2", "e_conv9_2"), ("e_conv10_2", "e_conv10_2")]


def texture_variable_shapes(spec):
    """[(tf_name, shape, kind)]: kind 'wx' xavier filter, 'wn' N(0,0.02) filter, 'b' bias 0.001,
    'bs' slim bias 0, 'a' alpha 0."""
    s = spec
    out = []
    t = "texture_encoder/"
    F = s.tex_res ** 3 * s.tex_c0
    out += [(t + "e_tex_fc1/fully_connected/weights", [s.z_dim, F], 'wn'), (t + "e_tex_fc1/fully_connected/biases", [F], 'b'),
            (t + "e_tex_fc1/alpha", [F], 'a'),
            (t + "e_tex_conv0/conv3d_transpose/weights", [4, 4, 4, s.tex_c0, s.tex_c0], 'wn'),
            (t + "e_tex_conv0/conv3d_transpose/biases", [s.tex_c0], 'b'), (t + "e_tex_conv0/alpha", [s.tex_c0], 'a'),
            (t + "e_tex_conv1/conv3d_transpose/weights", [4, 4, 4, s.tex_c1, s.tex_c0], 'wn'),
            (t + "e_tex_conv1/conv3d_transpose/biases", [s.tex_c1], 'b'), (t + "e_tex_conv1/alpha", [s.tex_c1], 'a'),
            (t + "e_tex_conv2/conv3d/weights", [4, 4, 4, s.tex_c1, s.tex_c2], 'wn'),
            (t + "e_tex_conv2/conv3d/biases", [s.tex_c2], 'b'), (t + "e_tex_conv2/alpha", [s.tex_c2], 'a')]
    e = "encoder/"
    cin = 1 + s.tex_c2
    for name, k, ci, co in (("e_conv1", 5, cin, s.c1), ("e_conv2", 3, s.c1, s.c2), ("e_conv3", 3, s.c2, s.c3)):
        out += [(e + "%s/%s/weights" % (name, name), [k, k, k, ci, co], 'wx'), (e + "%s/%s/biases" % (name, name), [co], 'b'),
                (e + "%s/alpha" % name, [co], 'a')]
    for i in range(1, s.n_res1 + 1):
        sc = e + "res1_%d/" % i
        out.append((sc + "alpha", [s.c3], 'a'))
        for n in ("con1_3X3", "conv2_3x3"):
            out += [(sc + n + "/weights", [3, 3, 3, s.c3, s.c3], 'wx'), (sc + n + "/biases", [s.c3], 'b')]
    out += [(e + "res1_skip/con1_3X3/weights", [3, 3, 3, s.c3, s.c3], 'wx'), (e + "res1_skip/con1_3X3/biases", [s.c3], 'b')]
    Fp = s.w_res2
    out += [(e + "projection_unit/Conv/weights", [1, 1, Fp, Fp], 'wx'), (e + "projection_unit/Conv/biases", [Fp], 'bs'),
            (e + "projection_unit/alpha", [Fp], 'a')]

    def res2d(prefix, n, width):
        for i in range(1, n + 1):
            sc = e + "%s_%d/" % (prefix, i)
            out.append((sc + "alpha", [width], 'a'))
            for nm in ("con1_3X3", "conv2_3x3"):          # slim convs inside res_block_2d: zero biases
                out.extend([(sc + nm + "/weights", [3, 3, width, width], 'wx'), (sc + nm + "/biases", [width], 'bs')])
        # the skip conv is the hand-rolled conv2d (:94, :109): bias 0.001
        out.extend([(e + "%s_skip/con1_3X3/weights" % prefix, [3, 3, width, width], 'wx'),
                    (e + "%s_skip/con1_3X3/biases" % prefix, [width], 'b')])

    res2d("res2", s.n_res2, Fp)
    out += [(e + "e_conv5/e_conv5/weights", [4, 4, Fp, s.w5], 'wx'), (e + "e_conv5/e_conv5/biases", [s.w5], 'b'),
            (e + "e_conv5/alpha", [s.w5], 'a')]
    res2d("res3", s.n_res3, s.w5)
    for head, sfx in HEADS:
        sc = _head_scopes(head, sfx)
        p = e + head + "/"
        out += [(p + "%s/%s/weights" % sc[0], [4, 4, s.w5, s.w6], 'wx'), (p + "%s/%s/biases" % sc[0], [s.w6], 'b'),
                (p + "%s/alpha" % sc[0][0], [s.w6], 'a')]
        cin = s.w6
        for (vs, cs), co in zip(sc[1:4], (s.w7, s.w8, s.w9)):
            out += [(p + "%s/%s/weights" % (vs, cs), [4, 4, co, cin], 'wx'), (p + "%s/%s/biases" % (vs, cs), [co], 'b'),
                    (p + "%s/alpha" % vs, [co], 'a')]
            cin = co
        vs, cs = sc[4]
        out += [(p + "%s/%s/weights" % (vs, cs), [4, 4, 3, cin], 'wx'), (p + "%s/%s/biases" % (vs, cs), [3], 'b')]
    return out


def init_texture_weights(spec, seed=1234, perturb=False):
    rng = np.random.default_rng(seed)
    xav, nrm = xavier_initializer(), random_normal_initializer(0.02)
    w = {}
    for name, shape, kind in texture_variable_shapes(spec):
        if kind == 'wx':
            w[name] = xav(shape, rng)
        elif kind == 'wn':
            w[name] = nrm(shape, rng)
        elif kind == 'b':
            w[name] = np.full(shape, 0.001, np.float32)
        else:
            w[name] = np.zeros(shape, np.float32)
        if perturb and kind in ('b', 'bs'):
            w[name] = (rng.standard_normal(shape) * 0.01).astype(np.float32)
        if perturb and kind == 'a':
            w[name] = rng.uniform(0.0, 0.25, shape).astype(np.float32)
    return w


def _alpha(st, scope, ch):
    with st.variable_scope(scope):
        a, _ = st.get_variable('alpha', shape=[ch], initializer=constant_initializer(0.0))
    return a


def decoder_texture(z_in, spec=None, taps=None):
    """RenderNet_Texture_Face_Normal.py:34-46: z [B,199] -> FC+PReLU -> [B,32,32,32,4] ->
    conv3d_transpose k4 s1 (4) -> conv3d_transpose k4 s2 (8) -> conv3d k4 s1 (4): [B,64,64,64,4]."""
    s = spec or TextureSpec()
    st = V.get_default_store()
    B = z_in.shape[0]
    with st.variable_scope("texture_encoder"):
        F = s.tex_res ** 3 * s.tex_c0
        a = _alpha(st, 'e_tex_fc1', F)
        with st.variable_scope('e_tex_fc1'):
            zP = LU.fully_connected(z_in, F, activation_alpha=a)
        z_resized = zP.reshape(B, s.tex_res, s.tex_res, s.tex_res, s.tex_c0)
        a = _alpha(st, 'e_tex_conv0', s.tex_c0)
        with st.variable_scope('e_tex_conv0'):
            conv0 = LU.conv3d_transpose(z_resized, s.tex_c0, kernel_size=[4, 4, 4], stride=[1, 1, 1], activation_alpha=a)
        a = _alpha(st, 'e_tex_conv1', s.tex_c1)
        with st.variable_scope('e_tex_conv1'):
            conv1 = LU.conv3d_transpose(conv0, s.tex_c1, kernel_size=[4, 4, 4], stride=[2, 2, 2], activation_alpha=a)
        a = _alpha(st, 'e_tex_conv2', s.tex_c2)
        with st.variable_scope('e_tex_conv2'):
            conv2 = LU.conv3d(conv1, s.tex_c2, kernel_size=[4, 4, 4], stride=[1, 1, 1], activation_alpha=a)
    if taps is not None:
        taps["tex_fc"], taps["tex_conv0"], taps["tex_conv1"], taps["texture_decoded"] = z_resized, conv0, conv1, conv2
    return conv2


def RenderNetTexture(models_in, prob=0.75, reuse=False, spec=None, taps=None, is_training=False):
    """RenderNet_Texture_Face_Normal.py:48-147.  models_in [B,H,W,D,5]; returns (image [B,4H,4W,3], normal
    [B,4H,4W,3]).  `is_training` stands for the reference's graph-level placeholder (:161): with it, tf.nn.dropout
    runs after e_conv1..3, e_conv5 and the first four layers of each head (:55-65, :101, :116-125, :133-142)."""
    from .shader import _dropout
    s = spec or TextureSpec()
    st = V.get_default_store()
    xav = xavier_initializer
    kp = LU.keep_prob(prob, is_training)

    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    with st.variable_scope("encoder"):
        net = models_in
        for name, co, k, stride in (("e_conv1", s.c1, 5, [2, 2, 2]), ("e_conv2", s.c2, 3, [1, 1, 2]), ("e_conv3", s.c3, 3, [1, 1, 1])):
            a = _alpha(st, name, co)
            with st.variable_scope(name):
                net = LU.conv3d(net, co, kernel_size=[k, k, k], stride=stride, reuse=reuse, pad="SAME", scope=name,
                                weight_initializer_type=xav(), activation_alpha=a)
                net = _dropout(net, kp)
            tap("enc" + name[-1], net)
        enc3 = net
        for i in range(1, s.n_res1 + 1):
            net = res_block_3d(net, s.c3, scope='res1_%d' % i)
        with st.variable_scope('res1_skip'):
            enc3_skip = LU.conv3d(net, s.c3, kernel_size=[3, 3, 3], stride=[1, 1, 1], pad="SAME", scope="con1_3X3",
                                  weight_initializer_type=xav(), residual=enc3)
        tap("enc3_skip", enc3_skip)
        enc4 = tap("enc4", projection_unit(enc3_skip))
        enc4_skip = LU.res_stack_2d(enc4, s.w_res2, s.n_res2, 'res2_%d', skip_scope='res2_skip', skip_residual=enc4,
                                    skip_default_bias=0.001)
        tap("enc4_skip", enc4_skip)
        a5 = _alpha(st, 'e_conv5', s.w5)
        with st.variable_scope('e_conv5'):
            enc5 = LU.conv2d(enc4_skip, s.w5, kernel_size=[4, 4], stride=[1, 1], scope='e_conv5',
                             weight_initializer_type=xav(), activation_alpha=a5)
            enc5 = _dropout(enc5, kp)
        tap("enc5", enc5)
        enc5_skip = LU.res_stack_2d(enc5, s.w5, s.n_res3, 'res3_%d', skip_scope='res3_skip', skip_residual=enc5,
                                    skip_default_bias=0.001)
        tap("enc5_skip", enc5_skip)

        outs = []
        for head, sfx in HEADS:
            sc = _head_scopes(head, sfx)
            with st.variable_scope(head):
                a = _alpha(st, sc[0][0], s.w6)
                with st.variable_scope(sc[0][0]):
                    net = LU.conv2d(enc5_skip, s.w6, kernel_size=[4, 4], stride=[1, 1], scope=sc[0][1],
                                    weight_initializer_type=xav(), activation_alpha=a)
                    net = _dropout(net, kp)
                for (vs, cs), co in zip(sc[1:4], (s.w7, s.w8, s.w9)):
                    a = _alpha(st, vs, co)
                    with st.variable_scope(vs):
                        net = LU.conv2d_transpose(net, co, [4, 4], stride=[2, 2], scope=cs, weight_initializer_type=xav(),
                                                  activation_alpha=a)
                        net = _dropout(net, kp)
                vs, cs = sc[4]
                with st.variable_scope(vs):
                    net = LU.conv2d_transpose(net, 3, [4, 4], stride=[1, 1], scope=cs, weight_initializer_type=xav(),
                                              sigmoid=True)
            outs.append(tap(head.lower(), net))
        return outs[0], outs[1]


class TextureRenderer:
    """Graph of RenderNet_Texture_Face_Normal.py:152-179 on one GPU."""

    def __init__(self, spec=None, weights=None, device="cuda", seed=1234, gemm=None):
        """gemm: this renderer's multiply-stage mode (see shader.Renderer); None = the process default."""
        from . import ops
        if gemm is not None and gemm not in ops.GEMM_MODES:
            raise ValueError("gemm=%r: expected one of %s" % (gemm, ", ".join(ops.GEMM_MODES)))
        self.gemm = gemm
        self.spec = (spec or TextureSpec()).check()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("rendernet_amd.TextureRenderer needs a HIP device; there is no CPU render path")
        self.store = V.VariableStore(self.device, seed)
        self.store.load_state_dict(weights if weights is not None else init_texture_weights(self.spec, seed))

    def render(self, voxels, textures, poses, taps=None):
        """voxels [B,S,S,S,1], textures [B,z_dim], poses [B,3] -> (image, normal) HIP tensors [B,4N',4N',3]."""
        s = self.spec
        vox = torch.as_tensor(voxels, dtype=torch.float32).to(self.device)
        tex = torch.as_tensor(textures, dtype=torch.float32).to(self.device)
        pose = torch.as_tensor(np.asarray(poses, np.float32) if not isinstance(poses, torch.Tensor) else poses,
                               dtype=torch.float32).to(self.device)
        from . import ops
        old = V._default
        V.set_default_store(self.store)
        try:
            with ops.gemm_mode(self.gemm):
                tex_vol = decoder_texture(tex, s, taps)                                                 # :169
                # :165-166 + :171-172 + :178 -- both resamplers and the concat in one pass
                net_in = rotation_resampling_concat_to_image(vox, tex_vol, pose, size=s.size, new_size=s.new_size)
                if taps is not None:
                    taps["net_in"] = net_in
                return RenderNetTexture(net_in, spec=s, taps=taps)
        finally:
            V._default = old
