"""Inverse rendering through the frozen renderer (Reconstruct_RenderNet_Face.py) on the MI355X path.

The reference recovers a face's shape code, pose, texture code and light azimuth from one image by gradient descent
THROUGH the pretrained networks (:334-413):

    recon_shape   = decoder_3d_pretrained(initial_vector)              FC -> 4 x (conv3d_transpose s2 + ELU) -> sigmoid   (:31-72)
    recon_texture = texture_decoder_pretrained(initial_texture)        the texture decoder of config 3                  (:74-112)
    rotated_*     = transform(tf_rotation_resampling(*, initial_param))                                                   (:360-364)
    img, normal   = RenderNet_pretrained(concat(rotated_model, rotated_texture))                                          (:366-367)
    shading       = tf_phong_composite(normal, tf_generate_light_pos(initial_light, elevation), ...)                      (:358, :377)
    compos_pred   = img * shading ;  recon_loss[b] = mean((target - compos_pred)^2) over (h, w, c)                        (:378-383)
    four GradientDescentOptimizers, one per latent group, on d sum(recon_loss) / d latent                                 (:397-413)

Here every piece is the HIP operator of the forward path plus its input-gradient kernels; the weights are frozen
(`ops.TrainContext(frozen=True)`: no filter / bias / alpha gradients are computed).  The three pretrained builders
(`decoder_3d_pretrained` :31-72, `texture_decoder_pretrained` :74-112, `RenderNet_pretrained` :113-302) follow the
reference's own lines, read their tensors from the reference's weight dicts by the reference's keys (`load_weights`,
tools/model_util.py:26-39) and are NOT the training graph of `rendernet_amd.texture`: their res blocks take the
`weight_dict` branch of tools/layer_util.py:75-88 / :107-121 -- tf.nn.relu and no `alpha` variable -- the depth collapse is a
hand-rolled 1x1 conv2d under `e_conv4`, and the head scopes are spelled `e_conv{6,7,8,9,11}_{1,2}` (DESIGN.md §2 lists every
difference).
"""
from dataclasses import dataclass
import math

import numpy as np
import torch

from . import _lib as L
from . import ops
from . import variables as V
from .texture import TextureSpec, texture_variable_shapes
from .tools import layer_util as LU
from .tools import Phong_shading as Phong
from .tools.model_util import load_weights  # noqa: F401  (tools/model_util.py:26-39)
from .tools.resampling_voxel_grid import rotation_resampling_concat_to_image, rotation_resampling_to_image  # noqa: F401
from .variables import random_normal_initializer


@dataclass
class ShapeDecoderSpec:
    """decoder_3d_pretrained (:31-72): z [B,200] -> FC -> [B,4,4,4,256] -> conv3d_transpose k4 s2 to 128, 64, 32, 16
    channels (ELU) -> conv3d_transpose k4 s1 to 1 channel (sigmoid): [B,64,64,64,1]."""
    z_dim: int = 200
    base: int = 4
    chans: tuple = (256, 128, 64, 32, 16)

    @property
    def size(self):
        return self.base * 2 ** (len(self.chans) - 1)


def tiny_shape_decoder_spec():
    return ShapeDecoderSpec(z_dim=20, base=2, chans=(32, 16, 8, 8))          # 2 -> 4 -> 8 -> 16


def shape_decoder_variable_shapes(spec):
    """[(tf_name, shape, kind)] in the scopes of :40-70."""
    s = spec
    F = s.base ** 3 * s.chans[0]
    out = [("g_zP/g_gc1/weights", [s.z_dim, F], 'wn'), ("g_zP/g_gc1/biases", [F], 'b')]
    for i in range(1, len(s.chans)):
        sc = "g_conv%d/g_conv%d/" % (i, i)
        out += [(sc + "weights", [4, 4, 4, s.chans[i], s.chans[i - 1]], 'wn'), (sc + "biases", [s.chans[i]], 'b')]
    n = len(s.chans)
    out += [("g_conv%d/weights" % n, [4, 4, 4, 1, s.chans[-1]], 'wn'), ("g_conv%d/biases" % n, [1], 'b')]
    return out


def init_shape_decoder_weights(spec, seed=4321, perturb=False):
    rng = np.random.default_rng(seed)
    nrm = random_normal_initializer(0.02)
    w = {}
    for name, shape, kind in shape_decoder_variable_shapes(spec):
        if kind == 'wn':
            w[name] = nrm(shape, rng)
        else:
            w[name] = (rng.standard_normal(shape) * 0.01).astype(np.float32) if perturb else np.full(shape, 0.001, np.float32)
    return w


def decoder_3d_pretrained(z_in, weight_dict, trainable=False, taps=None):
    """Reconstruct_RenderNet_Face.py:31-72 (ELU / sigmoid run in the transposed convs' epilogues).  Widths come from the
    tensors of `weight_dict` (keys g_zP_g_gc1_*, g_conv<i>_g_conv<i>_*, g_conv<n>_*), like TF takes a variable's shape from
    its initialiser (tools/layer_util.py:287-290, :333)."""
    wd = weight_dict
    st = V.get_default_store()
    B = z_in.shape[0]
    n = 1
    while "g_conv%d_g_conv%d_weights" % (n, n) in wd:
        n += 1                                                          # the last layer, g_conv<n>, has no doubled scope (:66-70)
    c0 = int(np.shape(wd["g_conv1_g_conv1_weights"])[4])
    F = int(np.shape(wd["g_zP_g_gc1_weights"])[1])
    base = int(round((F // c0) ** (1.0 / 3.0)))
    with st.variable_scope('g_zP'):
        zP = LU.fully_connected(z_in, F, scope='g_gc1', trainable=trainable, weight_initializer=wd["g_zP_g_gc1_weights"],
                                bias_initializer=wd["g_zP_g_gc1_biases"])
    net = zP.reshape(B, base, base, base, c0)
    for i in range(1, n):
        sc = 'g_conv%d' % i
        w = wd["%s_%s_weights" % (sc, sc)]
        with st.variable_scope(sc):
            net = LU.conv3d_transpose(net, int(np.shape(w)[3]), kernel_size=[4, 4, 4], stride=[2, 2, 2], pad="SAME", scope=sc,
                                      trainable=trainable, weight_initializer=w, bias_initializer=wd["%s_%s_biases" % (sc, sc)],
                                      elu=True)
        if taps is not None:
            taps["gen%d" % i] = net
    sc = 'g_conv%d' % n
    return LU.conv3d_transpose(net, 1, kernel_size=[4, 4, 4], stride=[1, 1, 1], pad="SAME", scope=sc, trainable=trainable,
                               weight_initializer=wd[sc + "_weights"], bias_initializer=wd[sc + "_biases"], sigmoid=True)


def _prelu_var(st, scope, alpha):
    """prelu(x, alpha=weight_dict[...]) (tools/layer_util.py:41-43): the variable `alpha` of the enclosing scope, loaded."""
    with st.variable_scope(scope):
        a, _ = st.get_variable('alpha', shape=list(np.shape(alpha)), initializer=np.asarray(alpha, np.float32))
    return a


def texture_decoder_pretrained(z_in, weight_dict, trainable=False, taps=None):
    """Reconstruct_RenderNet_Face.py:74-112: z [B,199] -> FC + PReLU -> [B,32,32,32,4] -> conv3d_transpose k4 s1 -> k4 s2 ->
    conv3d k4 s1, every layer PReLU with a loaded alpha.  The FC is declared with 4*4*4*512 outputs (:86) but its matrix comes
    from the dict (tools/layer_util.py:333): the loaded [199, 32^3*4] tensor decides, as in TF."""
    wd = weight_dict
    st = V.get_default_store()
    B = z_in.shape[0]
    with st.variable_scope("texture_encoder"):
        F = int(np.shape(wd["e_tex_dc1_g_gc1_weights"])[1])
        c0 = int(np.shape(wd["e_tex_conv0_conv2d_transpose_weights"])[4])
        res = int(round((F // c0) ** (1.0 / 3.0)))
        a = _prelu_var(st, 'e_tex_dc1', wd["e_tex_dc1_alpha"])
        with st.variable_scope('e_tex_dc1'):
            zP = LU.fully_connected(z_in, F, scope='g_gc1', trainable=trainable, weight_initializer=wd["e_tex_dc1_g_gc1_weights"],
                                    bias_initializer=wd["e_tex_dc1_g_gc1_biases"], activation_alpha=a)
        z_resize = zP.reshape(B, res, res, res, c0)
        a = _prelu_var(st, 'e_tex_conv0', wd["e_tex_conv0_alpha"])
        with st.variable_scope('e_tex_conv0'):
            w = wd["e_tex_conv0_conv2d_transpose_weights"]
            conv0 = LU.conv3d_transpose(z_resize, int(np.shape(w)[3]), kernel_size=[4, 4, 4], stride=[1, 1, 1], trainable=trainable,
                                        weight_initializer=w, bias_initializer=wd["e_tex_conv0_conv2d_transpose_biases"],
                                        activation_alpha=a)
        a = _prelu_var(st, 'e_tex_conv1', wd["e_tex_conv1_alpha"])
        with st.variable_scope('e_tex_conv1'):
            w = wd["e_tex_conv1_conv2d_transpose_weights"]
            conv1 = LU.conv3d_transpose(conv0, int(np.shape(w)[3]), kernel_size=[4, 4, 4], stride=[2, 2, 2], trainable=trainable,
                                        weight_initializer=w, bias_initializer=wd["e_tex_conv1_conv2d_transpose_biases"],
                                        activation_alpha=a)
        a = _prelu_var(st, 'e_tex_conv2', wd["e_tex_conv2_alpha"])
        with st.variable_scope('e_tex_conv2'):
            w = wd["e_tex_conv2_conv3d_weights"]
            conv2 = LU.conv3d(conv1, int(np.shape(w)[4]), kernel_size=[4, 4, 4], stride=[1, 1, 1], trainable=trainable,
                              weight_initializer=w, bias_initializer=wd["e_tex_conv2_conv3d_biases"], activation_alpha=a)
    if taps is not None:
        taps["tex_fc"], taps["tex_conv0"], taps["tex_conv1"], taps["texture_decoded"] = z_resize, conv0, conv1, conv2
    return conv2


def RenderNet_pretrained(models_in, weight_dict, prob=1.0, trainable=False, taps=None):
    """Reconstruct_RenderNet_Face.py:113-302.  models_in [B,H,W,D,5] (geometry + 4 texture channels, image-aligned) ->
    (albedo [B,4H,4W,3], normal map [B,4H,4W,3]).  Line by line the reference's pretrained graph, NOT the training graph of
    RenderNet_Texture_Face_Normal.py:48-147:
      * res_block_3d / res_block_2d are called WITH the weight dict (:150-159, :183-192, :213-217): tf.nn.relu between the
        two convs and no `alpha` (tools/layer_util.py:75-88, :107-121); their 2-D convs are the hand-rolled conv2d;
      * the depth collapse (:170-180) is reshape + a hand-rolled 1x1 conv2d + PReLU under scope e_conv4 (keys
        e_conv4_e_conv4_*, e_conv4_alpha), not the slim projection_unit;
      * head layers: e_conv6_h (conv 4x4), e_conv7_h / e_conv8_h / e_conv9_h (conv_transpose 4x4 s2), e_conv11_h
        (conv_transpose 4x4 s1 + sigmoid), h = 1 under "Image", 2 under "Normal" (:226-301); there is no e_conv10;
      * tf.nn.dropout(x, prob) at the reference's sites (:131, :138, :145, :179, :208, :233, :240, :247, :271, :278, :285, :292 -- behind
        e_conv1..5, e_conv6_h..e_conv8_h and, in the Normal head only, e_conv9_2): the identity at the prob = 1.0 the script passes (:367),
        ops.dropout (x / prob * floor(prob + u), Philox) otherwise.
    `trainable` is accepted like the reference's argument; whether parameter gradients are produced is decided by the ops.TrainContext
    the call runs under (frozen=True: input gradients only, as the reference's optimisers only list the latents, :397-413).  The res
    blocks' ReLU has no parameter: under a non-frozen context no gradient is asked for it.
    Widths are those of the loaded tensors."""
    prob = float(prob)
    if not 0.0 < prob <= 1.0:
        raise L.RenderNetHipError("RenderNet_pretrained: prob = %g is not a keep probability in (0, 1]" % prob)
    drop = (lambda t: t) if prob >= 1.0 else (lambda t: ops.dropout(t, prob))
    wd = weight_dict
    st = V.get_default_store()
    B = models_in.shape[0]

    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    def nout(key, axis):
        return int(np.shape(wd[key])[axis])

    with st.variable_scope("encoder"):
        net = models_in
        for name, k, stride in (("e_conv1", 5, [2, 2, 2]), ("e_conv2", 3, [1, 1, 2]), ("e_conv3", 3, [1, 1, 1])):     # :128-147
            key = "%s_%s" % (name, name)
            a = _prelu_var(st, name, wd[name + "_alpha"])
            with st.variable_scope(name):
                net = LU.conv3d(net, nout(key + "_weights", 4), kernel_size=[k, k, k], stride=stride, pad="SAME", scope=name,
                                trainable=trainable, weight_initializer=wd[key + "_weights"], bias_initializer=wd[key + "_biases"],
                                activation_alpha=a)
            net = drop(net)                                                                                           # :131, :138, :145
            tap("enc" + name[-1], net)
        shortcut = net
        c3 = int(net.shape[-1])
        k = 1
        while "res1_%d_con1_3X3_weights" % k in wd:                                                                  # :150-159
            net = LU.res_block_3d(net, c3, scope='res1_%d' % k, weight_dict=wd, trainable=trainable)
            k += 1
        with st.variable_scope('res1_skip'):                                                                          # :161-167
            enc3_skip = LU.conv3d(net, c3, kernel_size=[3, 3, 3], stride=[1, 1, 1], pad="SAME", scope="con1_3X3", trainable=trainable,
                                  weight_initializer=wd["res1_skip_con1_3X3_weights"], bias_initializer=wd["res1_skip_con1_3X3_biases"],
                                  residual=shortcut)
        tap("enc3_skip", enc3_skip)
        H, W = enc3_skip.shape[1], enc3_skip.shape[2]
        enc3_2d = enc3_skip.reshape(B, H, W, enc3_skip.shape[3] * enc3_skip.shape[4])                                  # :172 (f = d*C + c)
        a = _prelu_var(st, 'e_conv4', wd["e_conv4_alpha"])
        with st.variable_scope('e_conv4'):                                                                            # :174-181
            enc4 = LU.conv2d(enc3_2d, nout("e_conv4_e_conv4_weights", 3), kernel_size=[1, 1], scope='e_conv4', trainable=trainable,
                             weight_initializer=wd["e_conv4_e_conv4_weights"], bias_initializer=wd["e_conv4_e_conv4_biases"],
                             activation_alpha=a)
        enc4 = drop(enc4)                                                                                             # :179
        tap("enc4", enc4)

        def res_stack(x, prefix):
            """res<p>_1 .. res<p>_n with the weight dict + the res<p>_skip conv and its shortcut (:183-200, :213-225)."""
            n = 0
            while "%s_%d_con1_3X3_weights" % (prefix, n + 1) in wd:
                n += 1
            return LU.res_stack_2d(x, int(x.shape[-1]), n, prefix + '_%d', skip_scope=prefix + '_skip', skip_residual=x,
                                   skip_default_bias=0.001, weight_dict=wd)

        enc4_skip = tap("enc4_skip", res_stack(enc4, "res2"))
        a = _prelu_var(st, 'e_conv5', wd["e_conv5_alpha"])
        with st.variable_scope('e_conv5'):                                                                            # :202-209
            enc5 = LU.conv2d(enc4_skip, nout("e_conv5_e_conv5_weights", 3), kernel_size=[4, 4], scope='e_conv5', trainable=trainable,
                             weight_initializer=wd["e_conv5_e_conv5_weights"], bias_initializer=wd["e_conv5_e_conv5_biases"],
                             activation_alpha=a)
        enc5 = drop(enc5)                                                                                             # :208
        tap("enc5", enc5)
        enc5_skip = tap("enc5_skip", res_stack(enc5, "res3"))

        outs = []
        for head, h in (("Image", "1"), ("Normal", "2")):                                                              # :226-301
            with st.variable_scope(head):
                name = "e_conv6_" + h
                key = "%s_%s_%s" % (head, name, name)
                a = _prelu_var(st, name, wd["%s_%s_alpha" % (head, name)])
                with st.variable_scope(name):
                    net = LU.conv2d(enc5_skip, nout(key + "_weights", 3), kernel_size=[4, 4], scope=name, trainable=trainable,
                                    weight_initializer=wd[key + "_weights"], bias_initializer=wd[key + "_biases"], activation_alpha=a)
                net = drop(net)                                                                                       # :233, :271
                for num in (7, 8, 9):
                    name = "e_conv%d_%s" % (num, h)
                    key = "%s_%s_%s" % (head, name, name)
                    a = _prelu_var(st, name, wd["%s_%s_alpha" % (head, name)])
                    with st.variable_scope(name):
                        net = LU.conv2d_transpose(net, nout(key + "_weights", 2), [4, 4], stride=[2, 2], scope=name, trainable=trainable,
                                                  weight_initializer=wd[key + "_weights"], bias_initializer=wd[key + "_biases"],
                                                  activation_alpha=a)
                    if num < 9 or head == "Normal":
                        net = drop(net)                                                           # :240, :247, :278, :285, :292 (e_conv9_1: none)
                name = "e_conv11_" + h
                key = "%s_%s_%s" % (head, name, name)
                # the Normal head opens variable scope 'e_conv11' around conv scope 'e_conv11_2' (:295-296); the Image head
                # 'e_conv11_1' around 'e_conv11_1' (:257-258); the dict keys are spelled alike for both
                with st.variable_scope(name if head == "Image" else 'e_conv11'):
                    net = LU.conv2d_transpose(net, nout(key + "_weights", 2), [4, 4], stride=[1, 1], scope=name, trainable=trainable,
                                              weight_initializer=wd[key + "_weights"], bias_initializer=wd[key + "_biases"],
                                              sigmoid=True)
            outs.append(tap(head.lower(), net))
    return outs[0], outs[1]


# ---------------------------------------------------------------------------------------------
# pretrained weight folders (tools/model_util.py:26-39): one `<key>.txt.npz` per tensor, arr_0
# ---------------------------------------------------------------------------------------------
def pretrained_rendernet_shapes(tex_spec=None):
    """[(key, shape, kind)] of the tensors texture_decoder_pretrained + RenderNet_pretrained read from the RenderNet weight
    folder, by the reference's keys (Reconstruct_RenderNet_Face.py:86-110, :128-299; res blocks: scope + '_con1_3X3_weights'
    ..., tools/layer_util.py:78-85, :111-118).  kind: 'wx' / 'wn' filter, 'b' bias, 'a' PReLU slope.  There is NO res*_alpha."""
    s = tex_spec or TextureSpec()
    out = []
    F = s.tex_res ** 3 * s.tex_c0
    out += [("e_tex_dc1_g_gc1_weights", [s.z_dim, F], 'wn'), ("e_tex_dc1_g_gc1_biases", [F], 'b'), ("e_tex_dc1_alpha", [F], 'a'),
            ("e_tex_conv0_conv2d_transpose_weights", [4, 4, 4, s.tex_c0, s.tex_c0], 'wn'), ("e_tex_conv0_conv2d_transpose_biases", [s.tex_c0], 'b'),
            ("e_tex_conv0_alpha", [s.tex_c0], 'a'),
            ("e_tex_conv1_conv2d_transpose_weights", [4, 4, 4, s.tex_c1, s.tex_c0], 'wn'), ("e_tex_conv1_conv2d_transpose_biases", [s.tex_c1], 'b'),
            ("e_tex_conv1_alpha", [s.tex_c1], 'a'),
            ("e_tex_conv2_conv3d_weights", [4, 4, 4, s.tex_c1, s.tex_c2], 'wn'), ("e_tex_conv2_conv3d_biases", [s.tex_c2], 'b'),
            ("e_tex_conv2_alpha", [s.tex_c2], 'a')]
    cin = 1 + s.tex_c2
    for name, k, ci, co in (("e_conv1", 5, cin, s.c1), ("e_conv2", 3, s.c1, s.c2), ("e_conv3", 3, s.c2, s.c3)):
        out += [("%s_%s_weights" % (name, name), [k, k, k, ci, co], 'wx'), ("%s_%s_biases" % (name, name), [co], 'b'), (name + "_alpha", [co], 'a')]
    for i in range(1, s.n_res1 + 1):
        for n in ("con1_3X3", "conv2_3x3"):
            out += [("res1_%d_%s_weights" % (i, n), [3, 3, 3, s.c3, s.c3], 'wx'), ("res1_%d_%s_biases" % (i, n), [s.c3], 'b')]
    out += [("res1_skip_con1_3X3_weights", [3, 3, 3, s.c3, s.c3], 'wx'), ("res1_skip_con1_3X3_biases", [s.c3], 'b')]
    Fp = s.w_res2
    out += [("e_conv4_e_conv4_weights", [1, 1, Fp, Fp], 'wx'), ("e_conv4_e_conv4_biases", [Fp], 'b'), ("e_conv4_alpha", [Fp], 'a')]
    for prefix, n, width in (("res2", s.n_res2, Fp), ("res3", s.n_res3, s.w5)):
        if prefix == "res3":
            out += [("e_conv5_e_conv5_weights", [4, 4, Fp, s.w5], 'wx'), ("e_conv5_e_conv5_biases", [s.w5], 'b'), ("e_conv5_alpha", [s.w5], 'a')]
        for i in range(1, n + 1):
            for nm in ("con1_3X3", "conv2_3x3"):
                out += [("%s_%d_%s_weights" % (prefix, i, nm), [3, 3, width, width], 'wx'), ("%s_%d_%s_biases" % (prefix, i, nm), [width], 'b')]
        out += [("%s_skip_con1_3X3_weights" % prefix, [3, 3, width, width], 'wx'), ("%s_skip_con1_3X3_biases" % prefix, [width], 'b')]
    for head, h in (("Image", "1"), ("Normal", "2")):
        k6 = "%s_e_conv6_%s" % (head, h)
        out += [("%s_e_conv6_%s_weights" % (k6, h), [4, 4, s.w5, s.w6], 'wx'), ("%s_e_conv6_%s_biases" % (k6, h), [s.w6], 'b'), (k6 + "_alpha", [s.w6], 'a')]
        cin = s.w6
        for num, co in ((7, s.w7), (8, s.w8), (9, s.w9)):
            kk = "%s_e_conv%d_%s" % (head, num, h)
            out += [("%s_e_conv%d_%s_weights" % (kk, num, h), [4, 4, co, cin], 'wx'), ("%s_e_conv%d_%s_biases" % (kk, num, h), [co], 'b'),
                    (kk + "_alpha", [co], 'a')]
            cin = co
        kk = "%s_e_conv11_%s" % (head, h)
        out += [("%s_e_conv11_%s_weights" % (kk, h), [4, 4, 3, cin], 'wx'), ("%s_e_conv11_%s_biases" % (kk, h), [3], 'b')]
    return out


def pretrained_decoder_shapes(dec_spec=None):
    """[(key, shape, kind)] of the shape decoder's weight folder (Reconstruct_RenderNet_Face.py:40-70)."""
    return [(name.replace('/', '_'), shape, kind) for name, shape, kind in shape_decoder_variable_shapes(dec_spec or ShapeDecoderSpec())]


def init_pretrained_weight_dicts(tex_spec=None, dec_spec=None, seed=1234, perturb=False):
    """Random stand-ins for the two weight folders (no pretrained weights ship with the reference): (weight_dict_rendernet,
    weight_dict_decoder) keyed like `load_weights` returns them.  perturb: random biases and PReLU slopes in (0, 0.25)."""
    rng = np.random.default_rng(seed)
    from .variables import xavier_initializer
    xav, nrm = xavier_initializer(), random_normal_initializer(0.02)

    def fill(items):
        w = {}
        for key, shape, kind in items:
            if kind == 'wx':
                w[key] = xav(shape, rng)
            elif kind == 'wn':
                w[key] = nrm(shape, rng)
            elif kind == 'b':
                w[key] = (rng.standard_normal(shape) * 0.01).astype(np.float32) if perturb else np.full(shape, 0.001, np.float32)
            else:
                w[key] = rng.uniform(0.0, 0.25, shape).astype(np.float32) if perturb else np.zeros(shape, np.float32)
        return w
    return fill(pretrained_rendernet_shapes(tex_spec)), fill(pretrained_decoder_shapes(dec_spec))


def pretrained_key_map(tex_spec=None, dec_spec=None):
    """{key in the reference's weight dicts: variable name of the TRAINING graphs of this package} (rendernet_amd.texture,
    shape_decoder_variable_shapes) -- for carrying a pretrained folder into RenderNet_Texture_Face_Normal.py's graph (continue
    training / render with TextureRenderer).  The res blocks' `alpha` variables of that graph have NO key: the pretrained graph
    has a ReLU there (tools/layer_util.py:75-88, :107-121); `state_from_pretrained` sets them to zero, which is the same
    function (max(0,x) + 0*min(0,x))."""
    m = {}
    for name, _, _ in shape_decoder_variable_shapes(dec_spec or ShapeDecoderSpec()):
        m[name.replace('/', '_')] = name
    head_alias = {}
    for head, sfx in (("Image", "1"), ("Normal", "2")):
        from .texture import _head_scopes
        for (vs, cs), num in zip(_head_scopes(head, "_" + sfx), (6, 7, 8, 9, 11)):
            head_alias["%s/%s/%s" % (head, vs, cs)] = "%s_e_conv%d_%s_e_conv%d_%s" % (head, num, sfx, num, sfx)
            head_alias["%s/%s" % (head, vs)] = "%s_e_conv%d_%s" % (head, num, sfx)
    for name, _, _ in texture_variable_shapes(tex_spec or TextureSpec()):
        scope, leaf = name.rsplit('/', 1)
        outer, inner = scope.split('/', 1)
        if leaf == "alpha" and inner.split('_')[0] in ("res1", "res2", "res3"):
            continue                                       # res-block slopes: not in the pretrained folders
        if outer == "texture_encoder":
            inner = inner.replace("e_tex_fc1/fully_connected", "e_tex_dc1_g_gc1").replace("e_tex_fc1", "e_tex_dc1")
            inner = inner.replace("conv3d_transpose", "conv2d_transpose")
        else:
            inner = inner.replace("projection_unit/Conv", "e_conv4_e_conv4").replace("projection_unit", "e_conv4")
            inner = head_alias.get(inner, inner)
        m[inner.replace('/', '_') + "_" + leaf] = name
    return m


def state_from_pretrained(weight_dict_rendernet, weight_dict_decoder, tex_spec=None, dec_spec=None):
    """Weight dicts of `load_weights` -> {training-graph variable name: ndarray} (TextureRenderer / TextureTrainer /
    decoder state).  The res blocks' PReLU slopes, which the pretrained folders do not have, become zeros (= the ReLU of the
    pretrained graph).  Raises on a missing key."""
    ts = tex_spec or TextureSpec()
    km = pretrained_key_map(ts, dec_spec)
    src = dict(weight_dict_rendernet)
    src.update(weight_dict_decoder)
    missing = [k for k in km if k not in src]
    if missing:
        raise KeyError("pretrained weights are missing %d tensors, e.g. %s" % (len(missing), missing[:4]))
    state = {name: np.asarray(src[key], np.float32) for key, name in km.items()}
    for name, shape, kind in texture_variable_shapes(ts):
        if name not in state:
            assert kind == 'a' and name.split('/')[1].split('_')[0] in ("res1", "res2", "res3"), name
            state[name] = np.zeros(shape, np.float32)
    return state


def check_pretrained_weight_dicts(weight_dict_rendernet, weight_dict_decoder, tex_spec=None, dec_spec=None):
    """Every tensor the three pretrained builders will read is present with the expected shape; raises KeyError / ValueError
    with the first few offenders (a KeyError deep inside graph construction is what the reference gives)."""
    missing, bad = [], []
    for wd, items in ((weight_dict_rendernet, pretrained_rendernet_shapes(tex_spec)), (weight_dict_decoder, pretrained_decoder_shapes(dec_spec))):
        for key, shape, _ in items:
            if key not in wd:
                missing.append(key)
            elif list(np.shape(wd[key])) != list(shape):
                bad.append((key, tuple(np.shape(wd[key])), tuple(shape)))
    if missing:
        raise KeyError("pretrained weights are missing %d tensors, e.g. %s" % (len(missing), missing[:4]))
    if bad:
        raise ValueError("pretrained tensors with unexpected shapes, e.g. %s" % (bad[:4],))


def create_param_center(batch_size=5, phi_mid=90, phi_range=240, theta_mid=90, theta_range=120):
    """Reconstruct_RenderNet_Face.py:301-318: the five pose hypotheses (corners + centre) of one search window."""
    if batch_size < 5:
        raise ValueError("create_param_center fills five hypotheses; batch_size=%d" % batch_size)
    rad = math.pi / 180.0
    phi_min = ((phi_mid - phi_range * 0.5) % 360) * rad
    phi_max = ((phi_mid + phi_range * 0.5) % 360) * rad
    theta_min = (90 - (theta_mid - theta_range * 0.5)) * rad
    theta_max = (90 - (theta_mid + theta_range * 0.5)) * rad
    params = np.zeros((batch_size, 3), np.float32)
    params[0] = (phi_min, theta_min, 1.0)
    params[1] = (phi_min, theta_max, 1.0)
    params[2] = (phi_mid * rad, (90 - theta_mid) * rad, 1.0)
    params[3] = (phi_max, theta_min, 1.0)
    params[4] = (phi_max, theta_max, 1.0)
    return params


class Reconstructor:
    """The graph and optimisers of Reconstruct_RenderNet_Face.py:334-413 on one GPU.

    Latents (`initial_vector` [B,z], `initial_param` [B,3], `initial_texture` [B,199], `initial_light` [B,1]) are device
    tensors set with `assign`; `step(target)` = one `sess.run([train_op, recon_loss])` (:491): forward, per-hypothesis
    loss, backward to the four latents, four SGD updates.  Returns the losses of the forward it ran (before the update)."""

    def __init__(self, tex_spec=None, dec_spec=None, weight_dict_rendernet=None, weight_dict_decoder=None, batch_size=5,
                 device="cuda", seed=1234, light_elevation_deg=105.0, light_col=(1.0, 1.0, 1.0), ambient=0.0, k_diffuse=1.0,
                 shape_eta=0.8, pose_eta=0.01, tex_eta=0.8, light_eta=0.4, gemm=None):
        """gemm: this graph's multiply-stage mode (rendernet_amd.ops.gemm_mode); None = the process default.
        weight_dict_rendernet / weight_dict_decoder: what `load_weights(weight_dir)` / `load_weights(weight_dir_decoder)`
        return (:337-339; the texture decoder reads the RenderNet folder, :339); None: seeded random stand-ins."""
        if gemm is not None and gemm not in ops.GEMM_MODES:
            raise ValueError("gemm=%r: expected one of %s" % (gemm, ", ".join(ops.GEMM_MODES)))
        self.gemm = gemm
        self.tex_spec = (tex_spec or TextureSpec()).check()
        self.dec_spec = dec_spec or ShapeDecoderSpec()
        if self.dec_spec.size != self.tex_spec.size:
            raise ValueError("shape decoder emits %d^3, the renderer expects %d^3" % (self.dec_spec.size, self.tex_spec.size))
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("rendernet_amd.Reconstructor needs a HIP device; there is no CPU path")
        self.B = int(batch_size)
        self.store = V.VariableStore(self.device, seed)
        if weight_dict_rendernet is None or weight_dict_decoder is None:
            wr, wdec = init_pretrained_weight_dicts(self.tex_spec, self.dec_spec, seed)
            weight_dict_rendernet = wr if weight_dict_rendernet is None else weight_dict_rendernet
            weight_dict_decoder = wdec if weight_dict_decoder is None else weight_dict_decoder
        check_pretrained_weight_dicts(weight_dict_rendernet, weight_dict_decoder, self.tex_spec, self.dec_spec)
        self.weight_dict_MLP = self.weight_dict_texture = weight_dict_rendernet          # :337, :339
        self.weight_dict_decoder = weight_dict_decoder                                   # :338
        self.ctx = ops.TrainContext(frozen=True, device=self.device)
        self.elevation = (90.0 - float(light_elevation_deg)) * math.pi / 180.0          # :330
        self.light_col = torch.tensor([list(light_col)], dtype=torch.float32, device=self.device).expand(self.B, 3).contiguous()
        self.ambient, self.k_diffuse = float(ambient), float(k_diffuse)
        self.etas = {"vector": float(shape_eta), "param": float(pose_eta), "texture": float(tex_eta), "light": float(light_eta)}
        z = lambda n: torch.zeros((self.B, n), dtype=torch.float32, device=self.device, requires_grad=True)
        self.latents = {"vector": z(self.dec_spec.z_dim), "param": z(3), "texture": z(self.tex_spec.z_dim), "light": z(1)}
        self.loss = torch.zeros(self.B, dtype=torch.float64, device=self.device)
        self.global_step = 0

    # -- the assign ops (:351-354) ------------------------------------------------------------
    def assign(self, vector=None, param=None, texture=None, light=None):
        for name, val in (("vector", vector), ("param", param), ("texture", texture), ("light", light)):
            if val is not None:
                t = torch.as_tensor(np.asarray(val, np.float32) if not isinstance(val, torch.Tensor) else val,
                                    dtype=torch.float32).to(self.device).reshape(self.latents[name].shape)
                with torch.no_grad():
                    self.latents[name].copy_(t)

    def values(self):
        return {k: v.detach().cpu().numpy() for k, v in self.latents.items()}

    # -- graph ----------------------------------------------------------------------------------
    def forward(self, taps=None):
        """Returns (compos_pred, img_pred, normal_pred, recon_shape), all differentiable w.r.t. the latents."""
        ts, lat = self.tex_spec, self.latents
        old = V._default
        V.set_default_store(self.store)
        try:
            with ops.gemm_mode(self.gemm), ops.training(self.ctx):
                shape = decoder_3d_pretrained(lat["vector"], self.weight_dict_decoder, taps=taps)                  # :356
                tex = texture_decoder_pretrained(lat["texture"], self.weight_dict_texture, taps=taps)              # :357
                # :360-361 + :363-364 + :366 -- both resamplers and the concat in one pass
                net_in = rotation_resampling_concat_to_image(shape, tex, lat["param"], size=ts.size, new_size=ts.new_size)
                img, nrm = RenderNet_pretrained(net_in, self.weight_dict_MLP, prob=1.0, taps=taps)                 # :367
                light_dir = Phong.tf_generate_light_pos(lat["light"], self.elevation, self.B)                      # :358
                compos = Phong.tf_phong_composite(nrm, light_dir, self.light_col, self.ambient, self.k_diffuse,
                                                  with_mask=True, albedo=img)                                       # :377-378
        finally:
            V._default = old
        if taps is not None:
            taps.update(recon_shape=shape, recon_texture=tex, net_in=net_in, light_dir=light_dir)
        return compos, img, nrm, shape

    def recon_loss(self, compos, target, dpred=None):
        """recon_loss [B] = mean over (h, w, c) of (target - compos_pred)^2 (:383) into self.loss (float64, device);
        `dpred`, when given, receives d sum(recon_loss) / d compos_pred."""
        tgt = torch.as_tensor(target, dtype=torch.float32).to(self.device).contiguous()
        if tgt.shape != compos.shape:
            raise ValueError("target %s vs prediction %s" % (tuple(tgt.shape), tuple(compos.shape)))
        pred = compos.detach().contiguous()
        per = pred[0].numel()
        self.loss.zero_()
        lib, st = L.lib(), L.stream_ptr()
        for b in range(self.B):                        # one mean per hypothesis: B small launches of the loss kernel
            L.check(lib.rn_loss_fwd_bwd(L.ptr(pred[b]), L.ptr(tgt[b]), L.ptr(dpred[b]) if dpred is not None else None,
                                        self.loss[b:].data_ptr(), per, float(per), 1, st), "rn_loss_fwd_bwd")
        return self.loss

    def loss_and_backward(self, compos, target):
        """Losses into self.loss; d sum(recon_loss) / d latents into the latents' .grad (tf.gradients, :404)."""
        for t in self.latents.values():
            t.grad = None
        dpred = torch.empty_like(compos)
        self.recon_loss(compos, target, dpred)
        compos.backward(dpred)
        return self.loss

    def apply_gradients(self):
        """The four GradientDescentOptimizers (:397-413)."""
        lib, st = L.lib(), L.stream_ptr()
        for name, t in self.latents.items():
            if t.grad is None:
                continue
            g = t.grad.contiguous()
            L.check(lib.rn_sgd_step(L.ptr(t.detach()), L.ptr(g), t.numel(), self.etas[name], st), "rn_sgd_step")
        self.global_step += 1

    def step(self, target):
        compos, _, _, _ = self.forward()
        loss = self.loss_and_backward(compos, target).clone()
        self.apply_gradients()
        return loss


def shaded_target(target_albedo, target_normal, light_azimuth_deg, light_elevation_deg, light_col=(1.0, 1.0, 1.0),
                  ambient=0.0, k_diffuse=1.0):
    """Reconstruct_RenderNet_Face.py:430-444: albedo * np_phong_composite(normal, white background) with the light at the
    ground-truth azimuth / elevation.  Inputs [1,H,W,3] in [0,1] (ndarray); returns ndarray."""
    el = (90.0 - light_elevation_deg) * math.pi / 180.0
    az = light_azimuth_deg * math.pi / 180.0
    light_dir = np.array([[math.sin(el) * math.cos(az), math.sin(el) * math.sin(az), math.cos(el)]], np.float32)
    shading = Phong.np_phong_composite(np.asarray(target_normal, np.float32), light_dir, np.array([list(light_col)], np.float32),
                                       ambient, k_diffuse, background_col="white", with_mask=True)
    return np.asarray(target_albedo, np.float32) * shading, shading


def reconstruct(rec, target_compos, max_epochs=10, inner_step=200, log=print):
    """The coarse-to-fine search of :446-546: every epoch five pose hypotheses around the current best are optimised
    for `inner_step` steps; the one with the lowest loss seeds the next epoch with half the pose range.
    Returns (best latents dict, loss history)."""
    B = rec.B
    target = np.tile(np.asarray(target_compos, np.float32), (B, 1, 1, 1))
    phi_range, theta_range = 60.0, 30.0
    best, history = None, []
    for epoch in range(max_epochs):
        if epoch == 0:
            params = create_param_center(B, phi_mid=270, phi_range=phi_range, theta_mid=90, theta_range=theta_range)
            rec.assign(vector=np.full((B, rec.dec_spec.z_dim), 0.5, np.float32), param=params,
                       texture=np.random.randn(B, rec.tex_spec.z_dim).astype(np.float32),
                       light=(np.linspace(230, 320, num=B) * math.pi / 180.0)[:, None])
        else:
            phi_range /= 2
            theta_range /= 2
            params = create_param_center(B, phi_mid=best["param_deg"][0], phi_range=phi_range,
                                         theta_mid=best["param_deg"][1], theta_range=theta_range)
            rec.assign(vector=np.tile(best["vector"][None], (B, 1)), param=params,
                       texture=np.tile(best["texture"][None], (B, 1)), light=np.tile(best["light"][None], (B, 1)))
        for _ in range(inner_step):
            rec.step(target)
        with torch.no_grad():                                              # :524-530, the losses after the last update
            compos, _, _, _ = rec.forward()
            final = rec.recon_loss(compos, target).cpu().numpy()
        vals = rec.values()
        i = int(np.argmin(final))
        deg = vals["param"][i] * 180.0 / math.pi
        best = {"vector": vals["vector"][i], "texture": vals["texture"][i], "light": vals["light"][i],
                "param_deg": np.array([deg[0], 90 - deg[1], 1.0]), "loss": float(final[i])}
        history.append(final)
        log("epoch %d best hypothesis %d loss %.6f pose (%.1f, %.1f)" % (epoch, i, best["loss"], best["param_deg"][0], best["param_deg"][1]))
    return best, history
