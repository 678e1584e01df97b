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
(`ops.TrainContext(frozen=True)`: no filter / bias / alpha gradients are computed).  RenderNet_pretrained is the
two-head texture net of `rendernet_amd.texture` (same layers; the reference only spells some scopes differently --
`pretrained_key_map` translates the keys of its `*.txt.npz` weight folders).
"""
from dataclasses import dataclass
import math

import numpy as np
import torch

from . import _lib as L
from . import ops
from . import variables as V
from .texture import TextureSpec, texture_variable_shapes, init_texture_weights, decoder_texture, RenderNetTexture
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


def decoder_3d_pretrained(z_in, spec=None, taps=None):
    """Reconstruct_RenderNet_Face.py:31-72 (ELU / sigmoid run in the transposed convs' epilogues)."""
    s = spec or ShapeDecoderSpec()
    st = V.get_default_store()
    B = z_in.shape[0]
    with st.variable_scope('g_zP'):
        zP = LU.fully_connected(z_in, s.base ** 3 * s.chans[0], scope='g_gc1')
    net = zP.reshape(B, s.base, s.base, s.base, s.chans[0])
    for i in range(1, len(s.chans)):
        with st.variable_scope('g_conv%d' % i):
            net = LU.conv3d_transpose(net, s.chans[i], kernel_size=[4, 4, 4], stride=[2, 2, 2], pad="SAME",
                                      scope='g_conv%d' % i, elu=True)
        if taps is not None:
            taps["gen%d" % i] = net
    return LU.conv3d_transpose(net, 1, kernel_size=[4, 4, 4], stride=[1, 1, 1], pad="SAME", scope='g_conv%d' % len(s.chans),
                               sigmoid=True)


# ---------------------------------------------------------------------------------------------
# pretrained weight folders (tools/model_util.py:26-39): one `<key>.txt.npz` per tensor, arr_0
# ---------------------------------------------------------------------------------------------
def pretrained_key_map(tex_spec=None, dec_spec=None):
    """{key in the reference's weight dicts: variable name here}.  Keys are the TF variable names with '/' -> '_' and
    the outer 'encoder' / 'texture_encoder' scope dropped (Reconstruct_RenderNet_Face.py:40-326); the pretrained
    graph spells a few scopes differently from the training script whose names this package uses."""
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
        if outer == "texture_encoder":
            inner = inner.replace("e_tex_fc1/fully_connected", "e_tex_dc1_g_gc1").replace("e_tex_fc1", "e_tex_dc1")
            inner = inner.replace("conv3d_transpose", "conv2d_transpose")
        else:
            inner = inner.replace("projection_unit/Conv", "e_conv4_e_conv4").replace("projection_unit", "e_conv4")
            inner = head_alias.get(inner, inner)
        m[inner.replace('/', '_') + "_" + leaf] = name
    return m


def state_from_pretrained(weight_dict_rendernet, weight_dict_decoder, tex_spec=None, dec_spec=None):
    """Weight dicts of `load_weights` -> {variable name: ndarray} for Reconstructor(weights=...).  Raises on a missing key."""
    km = pretrained_key_map(tex_spec, dec_spec)
    src = dict(weight_dict_rendernet)
    src.update(weight_dict_decoder)
    missing = [k for k in km if k not in src]
    if missing:
        raise KeyError("pretrained weights are missing %d tensors, e.g. %s" % (len(missing), missing[:4]))
    return {name: np.asarray(src[key], np.float32) for key, name in km.items()}


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

    def __init__(self, tex_spec=None, dec_spec=None, weights=None, batch_size=5, device="cuda", seed=1234,
                 light_elevation_deg=105.0, light_col=(1.0, 1.0, 1.0), ambient=0.0, k_diffuse=1.0,
                 shape_eta=0.8, pose_eta=0.01, tex_eta=0.8, light_eta=0.4):
        self.tex_spec = (tex_spec or TextureSpec()).check()
        self.dec_spec = dec_spec or ShapeDecoderSpec()
        if self.dec_spec.size != self.tex_spec.size:
            raise ValueError("shape decoder emits %d^3, the renderer expects %d^3" % (self.dec_spec.size, self.tex_spec.size))
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("rendernet_amd.Reconstructor needs a HIP device; there is no CPU path")
        self.B = int(batch_size)
        self.store = V.VariableStore(self.device, seed)
        if weights is None:
            weights = dict(init_texture_weights(self.tex_spec, seed))
            weights.update(init_shape_decoder_weights(self.dec_spec, seed + 1))
        self.store.load_state_dict(weights)
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
            with ops.training(self.ctx):
                shape = decoder_3d_pretrained(lat["vector"], self.dec_spec, taps)                                  # :356
                tex = decoder_texture(lat["texture"], ts, taps)                                                    # :357
                # :360-361 + :363-364 + :366 -- both resamplers and the concat in one pass
                net_in = rotation_resampling_concat_to_image(shape, tex, lat["param"], size=ts.size, new_size=ts.new_size)
                img, nrm = RenderNetTexture(net_in, prob=1.0, spec=ts, taps=taps)                                  # :367
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
