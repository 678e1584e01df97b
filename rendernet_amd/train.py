"""The training step of the Phong-shader net on the MI355X path (BASELINE config 4).

Mirrors RenderNet_Shader.py:135-167 + :239-240: resample -> transform -> random crop (one window for
the whole batch, tools/model_util.py:77-100) -> RenderNet(is_training=True) -> BCE (greyscale) or
MSE (RGB) reconstruction loss -> `tf.train.AdamOptimizer(lr, beta1=0.5).minimize` with
`tf.train.exponential_decay(e_eta, global_step, decay_steps, 0.96, staircase=True)`.

MI355X design: forward, dgrad, wgrad, epilogue backward, loss and Adam are HIP kernels behind the C
ABI; parameters, gradients and both Adam moments live in four flat fp32 buffers (237.3 M floats =
949 MB each); wgrad kernels accumulate straight into the flat gradient buffer.  Data parallelism is one
process per GPU with the batch sharded by frame: the loss is a mean over the GLOBAL batch, so ranks
compute d(sum of their frames)/global_batch and the gradient all-reduce is a plain SUM (RCCL over
xGMI through torch.distributed, backend "nccl").  The flat gradient buffer is cut into contiguous
buckets in reverse creation order (= the order the backward finishes them); a bucket's all-reduce is
launched asynchronously the moment its last parameter gradient is complete, overlapping the remaining
backward; the optimiser step waits for all buckets.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L
from . import ops
from . import variables as V
from .shader import RenderNet, ShaderSpec, init_shader_weights
from .tools.resampling_voxel_grid import rotation_resampling_concat_to_image, rotation_resampling_to_image


def exponential_decay(lr0, step, decay_steps, rate=0.96, staircase=True):
    """tf.train.exponential_decay (RenderNet_Shader.py:165)."""
    e = step / float(decay_steps)
    return lr0 * rate ** (math.floor(e) if staircase else e)


def adam_lr_t(lr, t, beta1, beta2):
    """TF's AdamOptimizer folds the bias corrections into the step size (t = 1 on the first update)."""
    return lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)


def plan_buckets(layout, order, bucket_floats):
    """Cut the flat gradient buffer into contiguous buckets following `order` (the parameter names in
    the order their gradients complete, i.e. reverse creation order).  Returns [(lo, hi, [names])] with
    lo/hi float offsets; every parameter belongs to exactly one bucket.  Pure host logic (CPU-testable)."""
    buckets, cur, lo, hi = [], [], None, None
    for n in order:
        o, k = layout[n]
        e = o + (k + 3) // 4 * 4
        if cur and (max(hi, e) - min(lo, o)) > bucket_floats:
            buckets.append((lo, hi, cur))
            cur, lo, hi = [], None, None
        cur.append(n)
        lo = o if lo is None else min(lo, o)
        hi = e if hi is None else max(hi, e)
    if cur:
        buckets.append((lo, hi, cur))
    return buckets


class GradBuckets:
    """Bucketed, overlapped SUM all-reduce of a flat gradient buffer.  `ready(name)` is called as each
    parameter's gradient completes; when a bucket is complete its slice is all-reduced asynchronously
    (on the process group's own stream, ordered after the work already queued on the current stream).
    Backend-agnostic: RCCL ("nccl") on the GPUs, gloo in the CPU tests."""

    def __init__(self, flat_grad, layout, order, bucket_mb=100.0, group=None):
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = plan_buckets(layout, order, int(bucket_mb * 1e6 / 4))
        self.bucket_of = {n: i for i, (_, _, ns) in enumerate(self.buckets) for n in ns}
        self.reset()

    def reset(self):
        self.pending = [len(ns) for _, _, ns in self.buckets]
        self.seen = set()
        self.handles = []
        self.launched = []

    def ready(self, name):
        if name in self.seen:
            return
        self.seen.add(name)
        i = self.bucket_of[name]
        self.pending[i] -= 1
        if self.pending[i] == 0:
            self._launch(i)

    def _launch(self, i):
        self.launched.append(i)
        if self.world > 1:
            lo, hi, _ = self.buckets[i]
            self.handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Launch whatever is still pending (parameters that received no gradient) and wait for all."""
        for i, p in enumerate(self.pending):
            if p > 0:
                self.pending[i] = 0
                self._launch(i)
        for h in self.handles:
            h.wait()
        self.handles = []


class _TrainerBase:
    """Parameters / gradients / Adam moments of one net on one GPU in four flat buffers, the gradient buckets and the
    optimiser step.  With torch.distributed initialised every rank holds a full replica and a shard of the batch."""

    def __init__(self, spec, weights, device="cuda", seed=1234, e_eta=1e-5, decay_steps=100000,
                 beta1=0.5, beta2=0.999, epsilon=1e-8, keep_prob=1.0, bucket_mb=100.0, group=None, gemm=None):
        if gemm is not None and gemm not in ops.GEMM_MODES:
            raise ValueError("gemm=%r: expected one of %s" % (gemm, ", ".join(ops.GEMM_MODES)))
        self.gemm = gemm                  # this trainer's multiply-stage mode (ops.gemm_mode); None = the process default
        self.spec = spec
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("rendernet_amd trainers need a HIP device; there is no CPU training path")
        self.store = V.VariableStore(self.device, seed)
        self.store.load_state_dict(weights)
        self.names = list(self.store.vars.keys())
        self.param, self.layout = self.store.flatten(self.names)
        self.grad = torch.zeros_like(self.param)
        self.m = torch.zeros_like(self.param)
        self.v = torch.zeros_like(self.param)
        self.grad_views = {n: self.grad[o:o + k].view(self.store.vars[n].shape) for n, (o, k) in self.layout.items()}
        self._ptr_name = {self.store.vars[n].data_ptr(): n for n in self.names}
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = GradBuckets(self.grad, self.layout, list(reversed(self.names)), bucket_mb, group)
        self.ctx = ops.TrainContext({self.store.vars[n].data_ptr(): g for n, g in self.grad_views.items()},
                                    on_ready=lambda p: self.buckets.ready(self._ptr_name[p]), device=self.device)
        self.e_eta, self.decay_steps = float(e_eta), int(decay_steps)
        self.beta1, self.beta2, self.epsilon = float(beta1), float(beta2), float(epsilon)
        self.keep_prob = float(keep_prob)
        # dropout masks are a pure function of (seed, rank, global step, position of the dropout site): ranks of a
        # data-parallel job draw different masks for their shards (tf.nn.dropout draws per sample), every step draws new
        # ones, and a resumed run (global_step comes back from the checkpoint) continues the sequence instead of replaying it
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.dropout_seed = ops.mix_seed(seed, self.rank)
        self.global_step = 0
        self.checkpoint_extra = {}        # what load_checkpoint found beside the weights and the optimiser state
        self.loss_buf = torch.zeros(1, dtype=torch.float64, device=self.device)

    def _loss_grad(self, pred, target, global_batch, mse):
        """Loss kernel: value accumulated into self.loss_buf, returns d(loss)/d(pred)."""
        tgt = target.contiguous()
        dpred = torch.empty_like(pred)
        n = pred.numel()
        divisor = float(global_batch) if not mse else float(n // pred.shape[0] * global_batch)
        L.check(L.lib().rn_loss_fwd_bwd(L.ptr(pred), L.ptr(tgt), L.ptr(dpred), self.loss_buf.data_ptr(), n, divisor,
                                        1 if mse else 0, L.stream_ptr()), "rn_loss_fwd_bwd")
        return dpred

    def apply_gradients(self):
        self.global_step += 1
        lr = exponential_decay(self.e_eta, self.global_step - 1, self.decay_steps)
        lr_t = adam_lr_t(lr, self.global_step, self.beta1, self.beta2)
        L.check(L.lib().rn_adam_step(L.ptr(self.param), L.ptr(self.grad), L.ptr(self.m), L.ptr(self.v), self.param.numel(),
                                     lr_t, self.beta1, self.beta2, self.epsilon, 1.0, L.stream_ptr()), "rn_adam_step")
        self.store.repack_all()

    def _seed_dropout(self):
        """Called at the start of every forward through the trainer's graph: stream ids (global_step << 16) + k for the k-th
        dropout site, so an extra forward between two steps (the sample render every 600 steps) re-draws the masks of the
        coming step instead of shifting every later one."""
        ops.seed_dropout(self.dropout_seed, self.global_step << 16)

    def _begin_step(self):
        self.grad.zero_()
        self.loss_buf.zero_()
        self.buckets.reset()

    def _finish_step(self):
        self.buckets.finish()
        if self.world > 1:
            dist.all_reduce(self.loss_buf, op=dist.ReduceOp.SUM, group=self.group)
        self.apply_gradients()
        return self.loss_buf[0].clone()

    def state_dict(self):
        """The weights by TF variable name (what `sess_saver.save` keeps for inference)."""
        return self.store.state_dict()

    # -- checkpoint / resume ------------------------------------------------------------------
    # The reference's tf.train.Supervisor saves the whole graph state -- variables, the Adam slots `m` / `v` and
    # global_step (RenderNet_Shader.py:171-185, save_model_secs = checkpoint_secs) -- and restores it on restart, so the
    # learning-rate staircase, Adam's bias correction and the patch-size schedule continue where they stopped.
    def checkpoint(self, epoch=0, extra=None):
        """Everything a restart needs: weights + '__adam_m__' / '__adam_v__' (the flat moment buffers, layout = creation
        order of the variables, each 16-byte aligned) + '__global_step__' + '__epoch__', plus `extra` ({name: array}, stored
        as '__name__': the training scripts keep their validation history 'l1_all' there).  A checkpoint written in the
        middle of an epoch resumes at the START of that epoch (the tar stream's position is not part of the state) with the
        advanced global_step: the first batches of the epoch are seen again -- the reference restarts its epoch loop from 0
        after a Supervisor restore (RenderNet_Shader.py:193-202), so this replays less than it does."""
        sd = self.store.state_dict()
        for k, v in (extra or {}).items():
            sd["__%s__" % k] = np.asarray(v)
        sd["__adam_m__"] = self.m.cpu().numpy()
        sd["__adam_v__"] = self.v.cpu().numpy()
        sd["__global_step__"] = np.int64(self.global_step)
        sd["__epoch__"] = np.int64(epoch)
        return sd

    def save_checkpoint(self, path, epoch=0, extra=None):
        """Atomic: written next to `path` and renamed over it, so a kill mid-write leaves the previous file intact."""
        import os
        tmp = path + ".tmp.npz"
        np.savez(tmp, **self.checkpoint(epoch, extra))
        os.replace(tmp, path)

    def load_checkpoint(self, sd, strict=True):
        """Restore weights and the optimiser state; returns the epoch to resume at.  Everything is validated BEFORE anything
        is overwritten: a variable missing from `sd` or of another shape, moment buffers of another length, or only part of
        (m, v, global_step) present raise ValueError and leave the trainer untouched.  A weights-only file (no optimiser
        state at all, e.g. the per-epoch inference save) is accepted: moments and global_step restart from zero and so
        does the epoch -- the patch-size schedule must not resume late while the learning-rate schedule restarts.
        strict=False downgrades missing variables to a warning (they keep their current values)."""
        import warnings
        missing = [n for n in self.names if n not in sd]
        if missing:
            msg = "checkpoint lacks %d of %d variables (first: %s)" % (len(missing), len(self.names), missing[0])
            if strict:
                raise ValueError(msg)
            warnings.warn(msg)
        for n in self.names:
            if n not in sd:
                continue
            have_shape, want_shape = tuple(np.shape(sd[n])), tuple(self.store.vars[n].shape)
            # the SHAPE must match (same element count in another layout -- [kh,kw,Cin,Cout] vs [kh,kw,Cout,Cin] -- would load
            # silently wrong); the one documented exception: singleton axes may be dropped or added (a [C] bias saved as [1,C])
            if have_shape != want_shape and tuple(d for d in have_shape if d != 1) != tuple(d for d in want_shape if d != 1):
                raise ValueError("checkpoint variable %s has shape %s, the net's has %s" % (n, have_shape, want_shape))
        opt_keys = ("__adam_m__", "__adam_v__", "__global_step__")
        have = [k for k in opt_keys if k in sd]
        if have and len(have) != len(opt_keys):
            raise ValueError("checkpoint holds only part of the optimiser state (%s of %s)" % (have, list(opt_keys)))
        if have:
            for k in ("__adam_m__", "__adam_v__"):
                if int(np.asarray(sd[k]).size) != self.m.numel():
                    raise ValueError("checkpoint %s has %d floats, this net's flat buffer has %d (another spec?)"
                                     % (k, np.asarray(sd[k]).size, self.m.numel()))
        with torch.no_grad():
            for n in self.names:
                if n in sd:
                    self.store.vars[n].copy_(torch.as_tensor(np.asarray(sd[n], np.float32)).reshape(self.store.vars[n].shape))
            if have:
                self.m.copy_(torch.as_tensor(np.asarray(sd["__adam_m__"], np.float32)))
                self.v.copy_(torch.as_tensor(np.asarray(sd["__adam_v__"], np.float32)))
                self.global_step = int(sd["__global_step__"])
                # strict=False with variables missing: their values were NOT restored, so their moments must not be either
                # (stale m / v for weights they do not belong to) -- those slices restart from zero
                for n in missing:
                    o, k = self.layout[n]
                    self.m[o:o + k].zero_()
                    self.v[o:o + k].zero_()
            else:
                self.m.zero_()
                self.v.zero_()
                self.global_step = 0
        self.store.repack_all()
        known = set(opt_keys) | {"__epoch__"}
        self.checkpoint_extra = {k[2:-2]: np.asarray(sd[k]) for k in sd if str(k).startswith("__") and str(k).endswith("__")
                                 and k not in known}
        return int(sd["__epoch__"]) if (have and "__epoch__" in sd) else 0


class Trainer(_TrainerBase):
    """The Phong-shader net (RenderNet_Shader.py): crop -> RenderNet -> BCE (greyscale) | MSE (RGB) -> Adam."""

    def __init__(self, spec=None, weights=None, device="cuda", seed=1234, **kw):
        spec = (spec or ShaderSpec()).check()
        super().__init__(spec, weights if weights is not None else init_shader_weights(spec, seed), device, seed, **kw)
        self.mse = self.spec.out_ch != 1                      # RenderNet_Shader.py:159-163

    # -- pieces (also used one by one by the parity tests) ----------------------------------
    def forward(self, voxels, poses, patch_size=None, start_point=None, taps=None, net_in=None, is_training=True):
        """Forward through the trainer's graph: returns (prediction [b,4p,4p,ch], window).  `net_in` [b,p,p,N,C], when
        given, is an already resampled + cropped grid (voxels/poses are then ignored).  is_training=False is the
        reference's validation feed (RenderNet_Shader.py:279: dropout off)."""
        s = self.spec
        p = int(patch_size) if patch_size is not None else s.new_size
        if net_in is not None:
            net_in = torch.as_tensor(net_in, dtype=torch.float32).to(self.device).contiguous()
            if start_point is None:
                raise ValueError("net_in needs the start_point it was cropped at")
        else:
            vox = torch.as_tensor(voxels, dtype=torch.float32).to(self.device)
            pose = torch.as_tensor(poses, dtype=torch.float32).to(self.device)
        if start_point is None:
            start_point = torch.randint(0, s.new_size - p + 1, (2,)).tolist() if p != s.new_size else (0, 0)
        window = (int(start_point[0]), int(start_point[1]), p, p)
        old = V._default
        V.set_default_store(self.store)
        self._seed_dropout()
        try:
            with ops.gemm_mode(self.gemm), ops.training(self.ctx):
                if net_in is None:
                    net_in = rotation_resampling_to_image(vox, pose, size=s.size, new_size=s.new_size, window=window)
                if taps is not None:
                    taps["net_in"] = net_in
                pred = RenderNet(net_in, bool(is_training), prob=self.keep_prob, spec=s, taps=taps)
        finally:
            V._default = old
        return pred, window

    def loss_and_backward(self, pred, target_patch, global_batch):
        """Loss kernel (value accumulated into self.loss_buf, gradient w.r.t. pred) + the HIP backward."""
        pred.backward(self._loss_grad(pred, target_patch, global_batch, self.mse))

    # -- one step -----------------------------------------------------------------------------
    def step(self, voxels, poses, targets, patch_size=None, start_point=None, global_batch=None, net_in=None):
        """One optimiser step on this rank's shard (voxels [b,S,S,S,C], poses [b,3], targets [b,512,512,ch]).
        `start_point` must be the same on every rank (the reference draws one window per batch).
        Returns the loss as a 0-d float64 device tensor (global mean; identical on all ranks)."""
        b = int(targets.shape[0])
        gb = int(global_batch) if global_batch is not None else b * self.world
        self._begin_step()
        pred, (r, c, p, _) = self.forward(voxels, poses, patch_size, start_point, net_in=net_in)
        tgt = torch.as_tensor(targets, dtype=torch.float32).to(self.device)
        tgt = tgt[:, 4 * r:4 * (r + p), 4 * c:4 * (c + p), :]          # tools/model_util.py:99
        self.loss_and_backward(pred, tgt, gb)
        return self._finish_step()


class TextureTrainer(_TrainerBase):
    """The texture + normal net (RenderNet_Texture_Face_Normal.py:152-186): geometry and the decoded texture
    volume are resampled, cropped with ONE window (tools/model_util.py:103-152), concatenated and rendered by the
    two-head net; loss = MSE(image) + MSE(normal) (:182-183); Adam as for the shader.  The texture decoder is
    trained THROUGH the resampler (rn_resample_affine_bwd scatters the gradient back into the decoded volume)."""

    def __init__(self, spec=None, weights=None, device="cuda", seed=1234, **kw):
        from .texture import TextureSpec, init_texture_weights
        spec = (spec or TextureSpec()).check()
        super().__init__(spec, weights if weights is not None else init_texture_weights(spec, seed), device, seed, **kw)

    def forward(self, voxels, textures, poses, patch_size=None, start_point=None, taps=None, is_training=True):
        from .texture import decoder_texture, RenderNetTexture
        s = self.spec
        vox = torch.as_tensor(voxels, dtype=torch.float32).to(self.device)
        tex = torch.as_tensor(textures, dtype=torch.float32).to(self.device)
        pose = torch.as_tensor(poses, dtype=torch.float32).to(self.device)
        p = int(patch_size) if patch_size is not None else s.new_size
        if start_point is None:
            start_point = torch.randint(0, s.new_size - p + 1, (2,)).tolist() if p != s.new_size else (0, 0)
        window = (int(start_point[0]), int(start_point[1]), p, p)
        old = V._default
        V.set_default_store(self.store)
        self._seed_dropout()
        try:
            with ops.gemm_mode(self.gemm), ops.training(self.ctx):
                tex_vol = decoder_texture(tex, s, taps)
                net_in = rotation_resampling_concat_to_image(vox, tex_vol, pose, size=s.size, new_size=s.new_size, window=window)
                if taps is not None:
                    taps["net_in"] = net_in
                img, nrm = RenderNetTexture(net_in, prob=self.keep_prob, spec=s, taps=taps, is_training=bool(is_training))
        finally:
            V._default = old
        return img, nrm, window

    def loss_and_backward(self, img, nrm, img_patch, nrm_patch, global_batch):
        d_img = self._loss_grad(img, img_patch, global_batch, True)
        d_nrm = self._loss_grad(nrm, nrm_patch, global_batch, True)
        torch.autograd.backward([img, nrm], [d_img, d_nrm])

    def step(self, voxels, textures, poses, images, normals, patch_size=None, start_point=None, global_batch=None):
        b = int(images.shape[0])
        gb = int(global_batch) if global_batch is not None else b * self.world
        self._begin_step()
        img, nrm, (r, c, p, _) = self.forward(voxels, textures, poses, patch_size, start_point)
        crop = lambda t: torch.as_tensor(t, dtype=torch.float32).to(self.device)[:, 4 * r:4 * (r + p), 4 * c:4 * (c + p), :]
        self.loss_and_backward(img, nrm, crop(images), crop(normals), gb)
        return self._finish_step()
