"""CPU restatement of the reference's training step, RenderNet_Shader.py:154-167 + :239-240 --
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  crop     tf_random_crop_voxel_image (tools/model_util.py:77-100): voxel [:, r:r+p, c:c+p], image
           [:, 4r:4(r+p), 4c:4(c+p)] with ONE (r, c) for the batch
  loss     greyscale: mean_b(-sum(t*log(1e-6+p) + (1-t)*log(1e-6+1-p)))   (:160-161)
           RGB:       tf.losses.mean_squared_error                         (:163)
  grads    what tf.gradients derives -- here torch's CPU autograd over oracle/rendernet.py (an
           implementation independent of the HIP dgrad/wgrad kernels)
  update   tf.train.AdamOptimizer(lr, beta1=0.5) with exponential_decay(e_eta, step, decay_steps, 0.96,
           staircase=True) (:165-166); TF's Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
           m = b1*m+(1-b1)*g, v = b2*v+(1-b2)*g*g, p -= lr_t*m/(sqrt(v)+eps), eps = 1e-8 un-scaled.
"""
import math

import numpy as np
import torch

from . import layers as L
from . import rendernet as ON


def crop_voxel_image(vox_img, image, start, patch):
    """tools/model_util.py:95-99 given the drawn start point (r, c)."""
    r, c = int(start[0]), int(start[1])
    return vox_img[:, r:r + patch, c:c + patch], image[:, 4 * r:4 * (r + patch), 4 * c:4 * (c + patch)]


def mse_loss(pred, target):
    return torch.mean((pred - target) ** 2)


def loss_and_grads(net_in, target, weights, n_res1=10, n_res2=10, n_res3=5, greyscale=True, taps=None,
                   dtype=np.float32):
    """net_in [B,p,p,N,1] (already resampled + cropped), target [B,4p,4p,ch], weights {tf_name: ndarray}.
    Returns (loss float, {tf_name: gradient ndarray}, prediction ndarray).  dtype=float64 is used by the
    finite-difference check that pins this function (tests/test_oracle_train.py)."""
    wt = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=dtype)).requires_grad_(True) for k, v in weights.items()}
    x = torch.from_numpy(np.ascontiguousarray(net_in, dtype=dtype))
    t = torch.from_numpy(np.ascontiguousarray(target, dtype=dtype))
    pred = ON.rendernet_forward_torch(x, wt, taps, n_res1, n_res2, n_res3)
    loss = L.bce_loss(pred, t) if greyscale else mse_loss(pred, t)
    loss.backward()
    grads = {k: (v.grad.numpy().copy() if v.grad is not None else np.zeros_like(weights[k])) for k, v in wt.items()}
    return float(loss.item()), grads, pred.detach().numpy()


def exponential_decay(lr0, step, decay_steps, rate=0.96):
    return lr0 * rate ** math.floor(step / float(decay_steps))


class Adam:
    """tf.train.AdamOptimizer state over a dict of arrays (float32 arithmetic like TF's kernels)."""

    def __init__(self, e_eta=1e-5, decay_steps=100000, beta1=0.5, beta2=0.999, epsilon=1e-8):
        self.e_eta, self.decay_steps = e_eta, decay_steps
        self.b1, self.b2, self.eps = beta1, beta2, epsilon
        self.t = 0
        self.m, self.v = {}, {}

    def apply(self, weights, grads):
        lr = exponential_decay(self.e_eta, self.t, self.decay_steps)
        self.t += 1
        lr_t = np.float32(lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t))
        b1, b2, eps = np.float32(self.b1), np.float32(self.b2), np.float32(self.eps)
        out = {}
        for k, p in weights.items():
            g = grads[k].astype(np.float32)
            m = self.m.get(k, np.zeros_like(p)); v = self.v.get(k, np.zeros_like(p))
            m = b1 * m + (np.float32(1) - b1) * g
            v = b2 * v + (np.float32(1) - b2) * g * g
            self.m[k], self.v[k] = m, v
            out[k] = (p - lr_t * m / (np.sqrt(v) + eps)).astype(np.float32)
        return out
