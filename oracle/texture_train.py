"""CPU restatement of the texture + normal net's training graph, RenderNet_Texture_Face_Normal.py:152-186 --
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  rotated_models  = transform(tf_rotation_resampling(model_in))                          (:165-166)
  texture_rotated = transform(tf_rotation_resampling(decoder_texture(texture_in)))       (:169-172)
  one crop window for voxels, texture, image and normal (tools/model_util.py:103-152)    (:175)
  images_pred, normal_pred = RenderNet(concat([crop_vox, crop_tex], 4))                  (:178-179)
  loss = mean_squared_error(image) + mean_squared_error(normal)                          (:182-183)

The gradient reaches the texture decoder THROUGH the resampler, so the resampler is restated here with torch ops
(gather with the indices / weights of oracle/resample.py, differentiable w.r.t. the voxel values) and is checked
against the NumPy oracle bit for bit in tests/test_oracle_train.py.
"""
import numpy as np
import torch

from . import resample as R
from . import texture_net as TN


def resample_torch(vox_t, M_inv, new_size, mode="tf"):
    """oracle/resample.py::resampling_affine on a torch voxel tensor [B,S,S,S,C]; same float32 operation order."""
    B, S, C = vox_t.shape[0], vox_t.shape[1], vox_t.shape[4]
    outs = []
    for b in range(B):
        x, y, z = R.source_coords(M_inv[b], new_size, mode)
        x0 = np.floor(x).astype(np.int64); y0 = np.floor(y).astype(np.int64); z0 = np.floor(z).astype(np.int64)
        x1, y1, z1 = x0 + 1, y0 + 1, z0 + 1
        cl = lambda v: np.clip(v, 0, S - 1)
        x0, x1, y0, y1, z0, z1 = cl(x0), cl(x1), cl(y0), cl(y1), cl(z0), cl(z1)
        f32 = np.float32
        ax, bx = x1.astype(f32) - x, x - x0.astype(f32)
        ay, by = y1.astype(f32) - y, y - y0.astype(f32)
        az, bz = z1.astype(f32) - z, z - z0.astype(f32)
        flat = vox_t[b].reshape(-1, C)
        ix = lambda zz, yy, xx: torch.from_numpy((zz * S + yy) * S + xx)
        wt = lambda w: torch.from_numpy(w.astype(f32))[:, None]
        out = wt(ax * ay * az) * flat[ix(z0, y0, x0)]
        out = out + wt(ax * by * az) * flat[ix(z0, y1, x0)]
        out = out + wt(bx * ay * az) * flat[ix(z0, y0, x1)]
        out = out + wt(bx * by * az) * flat[ix(z0, y1, x1)]
        out = out + wt(ax * ay * bz) * flat[ix(z1, y0, x0)]
        out = out + wt(ax * by * bz) * flat[ix(z1, y1, x0)]
        out = out + wt(bx * ay * bz) * flat[ix(z1, y0, x1)]
        out = out + wt(bx * by * bz) * flat[ix(z1, y1, x1)]
        outs.append(out.reshape(new_size, new_size, new_size, C))
    return torch.stack(outs)


def to_image_layout(t):
    """tools/model_util.py:41-49 on a torch tensor: transpose dims 1<->2 then reverse dim 1."""
    return torch.flip(t.permute(0, 2, 1, 3, 4), dims=[1])


def loss_and_grads(voxels, textures, M_inv, images, normals, weights, start, patch, size, new_size, tex_res, n_res, c0=4):
    """Returns (loss, {name: grad}, (image_pred, normal_pred)).  M_inv [B,3,4] are the matrices the HIP path used
    (taken from rn_pose_to_affine so that both sides resample with the same coordinates)."""
    wt = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).requires_grad_(True) for k, v in weights.items()}
    vox = torch.from_numpy(np.ascontiguousarray(voxels, dtype=np.float32))
    geo = to_image_layout(resample_torch(vox, M_inv, new_size, "ordered"))
    tex_vol = TN.decoder_texture_torch(torch.from_numpy(np.asarray(textures, np.float32)), wt, tex_res, c0)
    tex_rot = to_image_layout(resample_torch(tex_vol, M_inv, new_size, "ordered"))
    r, c = int(start[0]), int(start[1])
    x = torch.cat([geo, tex_rot], dim=4)[:, r:r + patch, c:c + patch]
    img, nrm = TN.rendernet_texture_forward_torch(x.contiguous(), wt, n_res[0], n_res[1], n_res[2])
    ti = torch.from_numpy(np.ascontiguousarray(images[:, 4 * r:4 * (r + patch), 4 * c:4 * (c + patch)], dtype=np.float32))
    tn = torch.from_numpy(np.ascontiguousarray(normals[:, 4 * r:4 * (r + patch), 4 * c:4 * (c + patch)], dtype=np.float32))
    loss = torch.mean((ti - img) ** 2) + torch.mean((tn - nrm) ** 2)
    loss.backward()
    grads = {k: (v.grad.numpy().copy() if v.grad is not None else np.zeros_like(weights[k])) for k, v in wt.items()}
    return float(loss.item()), grads, (img.detach().numpy(), nrm.detach().numpy())
