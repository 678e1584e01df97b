"""Mirror of the reference's tools/Phong_shading.py: the NumPy half used by the demo (:138-228, :247-253) and the
differentiable TF half of the inverse-rendering graph (:23-130).  Every composite runs as one HIP elementwise kernel
(forward) / one more (backward)."""
import math
import numpy as np
import torch

from .. import ops


def generate_light_pos(elevation=90, azimuth=90):
    """tools/Phong_shading.py:247-253 (host, NumPy)."""
    elevation = np.array([[elevation]]) * math.pi / 180.0
    azimuth = np.array([[azimuth]]) * math.pi / 180.0
    x = -np.sin(elevation) * np.cos(azimuth)
    y = np.cos(elevation)
    z = -np.sin(elevation) * np.sin(azimuth)
    return np.hstack((x, y, z))


def _as_dev(x, device=None):
    if isinstance(x, torch.Tensor):
        return x.float()
    return torch.as_tensor(np.asarray(x, np.float32)).to(device or "cuda")


def np_phong_composite(images_in, light_dir, light_col, ambient_in, k_diffuse, background_col="Black", with_mask=True):
    """tools/Phong_shading.py:202-228 (np_mask :138-148, np_mask_white :150-160).  images_in: HIP tensor or ndarray
    [B,H,W,3]; returns the same kind.  The demo takes the black-background branch (RenderNet_demo.py:54-56), the
    inverse-rendering script shades its target with the white one (Reconstruct_RenderNet_Face.py:442)."""
    as_np = not isinstance(images_in, torch.Tensor)
    img = _as_dev(images_in)
    mode = "none" if not with_mask else ("np_black" if background_col.lower() == "black" else "np_white")
    out = ops.phong_composite(img, _as_dev(light_dir, img.device), _as_dev(light_col, img.device), ambient_in, k_diffuse, mode)
    return out.cpu().numpy() if as_np else out


# ---------------------------------------------------------------------------------------------
# The TensorFlow half (tools/Phong_shading.py:23-130): same shading, TF's mask thresholds, differentiable
# w.r.t. the normal map and the light direction (the inverse-rendering graph optimises the light azimuth).
# ---------------------------------------------------------------------------------------------
def tf_phong_composite(images_in, light_dir, light_col, ambient_in, k_diffuse, with_black_background=False, with_mask=True,
                       albedo=None):
    """tools/Phong_shading.py:88-111.  `albedo` (extension) multiplies the shading inside the same kernel:
    compos_pred = img_pred * shading (Reconstruct_RenderNet_Face.py:378)."""
    mode = "none" if not with_mask else ("tf_black" if with_black_background else "tf_white")
    return ops.phong_composite(images_in, light_dir, light_col, float(ambient_in), float(k_diffuse), mode, albedo=albedo)


def tf_phong_shading(images_in, light_dir, light_col, k_diffuse):
    """tools/Phong_shading.py:46-86: the diffuse term alone = the unmasked composite with zero ambient."""
    return ops.phong_composite(images_in, light_dir, light_col, 0.0, float(k_diffuse), "none")


def tf_generate_light_pos(batch_light_azimuth, light_elevation, batch_size):
    """tools/Phong_shading.py:113-130: [B,1] azimuths (a differentiable tensor) + one elevation -> [B,3]."""
    az = batch_light_azimuth.reshape(batch_size, 1).float()
    el = torch.full_like(az, float(light_elevation))
    return torch.cat((torch.sin(el) * torch.cos(az), torch.sin(el) * torch.sin(az), torch.cos(el)), dim=1)
