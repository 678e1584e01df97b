"""Mirror of the NumPy half of the reference's tools/Phong_shading.py used by the demo
(:138-148, :162-228, :247-253).  The composite runs as a HIP elementwise kernel."""
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


def np_phong_composite(images_in, light_dir, light_col, ambient_in, k_diffuse, background_col="Black", with_mask=True):
    """tools/Phong_shading.py:202-228, black background with mask (the branch the demo takes,
    RenderNet_demo.py:54-56).  images_in: HIP tensor or ndarray [B,H,W,3]; returns the same kind."""
    if background_col.lower() != "black" or not with_mask:
        raise NotImplementedError("only the demo's black-background masked composite is on the HIP path")
    as_np = not isinstance(images_in, torch.Tensor)
    img = torch.as_tensor(np.asarray(images_in, np.float32)).cuda() if as_np else images_in
    B = img.shape[0]
    ld = torch.as_tensor(np.broadcast_to(np.asarray(light_dir, np.float32), (B, 3)).copy()).to(img.device)
    lc = torch.as_tensor(np.broadcast_to(np.asarray(light_col, np.float32), (B, 3)).copy()).to(img.device)
    out = ops.phong_composite(img, ld, lc, ambient_in, k_diffuse)
    return out.cpu().numpy() if as_np else out
