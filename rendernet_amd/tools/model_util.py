"""Mirror of the reference's tools/model_util.py: the layout transform (:41-49), the crop helpers of the training
graphs (:77-100, :102-160) and the pretrained-weight folder loader (:26-39).  Host-side; the renderer / trainers fuse the
transform and the voxel crop into the resampler (`window`) and only use these for the image side and in tests."""
import glob
import os

import numpy as np
import torch


def load_weights(weight_dir):
    """tools/model_util.py:26-39: a folder of `<layer name>.txt.npz` files -> {layer name: arr_0}."""
    out = {}
    for path in glob.glob(os.path.join(weight_dir, "*.txt.npz")):
        with np.load(path) as data:
            out[os.path.basename(path).split('.')[0]] = data['arr_0']
    return out


def tf_transform_voxel_to_match_image(tensor_voxel):
    """tools/model_util.py:41-49: transpose dims 1<->2 then reverse dim 1.  (The renderer does not
    call this: the resampler writes the transformed layout directly.)"""
    return torch.flip(tensor_voxel.permute(0, 2, 1, 3, 4), dims=[1]).contiguous()


def tf_random_crop_voxel_image(voxels, images, patch_size, start_point=None, generator=None):
    """tools/model_util.py:77-100.  One random (row, col) start for the whole batch; voxels
    [:, r:r+p, c:c+p, :, :], images scaled by image_dim/voxel_dim.  `start_point` overrides the
    random draw (the reference draws with seed=None)."""
    voxel_dim, image_dim = voxels.shape[1], images.shape[1]
    f = image_dim // voxel_dim
    if start_point is None:
        start_point = torch.randint(0, voxel_dim - patch_size + 1, (2,), generator=generator).tolist()
    r, c = int(start_point[0]), int(start_point[1])
    vp = voxels[:, r:r + patch_size, c:c + patch_size].contiguous()
    ip = images[:, f * r:f * (r + patch_size), f * c:f * (c + patch_size)].contiguous()
    return vp, ip


def _crop_all(volumes, maps, patch_size, start_point, generator):
    voxel_dim, image_dim = volumes[0].shape[1], maps[0].shape[1]
    if patch_size == voxel_dim:                                           # :113-114: identity
        return tuple(volumes) + tuple(maps)
    f = image_dim // voxel_dim
    if start_point is None:
        start_point = torch.randint(0, voxel_dim - patch_size + 1, (2,), generator=generator).tolist()
    r, c = int(start_point[0]), int(start_point[1])
    return tuple(v[:, r:r + patch_size, c:c + patch_size].contiguous() for v in volumes) + \
        tuple(m[:, f * r:f * (r + patch_size), f * c:f * (c + patch_size)].contiguous() for m in maps)


def tf_random_crop_voxel_texture_image(voxels, texture, images, patch_size, start_point=None, generator=None):
    """tools/model_util.py:102-128: ONE window for the voxel grid, the texture grid and the image."""
    return _crop_all((voxels, texture), (images,), patch_size, start_point, generator)


def tf_random_crop_voxel_texture_image_normal(voxels, texture, images, normals, patch_size, start_point=None, generator=None):
    """tools/model_util.py:130-160: ONE window for the voxel grid, the texture grid, the image and the normal map
    (the crop of RenderNet_Texture_Face_Normal.py:175)."""
    return _crop_all((voxels, texture), (images, normals), patch_size, start_point, generator)
