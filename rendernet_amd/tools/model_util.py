"""Mirror of the hot-path parts of the reference's tools/model_util.py (:41-49, :77-100)."""
import torch


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
