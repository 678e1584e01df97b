"""Mirror of the TF half of the reference's tools/resampling_voxel_grid.py (lines 370-632) on
the fused HIP resampler (rn_resample_fwd)."""
from .. import ops


def tf_rotation_resampling(voxel_array, view_params, size=64, new_size=128):
    """tools/resampling_voxel_grid.py:616-632.  voxel_array [B,size,size,size,C] (HIP tensor),
    view_params [B,3] = (azimuth, elevation, scale) in radians.  Returns the rotated grid
    [B,new_size,new_size,new_size,C] in the reference's raw [b,z,y,x,c] order."""
    if voxel_array.shape[1] != size:
        raise ValueError("voxel grid is %d^3 but size=%d" % (voxel_array.shape[1], size))
    if view_params.shape[1] != 3:
        raise ValueError("view_params must be [B,3] (azimuth, elevation, scale); the 2-column branch "
                         "is dead code in the reference (resampling_voxel_grid.py:551,625)")
    return ops.resample(voxel_array, view_params, new_size, None, image_layout=False)


tf_rotation_translation_resampling = tf_rotation_resampling   # :634-650 is the same function body


def rotation_resampling_to_image(voxel_array, view_params, size=64, new_size=128, window=None):
    """The fused form used by the renderer: tf_rotation_resampling (:616) +
    tf_transform_voxel_to_match_image (tools/model_util.py:41-49) + the voxel crop of
    tf_random_crop_voxel_image (tools/model_util.py:95-98) in one pass.
    window = (row0, col0, rows, cols) or None for the whole grid."""
    if voxel_array.shape[1] != size:
        raise ValueError("voxel grid is %d^3 but size=%d" % (voxel_array.shape[1], size))
    return ops.resample(voxel_array, view_params, new_size, window, image_layout=True)


def rotation_resampling_concat_to_image(voxel_a, voxel_b, view_params, size=64, new_size=128, window=None):
    """`tf.concat([transform(tf_rotation_resampling(a, p)), transform(tf_rotation_resampling(b, p))], axis=4)` of the face
    renderer (RenderNet_Texture_Face_Normal.py:165-178; Reconstruct_RenderNet_Face.py:360-366) in one pass: both volumes
    are sampled with the same coordinates and the concatenated tensor is written directly."""
    if voxel_a.shape[1] != size or voxel_b.shape[1] != size:
        raise ValueError("voxel grids are %d^3 / %d^3 but size=%d" % (voxel_a.shape[1], voxel_b.shape[1], size))
    return ops.resample_concat(voxel_a, voxel_b, view_params, new_size, window, image_layout=True)


def tf_resampling_affine(voxel_array, m_inv, new_size=128, image_layout=False, window=None):
    """tf_resampling (:564-614) given the already inverted matrices total_M[:, 0:3, :] (:601-602)."""
    return ops.resample(voxel_array, m_inv.reshape(m_inv.shape[0], 12), new_size, window,
                        image_layout=image_layout, affine=True)
