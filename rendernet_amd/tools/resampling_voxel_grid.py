"""Mirror of the TF half of the reference's tools/resampling_voxel_grid.py (lines 370-632) on
the fused HIP resampler (rn_resample_fwd)."""
from .. import ops


def _pose3(view_params):
    """The reference reads columns 0, 1, 2 of view_params (azimuth, elevation, scale: tf_rotation_around_grid_centroid, :526-528, :555) and
    nothing else: a [B,5] pose (azimuth, elevation, scale, shiftX, shiftY -- the docstring of :634-641) renders like its first three
    columns, the shifts are silently unused.  Two columns: `tf.shape(view_params)[1] == 2` is a Python comparison of a tensor with an
    int, always False (:551, :625, :643), so the reference goes on to index column 2 and fails at graph construction -- an error here."""
    if view_params.dim() != 2 or view_params.shape[1] < 3:
        raise ValueError("view_params must be [B,>=3] (azimuth, elevation, scale, ...), got %s; the 2-column branch is dead code in "
                         "the reference (resampling_voxel_grid.py:551,625,643)" % (tuple(view_params.shape),))
    return view_params if view_params.shape[1] == 3 else view_params[:, :3].contiguous()


def tf_rotation_resampling(voxel_array, view_params, size=64, new_size=128):
    """tools/resampling_voxel_grid.py:616-632.  voxel_array [B,size,size,size,C] (HIP tensor),
    view_params [B,3] = (azimuth, elevation, scale) in radians (further columns are ignored, as in the reference).  Returns
    the rotated grid [B,new_size,new_size,new_size,C] in the reference's raw [b,z,y,x,c] order."""
    if voxel_array.shape[1] != size:
        raise ValueError("voxel grid is %d^3 but size=%d" % (voxel_array.shape[1], size))
    return ops.resample(voxel_array, _pose3(view_params), new_size, None, image_layout=False)


def tf_rotation_translation_resampling(voxel_array, view_params, size=64, new_size=128):
    """tools/resampling_voxel_grid.py:634-650: documented for [B,5] poses (azimuth, elevation, scale, shiftX, shiftY), but the body is
    that of tf_rotation_resampling -- the two shifts are never read (see _pose3)."""
    return tf_rotation_resampling(voxel_array, view_params, size=size, new_size=new_size)


def rotation_resampling_to_image(voxel_array, view_params, size=64, new_size=128, window=None):
    """The fused form used by the renderer: tf_rotation_resampling (:616) +
    tf_transform_voxel_to_match_image (tools/model_util.py:41-49) + the voxel crop of
    tf_random_crop_voxel_image (tools/model_util.py:95-98) in one pass.
    window = (row0, col0, rows, cols) or None for the whole grid."""
    if voxel_array.shape[1] != size:
        raise ValueError("voxel grid is %d^3 but size=%d" % (voxel_array.shape[1], size))
    return ops.resample(voxel_array, _pose3(view_params), new_size, window, image_layout=True)


def rotation_resampling_concat_to_image(voxel_a, voxel_b, view_params, size=64, new_size=128, window=None):
    """`tf.concat([transform(tf_rotation_resampling(a, p)), transform(tf_rotation_resampling(b, p))], axis=4)` of the face
    renderer (RenderNet_Texture_Face_Normal.py:165-178; Reconstruct_RenderNet_Face.py:360-366) in one pass: both volumes
    are sampled with the same coordinates and the concatenated tensor is written directly."""
    if voxel_a.shape[1] != size or voxel_b.shape[1] != size:
        raise ValueError("voxel grids are %d^3 / %d^3 but size=%d" % (voxel_a.shape[1], voxel_b.shape[1], size))
    return ops.resample_concat(voxel_a, voxel_b, _pose3(view_params), new_size, window, image_layout=True)


def tf_resampling_affine(voxel_array, m_inv, new_size=128, image_layout=False, window=None):
    """tf_resampling (:564-614) given the already inverted matrices total_M[:, 0:3, :] (:601-602)."""
    return ops.resample(voxel_array, m_inv.reshape(m_inv.shape[0], 12), new_size, window,
                        image_layout=image_layout, affine=True)
