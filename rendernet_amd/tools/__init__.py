"""Mirrors of the reference's `tools/` modules on the MI355X path (same function names and
argument meaning): layer_util, resampling_voxel_grid, model_util, binvox_rw, Phong_shading."""
