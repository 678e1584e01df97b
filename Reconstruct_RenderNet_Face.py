#!/usr/bin/env python
"""`python Reconstruct_RenderNet_Face.py <config.json>` -- the reference's inverse-rendering script
(Reconstruct_RenderNet_Face.py) on the MI355X path, same JSON keys (config_reconstruction_RenderNet.json:1-23).

Reads the target albedo / normal PNGs, shades the target with the NumPy Phong composite at the ground-truth light
(:430-444), then runs the coarse-to-fine latent search of :446-546: `max_epochs` rounds of five pose hypotheses, each
optimised for `inner_step` gradient steps on (shape code, pose, texture code, light azimuth) through the frozen
decoders + renderer (`rendernet_amd.reconstruct.Reconstructor`).  Every 100 steps and at the end the composite JPGs,
the thresholded voxel grids (.binvox), the raw volumes and texture codes (.npz) are written to `sample_save` with the
reference's file names (:508-520).

Weights: `weight_dir` / `weight_dir_decoder` are folders of `<tensor>.txt.npz` files (tools/model_util.py:26-39).  The
reference does not ship them; when a folder is missing the seeded initialisers are used and the script says so.
"""
import json
import math
import os
import shutil
import sys

import numpy as np


def _read_png(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), np.float32)[None] / 255.0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        raise SystemExit("usage: python Reconstruct_RenderNet_Face.py <config.json> [--max-steps N]")
    with open(argv[0], 'r') as fh:
        cfg = json.load(fh)
    for key in ('target_albedo', 'target_normal', 'target_azimuth_light', 'target_elevation_light', 'batch_size', 'z_dim',
                'inner_step', 'max_epochs', 'shape_eta', 'pose_eta', 'tex_eta', 'light_eta', 'sample_save'):
        if key not in cfg:
            raise KeyError("config is missing %r (see config_reconstruction_RenderNet.json)" % key)
    os.environ.setdefault("HIP_VISIBLE_DEVICES", "{0}".format(cfg.get('gpu', 0)))
    import torch
    from PIL import Image
    from rendernet_amd import reconstruct as RC
    from rendernet_amd.tools import binvox_rw

    sample_save = cfg['sample_save']
    os.makedirs(sample_save, exist_ok=True)
    shutil.copyfile(argv[0], os.path.join(sample_save, 'config.json'))                       # :421
    max_steps = int(argv[argv.index("--max-steps") + 1]) if "--max-steps" in argv else None

    dec_spec = RC.ShapeDecoderSpec(z_dim=int(cfg['z_dim']))
    weight_dict_MLP = weight_dict_decoder = None
    wd, wdd = cfg.get('weight_dir', ''), cfg.get('weight_dir_decoder', '')
    if os.path.isdir(wd) and os.path.isdir(wdd):
        weight_dict_MLP, weight_dict_decoder = RC.load_weights(wd), RC.load_weights(wdd)      # :337-339, by the reference's keys
    else:
        print("weight_dir / weight_dir_decoder not found: using seeded random weights (the reference ships no trained model)")
    ambient_in, k_diffuse, light_col = 0.0, 1.0, (1.0, 1.0, 1.0)                             # :323-325
    rec = RC.Reconstructor(dec_spec=dec_spec, weight_dict_rendernet=weight_dict_MLP, weight_dict_decoder=weight_dict_decoder,
                           batch_size=int(cfg['batch_size']),
                           light_elevation_deg=float(cfg['target_elevation_light']), light_col=light_col,
                           ambient=ambient_in, k_diffuse=k_diffuse, shape_eta=cfg['shape_eta'], pose_eta=cfg['pose_eta'],
                           tex_eta=cfg['tex_eta'], light_eta=cfg['light_eta'])

    target = _read_png(cfg['target_albedo'])                                                 # :430-431
    target_normal = _read_png(cfg['target_normal'])
    res = 4 * rec.tex_spec.new_size
    if target.shape[1:3] != (res, res) or target_normal.shape != target.shape:
        raise SystemExit("targets must be %dx%d RGB images" % (res, res))
    target_compos, target_shading = RC.shaded_target(target, target_normal, float(cfg['target_azimuth_light']),
                                                     float(cfg['target_elevation_light']), light_col, ambient_in, k_diffuse)
    to_u8 = lambda a: np.clip(a * 255., 0, 255).astype(np.uint8)
    Image.fromarray(to_u8(target_compos[0])).save(os.path.join(sample_save, "shaded_target.png"))       # :444-445
    Image.fromarray(to_u8(target_shading[0])).save(os.path.join(sample_save, "shading.png"))

    B, S = rec.B, rec.tex_spec.size
    tgt = torch.as_tensor(np.tile(target_compos, (B, 1, 1, 1)), dtype=torch.float32).cuda()
    state = {"step": 0}

    def dump(loss):
        """:497-520: composites, voxels, texture codes of all hypotheses."""
        with torch.no_grad():
            compos, _, _, shape = rec.forward()
        vals = rec.values()
        img, vox = to_u8(compos.cpu().numpy()), shape.cpu().numpy()
        for i in range(B):
            name = "{0}_{1}_p{2:.1f}_t_{3:.1f}_los_{4:.5f}".format(i, state["step"], int(vals["param"][i][0] * 180 / math.pi),
                                                                    int(90 - vals["param"][i][1] * 180 / math.pi), loss[i])
            Image.fromarray(img[i]).save(os.path.join(sample_save, name + ".jpg"))
            binvox_rw.save_binvox(vox[i].reshape(S, S, S) > cfg.get('threshold', 0.1), os.path.join(sample_save, name + ".binvox"))
            np.savez(os.path.join(sample_save, name + "_Param.txt"), vox[i].reshape(S, S, S))
            np.savez(os.path.join(sample_save, name + "_TEX.txt"), vals["texture"][i])

    # the search of rendernet_amd.reconstruct.reconstruct, unrolled here for the periodic dumps and the step cap
    phi_range, theta_range, best = 60.0, 30.0, None
    for epoch in range(int(cfg['max_epochs'])):
        if epoch == 0:                                                                       # :456-461
            params = RC.create_param_center(B, phi_mid=270, phi_range=phi_range, theta_mid=90, theta_range=theta_range)
            rec.assign(vector=np.full((B, dec_spec.z_dim), 0.5, np.float32), param=params,
                       texture=np.random.randn(B, rec.tex_spec.z_dim).astype(np.float32),
                       light=(np.linspace(230, 320, num=B) * math.pi / 180.0)[:, None])
        else:                                                                                # :463-473
            phi_range /= 2
            theta_range /= 2
            params = RC.create_param_center(B, phi_mid=best["param_deg"][0], phi_range=phi_range,
                                            theta_mid=best["param_deg"][1], theta_range=theta_range)
            rec.assign(vector=np.tile(best["vector"][None], (B, 1)), param=params,
                       texture=np.tile(best["texture"][None], (B, 1)), light=np.tile(best["light"][None], (B, 1)))
        for _ in range(int(cfg['inner_step'])):
            loss = rec.step(tgt).cpu().numpy()
            state["step"] += 1
            print("{0} {1}".format(state["step"], loss))
            if state["step"] % 100 == 0:
                dump(loss)
            if max_steps is not None and state["step"] >= max_steps:
                break
        with torch.no_grad():                                                                # :523-541
            final = rec.recon_loss(rec.forward()[0], tgt).cpu().numpy()
        vals = rec.values()
        i = int(np.argmin(final))
        deg = vals["param"][i] * 180.0 / math.pi
        best = {"vector": vals["vector"][i], "texture": vals["texture"][i], "light": vals["light"][i],
                "param_deg": np.array([deg[0], 90 - deg[1], 1.0])}
        np.savez(os.path.join(sample_save, "{0}_loss_.txt".format(state["step"])), final)
        print("BEST LOSS " + str(i))
        print("BEST PARAM " + str(best["param_deg"]))
        if max_steps is not None and state["step"] >= max_steps:
            dump(final)
            break


if __name__ == "__main__":
    main()
