#!/usr/bin/env python
"""`python RenderNet_Shader.py <config.json>` -- the reference's Phong-shader script
(RenderNet_Shader.py) on the MI355X path, same JSON keys (config_RenderNet.json:1-17,
README.md:41-70).

Round-1 scope: the forward render path.  The script builds the graph of RenderNet_Shader.py:135-156
(resample -> transform -> crop -> RenderNet), loads weights from `<sample_save>/<trained_model_name>.npz`
when present (else the seeded reference initialisers), renders every binvox under `model_path` at the
poses of the demo sweep and writes PNGs to `sample_save`.  The training loop (:193-306: Adam step,
BCE loss, tar data loader) needs the backward kernels and is not built yet: it raises.
"""
import glob
import json
import os
import sys

import numpy as np


def load_config(path):
    with open(path, 'r') as fh:
        cfg = json.load(fh)
    for key in ('model_path', 'sample_save', 'trained_model_name', 'is_greyscale', 'batch_size', 'keep_prob'):
        if key not in cfg:
            raise KeyError("config is missing %r (see config_RenderNet.json)" % key)
    return cfg


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        raise SystemExit("usage: python RenderNet_Shader.py <config.json> [--train]")
    cfg = load_config(argv[0])
    if "--train" in argv:
        raise NotImplementedError("the training step (RenderNet_Shader.py:193-306) needs the backward kernels, "
                                  "which are not built yet; this script runs the forward/validation render only")
    os.environ.setdefault("HIP_VISIBLE_DEVICES", "{0}".format(cfg.get('gpu', 0)))
    from PIL import Image
    from rendernet_amd.shader import Renderer, ShaderSpec, init_shader_weights
    from rendernet_amd.tools import binvox_rw

    grey = cfg['is_greyscale'].lower() == "true"                       # RenderNet_Shader.py:125
    spec = ShaderSpec(out_ch=1 if grey else 3).check()
    sample_save = cfg['sample_save']
    os.makedirs(sample_save, exist_ok=True)
    wpath = os.path.join(sample_save, cfg['trained_model_name'] + ".npz")
    weights = dict(np.load(wpath)) if os.path.exists(wpath) else init_shader_weights(spec, seed=1234)
    renderer = Renderer(spec, weights)

    files = sorted(glob.glob(os.path.join(cfg['model_path'], "*.binvox")))
    if not files:
        raise SystemExit("no .binvox files under model_path=%s" % cfg['model_path'])
    bs = int(cfg['batch_size'])
    for s in range(0, len(files), bs):
        chunk = files[s:s + bs]
        vox = []
        for p in chunk:
            with open(p, 'rb') as f:
                vox.append(binvox_rw.read_as_3d_array(f).data.astype(np.float32)[..., None])
        vox = np.stack(vox)
        poses = np.tile(np.array([[250 * np.pi / 180, 30 * np.pi / 180, 1.0]], np.float32), (len(chunk), 1))
        out = renderer.run("encoder/output:0", {"real_model_in:0": vox, "view_name:0": poses, "patch_size:0": 128,
                                                "is_training:0": False})
        for p, img in zip(chunk, out):
            name = os.path.basename(p).split('.binvox')[0]
            arr = np.clip(255 * img, 0, 255).astype(np.uint8)
            Image.fromarray(np.squeeze(arr)).save(os.path.join(sample_save, "VALID_%s_pred.png" % name))
            print("rendered", name, arr.shape)


if __name__ == "__main__":
    main()
