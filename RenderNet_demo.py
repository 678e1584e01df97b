#!/usr/bin/env python
"""RenderNet Phong shading demo on the MI355X path -- same command line as the reference's
RenderNet_demo.py (flags :72-108, pose convention :33-38, output naming :60-64).

The network outputs a normal map from a 3D voxel grid; the normal map is then Phong-shaded with
lighting control.  Differences from the reference, all forced by what the reference ships:
  * no frozen TF graph exists (./model/3d2d_renderer.pb is a Google-Drive link, README.md:185-189):
    weights come from --weights (an .npz keyed by the TF variable names) or, by default, from the
    seeded reference initialisers -- the demo then exercises the full path but renders noise;
  * --voxel_path defaults to ./voxel/Misc/bunny.binvox as in the reference and falls back to the
    shipped ./binvox/bunny.binvox when that path does not exist (reference defect);
  * --rotate renders the 72 poses as batches instead of 72 batch-1 session runs.
"""
import argparse
import math
import os

import numpy as np

# Phong shading parameters (RenderNet_demo.py:17-20)
AMBIENT_IN = (0.1)
K_DIFFUSE = .9
LIGHT_COL = np.array([[1., 1., 1.]])


def compute_pose_param(azimuth, elevation, radius):
    """RenderNet_demo.py:33-38."""
    phi = azimuth * math.pi / 180.0
    theta = (90 - elevation) * math.pi / 180
    param = np.array([phi, theta, 3.3 / radius])
    return np.expand_dims(param, axis=0)


def _str2bool(v):
    # the reference declares `type=bool` (:103-104): any non-empty string is True
    return bool(v)


def build_parser():
    fmt_cls = argparse.ArgumentDefaultsHelpFormatter
    parser = argparse.ArgumentParser(formatter_class=fmt_cls)
    parser.add_argument('--voxel_path', type=str, default="./voxel/Misc/bunny.binvox", help="Path to the input voxel.")
    parser.add_argument('--azimuth', type=float, default=250, help="Value of azimuth, between (0,360)")
    parser.add_argument('--elevation', type=float, default=60, help="Value of elevation, between (0,360)")
    parser.add_argument('--light_azimuth', type=float, default=250, help="Value of azimuth for light, between (0,360)")
    parser.add_argument('--light_elevation', type=float, default=60, help="Value of elevation for light, between (0,360)")
    parser.add_argument('--radius', type=float, default=3.3, help="Value of radius, between (2.5, 4.5)")
    parser.add_argument('--render_dir', type=str, default='./render', help='Path to the rendered images.')
    parser.add_argument('--rotate', type=_str2bool, default=False,
                        help='Flag rotate and render an object by 360 degree in azimuth. Overwrites early settings in azimuth.')
    # additions (not in the reference)
    parser.add_argument('--weights', type=str, default=None,
                        help='.npz of weights keyed by TF variable names, or the reference\'s frozen graph `*.pb` '
                             '(its Const nodes are read without TensorFlow); default: seeded random initialisation')
    parser.add_argument('--batch', type=int, default=24, help='poses rendered per launch with --rotate')
    parser.add_argument('--gemm', choices=['f32', 'split', 'split16'], default=None,
                        help='multiply stage of the wide convs: default = the library default (`split`: every fp32 operand as three '
                             'bf16 pieces that sum exactly to it, six piece products, fp32 accumulation -- fp32-class error on the '
                             '16x faster bf16 matrix pipe; include/rendernet_hip.h, rn_conv2d_winograd_split_fwd; env RN_WINO_GEMM); '
                             '`f32` = exact-fp32 MFMA everywhere; `split16` = two fp16 pieces of value / tensor scale (22-bit operands)')
    parser.add_argument('--no_winograd_check', action='store_true',
                        help='weights given with --weights are checked once against F(4x4,3x3) where the net would take F(6x6,3x3) '
                             '(Renderer.validate_winograd, tolerance 2e-4 of max|y| per layer); this flag skips the check')
    parser.add_argument('--gif', type=str, default=None,
                        help='with --rotate: also write the 72 frames as an animated GIF to this path (the reference ships '
                             'such turntables under images/*.gif, README.md)')
    return parser


def save_path_for(render_dir, count, model_name, azimuth, elevation, radius, light_azimuth, light_elevation):
    """RenderNet_demo.py:60-64."""
    return os.path.join(render_dir, str(count).zfill(3) + "_" + model_name + "_pose_%f_%f_%f_light_%f_%f.png" %
                        (azimuth, elevation, radius, light_azimuth, light_elevation))


def render(azimuths, elevation, radius, renderer, voxel, light_dir, render_dir, count0, light_azimuth,
           light_elevation, model_name):
    """RenderNet_demo.py:41-66 for a batch of azimuths."""
    from PIL import Image
    from rendernet_amd.tools import Phong_shading
    params = np.concatenate([compute_pose_param(a, elevation, radius) for a in azimuths], 0)
    vox = np.repeat(voxel, len(azimuths), axis=0)
    normals = renderer.render(vox, params)                              # HIP tensor [B,512,512,3]
    img_phong = Phong_shading.np_phong_composite(normals, light_dir, LIGHT_COL, AMBIENT_IN, K_DIFFUSE).cpu().numpy()
    paths = []
    for i, a in enumerate(azimuths):
        image_out = np.clip(255. * img_phong[i], 0, 255).astype(np.uint8)
        p = save_path_for(render_dir, count0 + i, model_name, a, elevation, radius, light_azimuth, light_elevation)
        print(p)
        Image.fromarray(image_out).save(p)
        paths.append(p)
    return paths


def main(argv=None):
    args = build_parser().parse_args(argv)
    from rendernet_amd.shader import Renderer, ShaderSpec, init_shader_weights
    from rendernet_amd.tools import binvox_rw, Phong_shading

    spec = ShaderSpec(out_ch=3).check()                       # the demo graph has the 3-channel normal-map head
    if args.weights and args.weights.endswith(".pb"):
        # the reference's frozen graph (RenderNet_demo.py:23-30, :111 `./model/3d2d_renderer.pb`, made by
        # demo/RenderNet_converter.py): its variables are Const nodes under their TF names
        from rendernet_amd.tools.graphdef import load_frozen_weights
        from rendernet_amd.shader import shader_variable_shapes
        weights = load_frozen_weights(args.weights, {name: tuple(shape) for name, shape, _ in shader_variable_shapes(spec)})
    elif args.weights:
        weights = dict(np.load(args.weights))
    else:
        print("no --weights given: using seeded random weights (the reference ships no trained model)")
        weights = init_shader_weights(spec, seed=1234)
    renderer = Renderer(spec, weights, gemm=args.gemm)      # None: the library default (env RN_WINO_GEMM, "split")

    if not os.path.exists(args.render_dir):
        os.makedirs(args.render_dir)
    light_dir = Phong_shading.generate_light_pos(args.light_elevation, args.light_azimuth)
    voxel_path = args.voxel_path
    if not os.path.exists(voxel_path) and voxel_path == "./voxel/Misc/bunny.binvox":
        voxel_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "binvox", "bunny.binvox")
    with open(voxel_path, 'rb') as f:
        voxel = np.reshape(binvox_rw.read_as_3d_array(f).data.astype(np.float32), (1, 64, 64, 64, 1))
    model_name = os.path.basename(voxel_path).split('.binvox')[0]
    if args.weights and not args.no_winograd_check:
        # trained weights have statistics nobody has looked at: one render with the F(6x6,3x3) rounding guard on, on the object
        # and a pose of this very run; filters whose F(6x6,3x3) output strays from F(4x4,3x3) are demoted for the session
        demoted = renderer.validate_winograd(voxel, compute_pose_param(args.azimuth, args.elevation, args.radius), tol=2e-4)
        print("winograd check (tolerance 2e-4 of max|conv|): %s" % ("all F(6x6,3x3) layers kept" if not demoted else
              "%d layer(s) demoted to F(4x4,3x3): %s" % (len(demoted), demoted)))

    if args.rotate:
        # RenderNet_demo.py:130-137: 72 poses, 5 degrees apart, numbered 000..071 -- rendered `--batch` poses per launch
        az = list(np.arange(0.0, 360.0, 5.0))
        paths = []
        for s in range(0, len(az), args.batch):
            paths += render(az[s:s + args.batch], args.elevation, args.radius, renderer, voxel, light_dir, args.render_dir, s,
                            args.light_azimuth, args.light_elevation, model_name)
        if args.gif:
            from PIL import Image
            frames = [Image.open(p).convert("P", palette=Image.ADAPTIVE) for p in paths]
            frames[0].save(args.gif, save_all=True, append_images=frames[1:], duration=60, loop=0)
            print(args.gif)
    else:
        render([args.azimuth], args.elevation, args.radius, renderer, voxel, light_dir, args.render_dir, 0,
               args.light_azimuth, args.light_elevation, model_name)


if __name__ == "__main__":
    main()
