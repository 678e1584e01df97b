#!/usr/bin/env python
"""`python RenderNet_Shader.py <config.json>` -- the reference's Phong-shader script
(RenderNet_Shader.py) on the MI355X path, same JSON keys (config_RenderNet.json:1-17,
README.md:41-70).

Without `--train` the script builds the graph of RenderNet_Shader.py:135-156 (resample -> transform -> crop ->
RenderNet), loads weights from `<sample_save>/<trained_model_name>.npz` when present (else the seeded reference
initialisers), renders every binvox under `model_path` at the demo pose and writes PNGs to `sample_save`.

`--train` runs the reference's training loop (:193-306): epochs over the image tar (`image_path`, poses parsed from
the member names) + binvox folder (`model_path`), crop 32 for the first five epochs then 64 (:204-207), BCE (greyscale)
or MSE loss, Adam(beta1=0.5) with staircase-decayed learning rate, a sample PNG every 600 steps, weights saved as
`<sample_save>/<trained_model_name>.npz` after every epoch and every `checkpoint_secs` (weights + Adam moments +
global_step + epoch, written atomically; a restart resumes from it like the reference's Supervisor), then the
validation pass over `image_path_valid` (:257-301, dropout off).  Under `torch.distributed.run` every rank trains on its shard of each batch and the gradients are
summed with bucketed RCCL all-reduces (rendernet_amd/train.py).
"""
import glob
import json
import os
import sys
import time

import numpy as np


def load_config(path):
    with open(path, 'r') as fh:
        cfg = json.load(fh)
    for key in ('model_path', 'sample_save', 'trained_model_name', 'is_greyscale', 'batch_size', 'keep_prob'):
        if key not in cfg:
            raise KeyError("config is missing %r (see config_RenderNet.json)" % key)
    return cfg


def _save_png(path, arr01):
    from PIL import Image
    Image.fromarray(np.squeeze(np.clip(255 * arr01, 0, 255).astype(np.uint8))).save(path)


def train(cfg, argv):
    """RenderNet_Shader.py:193-306 on the MI355X training step."""
    import random
    import torch
    import torch.distributed as dist
    from rendernet_amd.shader import ShaderSpec, init_shader_weights
    from rendernet_amd.train import Trainer
    from rendernet_amd.parallel import shard_range
    from rendernet_amd.tools.data_util import data_loader

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(cfg.get('gpu', 0)) if world == 1 else "0"))
    bs = int(cfg['batch_size'])
    if bs % world != 0:
        # an empty or short shard would leave its rank out of the bucket / loss all-reduces: rank 0 would block for ever
        raise SystemExit("batch_size %d is not a multiple of the %d ranks: every rank needs the same, non-empty shard of "
                         "each batch" % (bs, world))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # generous collective timeout: rank 0 validates alone at the end of an epoch while the others wait at a barrier
        dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(hours=6))
    grey = cfg['is_greyscale'].lower() == "true"
    spec = ShaderSpec(out_ch=1 if grey else 3).check()
    sample_save = cfg['sample_save']
    os.makedirs(sample_save, exist_ok=True)
    wpath = os.path.join(sample_save, cfg['trained_model_name'] + ".npz")
    tr = Trainer(spec, init_shader_weights(spec, seed=1234), device="cuda:%d" % local_rank, e_eta=cfg.get('e_eta', 1e-5),
                 decay_steps=cfg.get('decay_steps', 100000), keep_prob=cfg.get('keep_prob', 1.0))
    # resume like the reference's Supervisor (:171-185): weights, Adam moments, global_step and the epoch counter
    first_epoch = tr.load_checkpoint(dict(np.load(wpath))) if os.path.exists(wpath) else 0
    new_res = spec.new_size
    max_steps = int(argv[argv.index("--max-steps") + 1]) if "--max-steps" in argv else None
    ckpt_secs = float(cfg.get('checkpoint_secs', 7200))
    last_ckpt = time.time()
    l1_all = [float(v) for v in np.ravel(tr.checkpoint_extra.get("l1_all", []))]     # the validation history survives a restart
    for epoch in range(first_epoch, int(cfg['max_epochs'])):
        patch = new_res // 4 if epoch < 5 else new_res // 2                           # :204-207
        for images, models, params, names in data_loader(cfg, img_path=cfg['image_path'], model_path=cfg['model_path'],
                                                         flatten=grey, validation_mode=False, img_res=4 * new_res):
            images = images / 255.0                                                     # :224
            for idx in range(len(images) // bs):
                sl = slice(idx * bs, (idx + 1) * bs)
                lo, hi = shard_range(bs, rank, world)                                   # this rank's frames of the batch
                # one crop window per batch, the same on every rank (tools/model_util.py:92)
                start = torch.randint(0, new_res - patch + 1, (2,), device="cuda")
                if world > 1:
                    dist.broadcast(start, src=0)
                loss = tr.step(models[sl][lo:hi], params[sl][lo:hi], images[sl][lo:hi], patch_size=patch,
                               start_point=start.tolist(), global_batch=bs)
                step = tr.global_step
                if rank == 0:
                    print("Step {0} Loss {1}".format(step, float(loss.item())))
                if rank == 0 and time.time() - last_ckpt >= ckpt_secs:                  # Supervisor(save_model_secs=checkpoint_secs)
                    tr.save_checkpoint(wpath, epoch, {"l1_all": l1_all})
                    last_ckpt = time.time()
                if step % 600 == 0 and rank == 0:                                       # :242-253
                    with torch.no_grad():
                        pred, (r, c, p, _) = tr.forward(models[sl][lo:hi], params[sl][lo:hi], patch, start.tolist())
                    i = random.randint(0, hi - lo - 1)
                    tgt = images[sl][lo:hi][i, 4 * r:4 * (r + p), 4 * c:4 * (c + p)]
                    _save_png(os.path.join(sample_save, "{0}_train_target_{1}_patch.png".format(names[sl][lo + i], step)), tgt)
                    _save_png(os.path.join(sample_save, "{0}_train_{1}_patch.png".format(names[sl][lo + i], step)),
                              pred[i].detach().cpu().numpy())
                if max_steps is not None and step >= max_steps:
                    break
            if max_steps is not None and tr.global_step >= max_steps:
                break
        if rank == 0:
            tr.save_checkpoint(wpath, epoch + 1, {"l1_all": l1_all})                                        # :257 sess_saver.save (atomic)
            last_ckpt = time.time()
        # validation (:258-301): full-resolution render with is_training False (dropout off), mean absolute error; on
        # rank 0 while the other ranks wait at the barrier below (a generous timeout: torch's default is 10 min for nccl)
        if rank == 0 and cfg.get('image_path_valid') and os.path.exists(cfg['image_path_valid']):
            l1, cnt = 0.0, 0
            with torch.no_grad():
                for images, models, params, names in data_loader(cfg, img_path=cfg['image_path_valid'],
                                                                 model_path=cfg['model_path'], flatten=grey,
                                                                 validation_mode=True, img_res=4 * new_res):
                    images = images / 255.0
                    pred, _ = tr.forward(models, params, is_training=False)
                    pred = pred.cpu().numpy()
                    if cnt % 600 == 0:
                        _save_png(os.path.join(sample_save, "VALID_{0}_target_{1}.png".format(names[0], epoch)), images[0])
                        _save_png(os.path.join(sample_save, "VALID_{0}_pred_{1}.png".format(names[0], epoch)), pred[0])
                    acc = float(np.mean(np.absolute(images - pred)))
                    print("Validation accuracy {0}".format(acc))
                    l1 += acc
                    cnt += 1
            if cnt:
                l1_all.append(l1 / cnt)
                np.savez(os.path.join(sample_save, "L1 All.txt"), l1_all)
        if world > 1:
            dist.barrier()                      # nobody starts the next epoch's collectives while rank 0 validates
        if max_steps is not None and tr.global_step >= max_steps:
            break
    if world > 1:
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        raise SystemExit("usage: python RenderNet_Shader.py <config.json> [--train [--max-steps N]]")
    cfg = load_config(argv[0])
    if "--train" in argv:
        return train(cfg, argv)
    os.environ.setdefault("HIP_VISIBLE_DEVICES", "{0}".format(cfg.get('gpu', 0)))
    from PIL import Image
    from rendernet_amd.shader import Renderer, ShaderSpec, init_shader_weights
    from rendernet_amd.tools import binvox_rw

    grey = cfg['is_greyscale'].lower() == "true"                       # RenderNet_Shader.py:125
    spec = ShaderSpec(out_ch=1 if grey else 3).check()
    sample_save = cfg['sample_save']
    os.makedirs(sample_save, exist_ok=True)
    wpath = os.path.join(sample_save, cfg['trained_model_name'] + ".npz")
    if os.path.exists(wpath):
        weights = {k: v for k, v in np.load(wpath).items() if not k.startswith("__")}     # drop the optimiser state
    else:
        weights = init_shader_weights(spec, seed=1234)
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
