#!/usr/bin/env python
"""`python RenderNet_Texture_Face_Normal.py <config.json>` -- the reference's texture + normal face renderer
(RenderNet_Texture_Face_Normal.py; BASELINE configs[2]) on the MI355X path, same JSON keys
(config_RenderNet_texture.json:1-18): image_path, image_path_valid, normal_path, texture_path, model_path, gpu,
batch_size, max_epochs, batches_chunk, e_eta, keep_prob, decay_steps, trained_model_name, sample_save,
checkpoint_secs.

`--train` runs the reference's loop (:196-334): epochs over the image tar (poses, model and texture ids parsed from
the member names), crop new_res/4 for the first four epochs then new_res/2 (:226-229), loss = MSE(image) +
MSE(normal) (:182-183), Adam(beta1 = 0.5) with the staircase learning rate, sample PNGs every 600 steps
(`<name>_train_target_<step>_patch.png`, `..._target_normal_...`, `<name>_train_<step>_patch.png`,
`..._patch_normal.png`, :263-278), a checkpoint at the end of every epoch and every `checkpoint_secs`, then the
validation pass (:286-331: full-resolution render with dropout off, `VALID_<name>_{target,target_normal,pred,
pred_normal}_<epoch>.png`, mean absolute error appended to `L1 All.txt`).
Without `--train` it renders every `<id>.binvox` of `model_path` that has a texture code in `texture_path` at the
demo pose and writes `VALID_<id>_pred.png` / `VALID_<id>_pred_normal.png`.
Under `torch.distributed.run` the batch is sharded over the ranks (gradient all-reduce: rendernet_amd/train.py).
"""
import glob
import json
import os
import random
import shutil
import sys
import time

import numpy as np


def _save_png(path, arr01):
    from PIL import Image
    Image.fromarray(np.squeeze(np.clip(255 * arr01, 0, 255).astype(np.uint8))).save(path)


def load_config(path):
    with open(path, 'r') as fh:
        cfg = json.load(fh)
    for key in ('model_path', 'texture_path', 'sample_save', 'trained_model_name', 'batch_size', 'keep_prob'):
        if key not in cfg:
            raise KeyError("config is missing %r (see config_RenderNet_texture.json)" % key)
    return cfg


def train(cfg, argv):
    import torch
    import torch.distributed as dist
    from rendernet_amd.parallel import shard_range
    from rendernet_amd.texture import TextureSpec, init_texture_weights
    from rendernet_amd.tools.data_util import data_loader_image_texture_normal_face
    from rendernet_amd.train import TextureTrainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(cfg.get('gpu', 0)) if world == 1 else "0"))
    bs = int(cfg['batch_size'])
    if bs % world != 0:
        raise SystemExit("batch_size %d is not a multiple of the %d ranks: every rank needs the same, non-empty shard "
                         "(an empty shard would leave its rank out of the gradient all-reduce)" % (bs, world))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # generous collective timeout: rank 0 validates alone at the end of an epoch while the others wait at a barrier
        dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(hours=6))
    spec = TextureSpec().check()
    sample_save = cfg['sample_save']
    os.makedirs(sample_save, exist_ok=True)
    if rank == 0:
        shutil.copyfile(argv[0], os.path.join(sample_save, 'config.json'))               # :218
    wpath = os.path.join(sample_save, cfg['trained_model_name'] + ".npz")
    tr = TextureTrainer(spec, init_texture_weights(spec, seed=1234), device="cuda:%d" % local_rank,
                        e_eta=cfg.get('e_eta', 1e-5), decay_steps=cfg.get('decay_steps', 100000), keep_prob=cfg.get('keep_prob', 1.0))
    first_epoch = tr.load_checkpoint(dict(np.load(wpath))) if os.path.exists(wpath) else 0
    new_res = spec.new_size
    max_steps = int(argv[argv.index("--max-steps") + 1]) if "--max-steps" in argv else None
    ckpt_secs = float(cfg.get('checkpoint_secs', 7200))
    last_ckpt = time.time()
    l1_all = [float(v) for v in np.ravel(tr.checkpoint_extra.get("l1_all", []))]     # the validation history survives a restart
    lo, hi = shard_range(bs, rank, world)
    for epoch in range(first_epoch, int(cfg['max_epochs'])):
        patch = new_res // 4 if epoch < 4 else new_res // 2                            # :226-229
        loader = data_loader_image_texture_normal_face(cfg, img_path=cfg['image_path'], model_path=cfg['model_path'],
                                                       normal_path=cfg['normal_path'], texture_path=cfg['texture_path'],
                                                       validation_mode=False, img_res=4 * new_res)
        for images, normals, models, textures, params, names in loader:
            images, normals = images / 255.0, normals / 255.0                           # :245-246
            for idx in range(len(images) // bs):
                sl = slice(idx * bs + lo, idx * bs + hi)
                start = torch.randint(0, new_res - patch + 1, (2,), device="cuda")      # one window per batch, all ranks
                if world > 1:
                    dist.broadcast(start, src=0)
                loss = tr.step(models[sl], textures[sl], params[sl], images[sl], normals[sl], patch_size=patch,
                               start_point=start.tolist(), global_batch=bs)
                step = tr.global_step
                if rank == 0:
                    print("Step {0} Loss {1}".format(step, float(loss.item())))
                if step % 600 == 0 and rank == 0:                                      # :263-278
                    with torch.no_grad():
                        img, nrm, (r, c, p, _) = tr.forward(models[sl], textures[sl], params[sl], patch, start.tolist())
                    i = random.randint(0, hi - lo - 1)
                    nm = names[idx * bs + lo + i]
                    win = (slice(4 * r, 4 * (r + p)), slice(4 * c, 4 * (c + p)))
                    _save_png(os.path.join(sample_save, "{0}_train_target_{1}_patch.png".format(nm, step)), images[sl][i][win])
                    _save_png(os.path.join(sample_save, "{0}_train_target_normal_{1}_patch.png".format(nm, step)), normals[sl][i][win])
                    _save_png(os.path.join(sample_save, "{0}_train_{1}_patch.png".format(nm, step)), img[i].cpu().numpy())
                    _save_png(os.path.join(sample_save, "{0}_train_{1}_patch_normal.png".format(nm, step)), nrm[i].cpu().numpy())
                if rank == 0 and time.time() - last_ckpt >= ckpt_secs:                  # Supervisor(save_model_secs=checkpoint_secs)
                    tr.save_checkpoint(wpath, epoch, {"l1_all": l1_all})
                    last_ckpt = time.time()
                if max_steps is not None and step >= max_steps:
                    break
            if max_steps is not None and tr.global_step >= max_steps:
                break
        if rank == 0:
            tr.save_checkpoint(wpath, epoch + 1, {"l1_all": l1_all})                                        # :285 sess_saver.save
            last_ckpt = time.time()
        # validation (:287-331), on rank 0 while the others wait at the barrier below
        if rank == 0 and cfg.get('image_path_valid') and os.path.exists(cfg['image_path_valid']):
            l1, cnt = 0.0, 0
            loader = data_loader_image_texture_normal_face(cfg, img_path=cfg['image_path_valid'], model_path=cfg['model_path'],
                                                           normal_path=cfg['normal_path'], texture_path=cfg['texture_path'],
                                                           validation_mode=True, img_res=4 * new_res, add_noise=False)
            with torch.no_grad():
                for images, normals, models, textures, params, names in loader:
                    images, normals = images / 255.0, normals / 255.0
                    img, nrm, _ = tr.forward(models, textures, params, is_training=False)
                    img, nrm = img.cpu().numpy(), nrm.cpu().numpy()
                    if cnt % 600 == 0:
                        i = random.randint(0, len(names) - 1)
                        _save_png(os.path.join(sample_save, "VALID_{0}_target_{1}.png".format(names[i], epoch)), images[i])
                        _save_png(os.path.join(sample_save, "VALID_{0}_target_normal_{1}.png".format(names[i], epoch)), normals[i])
                        _save_png(os.path.join(sample_save, "VALID_{0}_pred_{1}.png".format(names[i], epoch)), img[i])
                        _save_png(os.path.join(sample_save, "VALID_{0}_pred_normal_{1}.png".format(names[i], epoch)), nrm[i])
                    l1 += float(np.mean(np.absolute(images - img)))
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
        raise SystemExit("usage: python RenderNet_Texture_Face_Normal.py <config.json> [--train [--max-steps N]]")
    cfg = load_config(argv[0])
    if "--train" in argv:
        return train(cfg, argv)
    os.environ.setdefault("HIP_VISIBLE_DEVICES", "{0}".format(cfg.get('gpu', 0)))
    from rendernet_amd.texture import TextureRenderer, TextureSpec, init_texture_weights
    from rendernet_amd.tools import binvox_rw
    from rendernet_amd.tools.data_util import _read_texture_code

    spec = TextureSpec().check()
    sample_save = cfg['sample_save']
    os.makedirs(sample_save, exist_ok=True)
    wpath = os.path.join(sample_save, cfg['trained_model_name'] + ".npz")
    if os.path.exists(wpath):
        weights = {k: v for k, v in np.load(wpath).items() if not k.startswith("__")}
    else:
        weights = init_texture_weights(spec, seed=1234)
    renderer = TextureRenderer(spec, weights)
    files = sorted(glob.glob(os.path.join(cfg['model_path'], "*.binvox")))
    if not files:
        raise SystemExit("no .binvox files under model_path=%s" % cfg['model_path'])
    bs = int(cfg['batch_size'])
    for s in range(0, len(files), bs):
        chunk = files[s:s + bs]
        vox, tex, names = [], [], []
        for p in chunk:
            name = os.path.basename(p).split('.binvox')[0]
            try:
                code = _read_texture_code(cfg['texture_path'], name.split('ly')[1] if 'ly' in name else name)
            except (OSError, IndexError):
                code = np.zeros(199, np.float32)
            with open(p, 'rb') as f:
                vox.append(binvox_rw.read_as_3d_array(f).data.astype(np.float32)[..., None])
            tex.append(code)
            names.append(name)
        poses = np.tile(np.array([[250 * np.pi / 180, 30 * np.pi / 180, 1.0]], np.float32), (len(chunk), 1))
        img, nrm = renderer.render(np.stack(vox), np.stack(tex), poses)
        for name, a, b in zip(names, img.cpu().numpy(), nrm.cpu().numpy()):
            _save_png(os.path.join(sample_save, "VALID_%s_pred.png" % name), a)
            _save_png(os.path.join(sample_save, "VALID_%s_pred_normal.png" % name), b)
            print("rendered", name, a.shape)


if __name__ == "__main__":
    main()
