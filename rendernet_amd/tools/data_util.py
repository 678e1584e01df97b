"""Training-data loaders of the reference (tools/data_util.py:13-157): a tar of rendered images whose file
names carry the pose, and a folder of 64^3 binvox models.  Host-side generators yielding NumPy batches
(images in [0,255], voxels {0,1}, poses in radians) exactly as the reference feeds its placeholders."""
import math
import os

import numpy as np

from . import binvox_rw
from . import utils


def extract_param_from_names(image_path):
    """tools/data_util.py:13-30: '..._p<azimuth>_t<elevation>_r<radius>...' -> [[azimuth, elevation, 3.3/radius]]
    (radians; the file's elevation runs 10..170 from the up axis and is mapped to 80..-80 from the horizon)."""
    ip, it, ir = image_path.find('_p'), image_path.find('_t'), image_path.find('_r')
    if min(ip, it, ir) < 0:
        raise ValueError("no _p/_t/_r pose pattern in %r" % image_path)
    azimuth = float(image_path[ip + 2:it]) * math.pi / 180.0
    elevation = (90.0 - float(image_path[it + 2:ir])) * math.pi / 180.0
    scale = 3.3 / float(image_path[ir + 2:ir + 5])
    return np.array([[azimuth, elevation, scale]])


def model_file_for(img_name, model_path):
    """tools/data_util.py:123-132: which binvox file an image name refers to."""
    parts = img_name.split('_')
    if 'ply' in parts[0]:
        return os.path.join(model_path, parts[0] + ".binvox")
    cand = os.path.join(model_path, "model_chair_%s_clean.binvox" % parts[2])
    if not os.path.exists(cand):
        cand = os.path.join(model_path, "model_normalized_%s_clean.binvox" % parts[2])
    return cand


def _pad_tail(arrays, names, counter, batch_size):
    """tools/data_util.py:144-157: a short tail is repeated up to one batch."""
    reps = int(np.ceil(float(batch_size) / counter))
    out = [np.repeat(a[:counter], reps, axis=0)[:batch_size] for a in arrays]
    return out, list(np.repeat(names[:counter], reps, axis=0)[:batch_size])


def model_loader(cfg, model_path):
    """tools/data_util.py:32-62: (voxels [n,64,64,64,1], names) chunks from a tar of binvox models."""
    chunk = cfg['batch_size'] * cfg['batches_chunk']
    mods = np.zeros((chunk, 64, 64, 64, 1), np.float32)
    names, counter = [], 0
    for ix, (mod, name) in enumerate(utils.NpyTarReader(model_path)):
        mods[ix % chunk] = np.reshape(mod.astype(np.float32), (64, 64, 64, 1))
        names.append(name)
        counter += 1
        if counter == chunk:
            yield mods, names
            mods, names, counter = np.zeros_like(mods), [], 0
    if counter > 0 and counter % cfg['batch_size'] != 0:
        (mods,), names = _pad_tail([mods], names, counter, cfg['batch_size'])
        yield mods, names
    elif counter > 0:
        yield mods[:counter], names


def data_loader(cfg, img_path, model_path, validation_mode=False, flatten=False, img_res=256, add_noise=False):
    """tools/data_util.py:64-157: (images [n,res,res,1|3] in 0..255, voxels [n,64,64,64,1], poses [n,3], names)
    chunks of batch_size*batches_chunk samples (batch_size when validating)."""
    chunk = cfg['batch_size'] if validation_mode else cfg['batch_size'] * cfg['batches_chunk']
    ch = 1 if flatten else 3

    def fresh():
        return (np.zeros((chunk, img_res, img_res, ch), np.float32), np.zeros((chunk, 64, 64, 64, 1), np.float32),
                np.zeros((chunk, 3), np.float32))

    ims, mods, params = fresh()
    names, counter = [], 0
    for item in utils.NpyTarReader(img_path):
        if not isinstance(item, tuple) or item[0] is None or item[1] is None:
            continue
        img, name = item
        idx = counter
        if flatten:
            ims[idx] = np.reshape(np.mean(img, axis=2) if img.ndim == 3 else img, (img_res, img_res, 1))
        else:
            ims[idx] = np.reshape(img[:, :, :3], (img_res, img_res, 3))            # ignore alpha
        if add_noise:
            ims[idx] += np.random.uniform(0.0, 1.0, size=ims[idx].shape)
        params[idx] = extract_param_from_names(name)[0]
        with open(model_file_for(name, model_path), 'rb') as f:
            mods[idx] = np.reshape(binvox_rw.read_as_3d_array(f).data.astype(np.float32), (64, 64, 64, 1))
        names.append(name)
        counter += 1
        if counter == chunk:
            yield ims, mods, params, names
            (ims, mods, params), names, counter = fresh(), [], 0
    if counter > 0:
        if counter % cfg['batch_size'] != 0:
            (ims, mods, params), names = _pad_tail([ims, mods, params], names, counter, cfg['batch_size'])
            yield ims, mods, params, names
        else:
            yield ims[:counter], mods[:counter], params[:counter], names


def _read_texture_code(texture_path, ident):
    """`beta<ident>.mat` (MATLAB file with the 199 Basel-face texture coefficients under 'beta',
    tools/data_util.py:184-186); a plain `beta<ident>.npy` is accepted as well."""
    mat = os.path.join(texture_path, "beta%s.mat" % ident)
    if os.path.exists(mat):
        import scipy.io
        return np.reshape(scipy.io.loadmat(mat)['beta'].astype(np.float32), 199)
    return np.reshape(np.load(os.path.join(texture_path, "beta%s.npy" % ident)).astype(np.float32), 199)


def data_loader_image_texture_normal_face(cfg, img_path, model_path, texture_path, normal_path, validation_mode=False,
                                          img_res=256, add_noise=True):
    """tools/data_util.py:159-233: (images, normals [n,res,res,3] in 0..255, voxels [n,64,64,64,1], texture codes
    [n,199], poses [n,3], names) chunks for the face renderer.  Image names look like `<model>ply<id>_..._p<az>_t<el>_r<rad>`;
    the model file is `<first field>.binvox`, the texture code `beta<id>.mat`, the normal map `<name>.png`."""
    from PIL import Image
    chunk = cfg['batch_size'] if validation_mode else cfg['batch_size'] * cfg['batches_chunk']

    def fresh():
        return (np.zeros((chunk, img_res, img_res, 3), np.float32), np.zeros((chunk, img_res, img_res, 3), np.float32),
                np.zeros((chunk, 64, 64, 64, 1), np.float32), np.zeros((chunk, 199), np.float32), np.zeros((chunk, 3), np.float32))

    ims, nrms, mods, texs, params = fresh()
    names, counter = [], 0
    for item in utils.NpyTarReader(img_path):
        if not isinstance(item, tuple) or item[0] is None or item[1] is None:
            continue
        img, name = item
        idx = counter
        ims[idx] = np.reshape(img.astype(np.float32)[:, :, :3], (img_res, img_res, 3))
        if add_noise:
            ims[idx] += np.random.uniform(0.0, 1.0, size=ims[idx].shape)
        first = name.split('_')[0]
        texs[idx] = _read_texture_code(texture_path, first.split('ly')[1])
        nrms[idx] = np.asarray(Image.open(os.path.join(normal_path, name + ".png")), np.float32)[:, :, :3]
        params[idx] = extract_param_from_names(name)[0]
        with open(os.path.join(model_path, first + ".binvox"), 'rb') as f:
            mods[idx] = np.reshape(binvox_rw.read_as_3d_array(f).data.astype(np.float32), (64, 64, 64, 1))
        names.append(name)
        counter += 1
        if counter == chunk:
            yield ims, nrms, mods, texs, params, names
            (ims, nrms, mods, texs, params), names, counter = fresh(), [], 0
    if counter > 0:
        if counter % cfg['batch_size'] != 0:
            (ims, nrms, mods, texs, params), names = _pad_tail([ims, nrms, mods, texs, params], names, counter, cfg['batch_size'])
            yield ims, nrms, mods, texs, params, names
        else:
            yield ims[:counter], nrms[:counter], mods[:counter], texs[:counter], params[:counter], names
