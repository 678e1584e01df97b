"""NumPy restatement of tools/binvox_rw.py:45-93 (reader), tools/Phong_shading.py:138-228, 247-253
(NumPy Phong composite) and RenderNet_demo.py:33-38 (pose parameters).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import math
import numpy as np


def read_binvox(path):
    """tools/binvox_rw.py:45-93 with fix_coords=True: header, RLE (value,count) byte pairs,
    reshape to dims, xzy -> xyz transpose.  np.bool (:85) is restated as bool (numpy 2)."""
    with open(path, 'rb') as fp:
        line = fp.readline().strip()
        if not line.startswith(b'#binvox'):
            raise IOError('Not a binvox file')
        dims = list(map(int, fp.readline().strip().split(b' ')[1:]))
        fp.readline()   # translate
        fp.readline()   # scale
        fp.readline()   # 'data'
        raw = np.frombuffer(fp.read(), dtype=np.uint8)
    values, counts = raw[::2], raw[1::2]
    data = np.repeat(values, counts).astype(bool).reshape(dims)
    return np.transpose(data, (0, 2, 1))


def write_binvox_bytes(data, dims, translate=(0.0, 0.0, 0.0), scale=1.0, axis_order='xyz'):
    """tools/binvox_rw.py:175-226 as the per-voxel state machine it is (pure-Python loop: small cases only)."""
    out = bytearray()
    out += ('#binvox 1\n' + 'dim ' + ' '.join(map(str, dims)) + '\n' + 'translate ' + ' '.join(map(str, translate)) + '\n' +
            'scale ' + str(scale) + '\n' + 'data\n').encode('latin-1')
    flat = (data.flatten() if axis_order == 'xzy' else np.transpose(data, (0, 2, 1)).flatten()).astype(np.uint8)
    state, ctr = int(flat[0]), 0
    for c in flat:
        c = int(c)
        if c == state:
            ctr += 1
            if ctr == 255:                      # :212-215
                out += bytes((state, ctr))
                ctr = 0
        else:                                   # :216-220
            out += bytes((state, ctr))
            state, ctr = c, 1
    if ctr > 0:                                 # :222-224
        out += bytes((state, ctr))
    return bytes(out)


def compute_pose_param(azimuth, elevation, radius):
    """RenderNet_demo.py:33-38 (== tools/data_util.py:111-118)."""
    phi = azimuth * math.pi / 180.0
    theta = (90 - elevation) * math.pi / 180
    return np.array([[phi, theta, 3.3 / radius]])


def generate_light_pos(elevation=90, azimuth=90):
    """tools/Phong_shading.py:247-253."""
    e = np.array([[elevation]]) * math.pi / 180.0
    a = np.array([[azimuth]]) * math.pi / 180.0
    return np.hstack((-np.sin(e) * np.cos(a), np.cos(e), -np.sin(e) * np.sin(a)))


def np_mask(images_in):
    """tools/Phong_shading.py:138-148."""
    m = np.linalg.norm(images_in, axis=3, keepdims=True)
    return 1. / (1. + np.exp(-(255. * m - 150)))


def np_phong_shading(img_batch, light_dir, light_col, k_diffuse):
    """tools/Phong_shading.py:162-200."""
    n = (img_batch - 0.5).reshape([-1, 3])
    n = n / np.linalg.norm(n, axis=1)[:, None]
    light_dir = light_dir / np.linalg.norm(light_dir, axis=1).reshape([-1, 1])
    npix = int(np.prod(img_batch.shape[1:3]))
    ld = np.repeat(light_dir, npix, 0)
    lc = np.repeat(light_col, npix, 0)
    d = np.maximum(np.sum(n * ld, axis=1, keepdims=True), 0.0)
    d = k_diffuse * (np.repeat(d, 3, 1) * lc)
    return np.clip(d.reshape(img_batch.shape), 0, 1)


def np_mask_white(images_in):
    """tools/Phong_shading.py:150-160."""
    m = np.linalg.norm(1. - images_in, axis=3, keepdims=True)
    return 1. / (1. + np.exp(-(255. * m - 80)))


def np_phong_composite(images_in, light_dir, light_col, ambient_in, k_diffuse, background_col="Black", with_mask=True):
    """tools/Phong_shading.py:202-228."""
    diffuse = np_phong_shading(images_in, light_dir, light_col, k_diffuse)
    if with_mask:
        mask = np_mask(images_in) if background_col.lower() == "black" else np_mask_white(images_in)
        compos = mask * (ambient_in + diffuse) + (1 - mask)
    else:
        compos = ambient_in + diffuse
    return np.clip(compos, 0, 1)
