"""Streaming TAR reader/writer of the reference's training data (tools/utils.py:22-114): the image sets are
plain tar files of PNG/JPG renders named `<model>_p<azimuth>_t<elevation>_r<radius>.png`, voxel sets are
`.binvox` members, arrays are `.npy` / zlib'd `.npy.z` members under `data/`.  Host-side (NumPy + PIL;
`scipy.misc.imread` of the reference no longer exists)."""
import io
import tarfile
import time
import zlib

import numpy as np

from . import binvox_rw

PREFIX = 'data/'
SUFFIX = '.npy.z'


class NpyTarWriter(object):
    """tools/utils.py:25-46: arrays as zlib-compressed .npy members."""

    def __init__(self, fname):
        self.tfile = tarfile.open(fname, 'w|')

    def add(self, arr, name):
        raw = io.BytesIO()
        np.save(raw, arr)
        payload = zlib.compress(raw.getvalue())
        info = tarfile.TarInfo(PREFIX + name + SUFFIX)
        info.size = len(payload)
        info.mtime = time.time()
        self.tfile.addfile(info, io.BytesIO(payload))

    def add_bytes(self, payload, member_name):
        """Raw member (a PNG / binvox file as stored in the reference's image and model tars)."""
        info = tarfile.TarInfo(member_name)
        info.size = len(payload)
        info.mtime = time.time()
        self.tfile.addfile(info, io.BytesIO(payload))

    def close(self):
        self.tfile.close()


def _model_stem(name):
    parts = name.split('_')
    return parts[0] if 'ply' in parts[0] else '_'.join(parts[:3]) + '_clean'


class NpyTarReader(object):
    """tools/utils.py:49-114: iterate a tar as a stream; yields, by member extension,
    .npy/.npy.z -> array; .binvox -> (bool voxels, model name); .png/.jpg -> (float32 image, member stem);
    anything else -> None.  Unreadable images yield (None, None) like the reference."""

    def __init__(self, fname):
        self.tfile = tarfile.open(fname, 'r|')

    def __iter__(self):
        return self

    def __next__(self):
        entry = self.tfile.next()
        while entry is not None and not entry.isfile():
            entry = self.tfile.next()
        if entry is None:
            self.close()
            raise StopIteration()
        data = self.tfile.extractfile(entry).read()
        ext = entry.name.split('.')
        if ext[-1].lower() == 'z':
            data = zlib.decompress(data)
            ext.pop()
        kind = ext[-1].lower()
        if kind == 'npy':
            return np.load(io.BytesIO(data))
        base = entry.name.rsplit('/', 1)[-1]
        if kind == 'binvox':
            vox = binvox_rw.read_as_3d_array(io.BytesIO(data))
            return vox.data, _model_stem(base.split('.')[0])
        if kind in ('jpg', 'jpeg', 'png'):
            try:
                from PIL import Image
                img = np.asarray(Image.open(io.BytesIO(data))).astype(np.float32)
            except (OSError, RuntimeError, TypeError, ValueError):
                return None, None
            return img, entry.name[:-(len(ext[-1]) + 1)].rsplit('/', 1)[-1]
        return None

    next = __next__

    def close(self):
        self.tfile.close()
