"""binvox reader / writer: the input surface of the renderer and the output of the inverse-rendering loop (reference
tools/binvox_rw.py:45-93, :175-239, after Daniel Maturana's binvox-rw-py).  Host-side NumPy; numpy-2 safe."""
import numpy as np


class Voxels(object):
    """Dense binvox model: `data` bool [d0,d1,d2]; dims/translate/scale are file metadata."""

    def __init__(self, data, dims, translate, scale, axis_order):
        assert axis_order in ('xzy', 'xyz')
        self.data, self.dims, self.translate, self.scale, self.axis_order = data, dims, translate, scale, axis_order

    def write(self, fp):
        write(self, fp)


def read_header(fp):
    line = fp.readline().strip()
    if not line.startswith(b'#binvox'):
        raise IOError('Not a binvox file')
    dims = [int(v) for v in fp.readline().strip().split(b' ')[1:]]
    translate = [float(v) for v in fp.readline().strip().split(b' ')[1:]]
    scale = [float(v) for v in fp.readline().strip().split(b' ')[1:]][0]
    fp.readline()
    return dims, translate, scale


def read_as_3d_array(fp, fix_coords=True):
    """Run-length decode (value,count byte pairs) to a dense bool array; fix_coords swaps xzy->xyz."""
    dims, translate, scale = read_header(fp)
    raw = np.frombuffer(fp.read(), dtype=np.uint8)
    if raw.size % 2:
        raise IOError('truncated binvox payload')
    values, counts = raw[::2], raw[1::2]
    data = np.repeat(values, counts).astype(bool)
    if data.size != int(np.prod(dims)):
        raise IOError('binvox payload decodes to %d voxels, header says %s' % (data.size, dims))
    data = data.reshape(dims)
    if fix_coords:
        return Voxels(np.transpose(data, (0, 2, 1)), dims, translate, scale, 'xyz')
    return Voxels(data, dims, translate, scale, 'xzy')


def write(voxel_model, fp):
    """tools/binvox_rw.py:175-226, dense models.  Same bytes as the reference's per-voxel state machine: runs are cut
    at 255, and a run whose length is a multiple of 255 is followed by a (value, 0) pair when the value switches
    (the reference resets its counter to 0 at 255 and dumps it again at the switch, :212-220)."""
    data = np.asarray(voxel_model.data)
    if data.ndim != 3:
        raise ValueError('only dense 3-D models are supported')
    fp.write(('#binvox 1\n' + 'dim ' + ' '.join(map(str, voxel_model.dims)) + '\n' +
              'translate ' + ' '.join(map(str, voxel_model.translate)) + '\n' +
              'scale ' + str(voxel_model.scale) + '\n' + 'data\n').encode('latin-1'))
    if voxel_model.axis_order not in ('xzy', 'xyz'):
        raise ValueError('Unsupported voxel model axis order')
    flat = (data if voxel_model.axis_order == 'xzy' else np.transpose(data, (0, 2, 1))).reshape(-1).astype(np.uint8)
    if flat.size == 0:
        return
    starts = np.flatnonzero(np.concatenate(([True], flat[1:] != flat[:-1])))
    lengths = np.diff(np.concatenate((starts, [flat.size])))
    values = flat[starts]
    full, rem = lengths // 255, lengths % 255
    tail = np.ones(lengths.size, np.int64)               # every run ends with its remainder pair (possibly count 0) ...
    if rem[-1] == 0:
        tail[-1] = 0                                     # ... except that the final flush skips an empty counter
    per_run = full + tail
    out_vals = np.repeat(values, per_run)
    out_cnts = np.full(out_vals.size, 255, np.uint8)
    ends = np.cumsum(per_run) - 1
    has_tail = tail.astype(bool)
    out_cnts[ends[has_tail]] = rem[has_tail].astype(np.uint8)
    fp.write(np.stack((out_vals, out_cnts), axis=1).tobytes())


def save_binvox(data, fname):
    """tools/binvox_rw.py:228-239: 3-D boolean array -> file, axis order 'xyz', unit scale."""
    data = np.asarray(data)
    model = Voxels(data, data.shape, [0.0, 0.0, 0.0], 1.0, 'xyz')
    with open(fname, 'wb') as f:
        write(model, f)
