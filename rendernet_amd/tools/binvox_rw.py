"""binvox reader: the input surface of the renderer (reference tools/binvox_rw.py:45-93,
after Daniel Maturana's binvox-rw-py).  Host-side NumPy; numpy-2 safe."""
import numpy as np


class Voxels(object):
    """Dense binvox model: `data` bool [d0,d1,d2]; dims/translate/scale are file metadata."""

    def __init__(self, data, dims, translate, scale, axis_order):
        assert axis_order in ('xzy', 'xyz')
        self.data, self.dims, self.translate, self.scale, self.axis_order = data, dims, translate, scale, axis_order


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
