"""Read the weights out of a frozen TensorFlow graph (`*.pb`) without TensorFlow or protobuf.

The reference's demo loads `./model/3d2d_renderer.pb` with `tf.import_graph_def` (RenderNet_demo.py:23-30, :111), a graph in
which `demo/RenderNet_converter.py:7-18` (`graph_util.convert_variables_to_constants`) has turned every variable into a `Const`
node that keeps the variable's name (`encoder/e_conv1/e_conv1/weights`, ... -- the names rendernet_amd.variables uses).  This
module walks the protobuf wire format of that file -- GraphDef { repeated NodeDef node = 1 }, NodeDef { name = 1, op = 2,
input = 3, device = 4, map<string, AttrValue> attr = 5 }, AttrValue { tensor = 8 }, TensorProto { dtype = 1, tensor_shape = 2,
tensor_content = 4, float_val = 5, double_val = 6, int_val = 7, int64_val = 10 } -- and returns {node name: ndarray} for the
`Const` nodes.  Host-side plumbing; nothing here touches the GPU.
"""
import struct

import numpy as np

# TensorFlow DataType enum values (types.proto) -> numpy dtype, for the types a frozen RenderNet graph holds
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_}


class GraphDefError(ValueError):
    pass


def _varint(buf, pos):
    result, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise GraphDefError("truncated varint at byte %d" % pos)
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise GraphDefError("varint longer than 64 bits at byte %d" % pos)


def fields(buf):
    """Yield (field number, wire type, value) of one message; value is an int (varint, fixed32/64 raw bits) or a memoryview
    (length-delimited)."""
    buf = memoryview(buf)
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            if pos + 8 > n:
                raise GraphDefError("truncated fixed64 field %d" % num)
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise GraphDefError("field %d claims %d bytes, %d left" % (num, ln, n - pos))
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            if pos + 4 > n:
                raise GraphDefError("truncated fixed32 field %d" % num)
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise GraphDefError("unsupported wire type %d (field %d)" % (wt, num))
        yield num, wt, val


def _shape(buf):
    dims = []
    for num, wt, val in fields(buf):                       # TensorShapeProto { repeated Dim dim = 2; bool unknown_rank = 3 }
        if num == 2 and wt == 2:
            size = 0
            for n2, w2, v2 in fields(val):                 # Dim { int64 size = 1; string name = 2 }
                if n2 == 1 and w2 == 0:
                    size = v2 - (1 << 64) if v2 >> 63 else v2
            dims.append(size)
    return dims


MAX_TENSOR_ELEMENTS = 1 << 31          # no tensor of the path comes near this (the largest filter: 9.4 M); a cap on untrusted input


def _repeated(val, wt, fmt, size):
    """A repeated scalar field arrives packed (one length-delimited blob) or one element per key."""
    if wt == 2:
        if len(val) % size != 0:
            raise GraphDefError("packed field of %d bytes is not a whole number of %d-byte items" % (len(val), size))
        return list(struct.unpack("<%d%s" % (len(val) // size, fmt), bytes(val)))
    return [struct.unpack("<" + fmt, struct.pack("<Q" if size == 8 else "<I", val))[0]]


def _tensor(buf):
    dtype, shape, content = None, [], None
    fvals, dvals, ivals = [], [], []
    for num, wt, val in fields(buf):
        if num == 1 and wt == 0:
            dtype = val
        elif num == 2 and wt == 2:
            shape = _shape(val)
        elif num == 4 and wt == 2:
            content = bytes(val)
        elif num == 5:
            fvals += _repeated(val, wt, "f", 4)
        elif num == 6:
            dvals += _repeated(val, wt, "d", 8)
        elif num in (7, 10, 11):                           # int_val / int64_val / bool_val: varints, packed or not
            if wt == 2:
                pos, mv = 0, memoryview(val)
                while pos < len(mv):
                    v, pos = _varint(mv, pos)
                    ivals.append(v - (1 << 64) if v >> 63 else v)
            else:
                ivals.append(val - (1 << 64) if val >> 63 else val)
    np_dtype = _DTYPES.get(dtype)
    if np_dtype is None:
        return None                                         # strings, resources, ...: not weights
    if any(d < 0 for d in shape):
        raise GraphDefError("tensor with an unknown / negative dimension: %s" % (shape,))
    count = 1
    for d in shape:
        count *= int(d)
        if count > MAX_TENSOR_ELEMENTS:
            raise GraphDefError("tensor shape %s exceeds %d elements" % (shape, MAX_TENSOR_ELEMENTS))
    if content is not None:
        item = np.dtype(np_dtype).itemsize
        if len(content) % item != 0:
            raise GraphDefError("tensor_content of %d bytes is not a whole number of %d-byte items" % (len(content), item))
        arr = np.frombuffer(content, dtype=np.dtype(np_dtype).newbyteorder("<")).astype(np_dtype)
    else:
        vals = fvals if np_dtype == np.float32 else dvals if np_dtype == np.float64 else ivals
        arr = np.asarray(vals, dtype=np_dtype)
        if arr.size == 1 and count > 1:                     # TensorProto's "splat": one value stands for the whole tensor
            arr = np.full(count, arr[0], dtype=np_dtype)
        elif arr.size == 0:
            arr = np.zeros(count, dtype=np_dtype)
    if arr.size != count:
        raise GraphDefError("tensor holds %d values for shape %s" % (arr.size, shape))
    return arr.reshape(shape)


def read_graphdef_constants(path_or_bytes, float_only=True):
    """{node name: ndarray} of the `Const` nodes of a serialized GraphDef (a frozen `.pb`).  float_only drops the integer /
    boolean constants (shapes, axes, strides) a graph is full of."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) else open(path_or_bytes, "rb").read()
    out = {}
    for num, wt, node in fields(data):
        if num != 1 or wt != 2:                             # GraphDef.node; versions / library are skipped
            continue
        name, op, tensor = None, None, None
        for n2, w2, v2 in fields(node):
            if n2 == 1 and w2 == 2:
                name = bytes(v2).decode("utf-8")
            elif n2 == 2 and w2 == 2:
                op = bytes(v2).decode("utf-8")
            elif n2 == 5 and w2 == 2:                       # one map entry: key = 1, value = 2 (AttrValue)
                key, attr = None, None
                for n3, w3, v3 in fields(v2):
                    if n3 == 1 and w3 == 2:
                        key = bytes(v3).decode("utf-8")
                    elif n3 == 2 and w3 == 2:
                        attr = v3
                if key == "value" and attr is not None:
                    for n4, w4, v4 in fields(attr):
                        if n4 == 8 and w4 == 2:             # AttrValue.tensor
                            tensor = v4
        if op == "Const" and name is not None and tensor is not None:
            arr = _tensor(tensor)
            if arr is not None and (not float_only or arr.dtype in (np.float32, np.float64)):
                out[name] = arr.astype(np.float32) if arr.dtype == np.float64 else arr
    return out


def load_frozen_weights(path_or_bytes, expected=None):
    """The variables of a frozen RenderNet graph as {TF variable name: float32 ndarray}: the float `Const` nodes, restricted to
    `expected` (an iterable of variable names, e.g. the keys of init_shader_weights(spec)) when given -- a frozen graph also
    holds float constants that are not variables (the 1e-6 of the loss, dropout keep probabilities, ...).  Raises when an
    expected variable is missing or has another shape than `expected[name]` (when `expected` maps names to arrays)."""
    consts = read_graphdef_constants(path_or_bytes, float_only=True)
    if expected is None:
        return consts
    # `expected`: {name: array} | {name: shape tuple} | iterable of names -- shapes are enough, nobody needs to materialise
    # 237 M random weights to ask for names (rendernet_amd.shader.shader_variable_shapes gives them)
    names = list(expected)
    missing = [n for n in names if n not in consts]
    if missing:
        raise GraphDefError("the graph lacks %d of %d variables (first: %s); it holds %d float constants"
                            % (len(missing), len(names), missing[0], len(consts)))
    out = {}
    for n in names:
        a = consts[n]
        want = None
        if hasattr(expected, "keys"):
            want = getattr(expected[n], "shape", None)
            if want is None and isinstance(expected[n], (tuple, list)):
                want = tuple(expected[n])
        if want is not None and tuple(a.shape) != tuple(want):
            raise GraphDefError("variable %s has shape %s in the graph, the net expects %s" % (n, tuple(a.shape), tuple(want)))
        out[n] = np.ascontiguousarray(a, dtype=np.float32)
    return out
