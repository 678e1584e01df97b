"""ORACLE (test infrastructure only): NumPy restatement of rn_dropout's mask -- tf.nn.dropout
(`x / keep_prob * floor(keep_prob + U[0,1))`, the op behind RenderNet_Shader.py:39,43,47,88,103,107-123 via
tools/layer_util.py:124-131) with the uniforms of Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random
numbers: as easy as 1, 2, 3", SC'11) keyed as include/rendernet_hip.h states: element e takes word e%4 of
philox(counter = (e/4 lo, e/4 hi, stream lo, stream hi), key = (seed lo, seed hi)); u = (word >> 8) * 2^-24.
TensorFlow's own generator cannot be reproduced (its stream depends on graph-level op seeds); what is pinned to the
reference is the formula, checked statistically in tests/."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over uint64 arrays holding 32-bit words."""
    c0, c1, c2, c3 = (np.asarray(v, np.uint64) & MASK for v in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & MASK, p1 & MASK, \
                         ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def uniforms(n, seed, stream):
    q = np.arange((n + 3) // 4, dtype=np.uint64)
    w = philox4x32_10(q & MASK, q >> np.uint64(32), np.uint64(stream & 0xFFFFFFFF), np.uint64((stream >> 32) & 0xFFFFFFFF),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(w, 1).reshape(-1)[:n]
    return ((words >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def dropout(x, keep_prob, seed, stream):
    x = np.asarray(x, np.float32)
    u = uniforms(x.size, seed, stream).reshape(x.shape)
    kp = np.float32(keep_prob)
    return (x * (np.float32(1.0) / kp) * np.floor(kp + u)).astype(np.float32)
