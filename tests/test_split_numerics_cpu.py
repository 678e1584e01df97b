"""The arithmetic of the two split operand formats (csrc/conv_wino_bf3.hip), restated in NumPy -- no GPU.

"split"   (bf16x3): x = p0 + p1 + p2 exactly, p_i bf16 (round to nearest even); a product is the six piece products i + j <= 2.
"split16" (fp16x2): x / s = h0 + h1 (+ at most 2^-22 relative), h_i fp16, s a power of two with |x / s| < 2^15 for the whole tensor;
                    a product is h0 h0 + h0 h1 + h1 h0, times the two scales.
With float64 accumulation only the OPERAND REPRESENTATION is left: the tests state that it is exact for bf16x3, 22-bit for fp16x2, and in
both cases below what the fp32 accumulation every route shares (exact MFMA chain or 16-bit MFMA with fp32 accumulate) leaves -- the
argument DESIGN.md section 4 makes for reporting the fp16x2 route; the measured counterpart is tests/test_gpu_wino_robust.py."""
import numpy as np
import pytest


def _bf16(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def split_bf3(x):
    p0 = _bf16(x)
    r1 = (x - p0).astype(np.float32)
    p1 = _bf16(r1)
    p2 = _bf16((r1 - p1).astype(np.float32))
    return p0, p1, p2


def h2_scale(amax, bound):
    """rn: h2_scale() -- the smallest power of two s with bound * amax / s <= 2^15."""
    t = np.float32(amax) * np.float32(bound) / np.float32(32768.0)
    if not t > 0:
        return 1.0
    m, e = np.frexp(t)
    return float(np.ldexp(1.0, e - 1 if m == 0.5 else e))


def split_h2(x, s):
    y = (x / np.float32(s)).astype(np.float32)                # exact: s is a power of two
    h0 = y.astype(np.float16)
    h1 = (y - h0.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h0, h1


CASES = {
    "N(0,1)": lambda r, m, k: r.standard_normal((m, k)),
    "abs + 30": lambda r, m, k: np.abs(r.standard_normal((m, k))) + 30,
    "log-normal gains": lambda r, m, k: np.abs(r.standard_normal((m, k))) * np.exp(1.5 * r.standard_normal(k)),
    "spikes x100": lambda r, m, k: 1 + 100 * np.abs(r.standard_normal((m, k))) * (r.random((m, k)) < 0.02),
    "six decades": lambda r, m, k: r.standard_normal((m, k)) * np.where(r.random((m, k)) < 0.001, 1e4, 1e-2),
}


def test_bf16x3_pieces_sum_exactly():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(100000) * np.exp(4 * rng.standard_normal(100000))).astype(np.float32)
    p0, p1, p2 = split_bf3(x)
    assert np.array_equal((p0.astype(np.float64) + p1 + p2).astype(np.float32), x)
    assert np.all(np.abs(x - p0) <= 2.0 ** -8 * np.abs(x)) and np.all(np.abs(x - p0 - p1) <= 2.0 ** -16 * np.abs(x))


def test_fp16x2_scale_bounds_and_precision():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(200000) * np.exp(3 * rng.standard_normal(200000))).astype(np.float32)
    for bound in (4.0, 49.0, 225.0):
        s = h2_scale(np.abs(x).max(), bound)
        assert np.log2(s) == np.round(np.log2(s))                                    # a power of two
        assert bound * np.abs(x).max() / s <= 32768.0 < 2 * bound * np.abs(x).max() / s + 1e-30 or s == 1.0
        h0, h1 = split_h2(x, s)
        assert np.all(np.isfinite(h0)) and np.all(np.isfinite(h1))
        rec = (h0.astype(np.float64) + h1.astype(np.float64)) * s
        big = np.abs(x) / s >= 2.0 ** -3                                             # both pieces normal: 22 mantissa bits
        assert np.all(np.abs(rec - x)[big] <= 2.0 ** -22 * np.abs(x)[big])
        assert np.all(np.abs(rec - x) <= np.maximum(2.0 ** -22 * np.abs(x), 2.0 ** -25 * s))     # below: absolute, <= half an fp16 subnormal step
    assert h2_scale(0.0, 225.0) == 1.0


@pytest.mark.parametrize("name", list(CASES))
def test_representation_error_is_below_the_shared_accumulation_error(name):
    """K = 1024 dot products: what the operand formats lose (float64 accumulation) against what an fp32 accumulation loses."""
    rng = np.random.default_rng(sum(map(ord, name)))
    M, K, N = 96, 1024, 32
    V = CASES[name](rng, M, K).astype(np.float32)
    U = (rng.uniform(-1, 1, (K, N)) * 0.02).astype(np.float32)
    ref = V.astype(np.float64) @ U.astype(np.float64)
    ymax = np.abs(ref).max()
    # fp32 accumulation in the order of a 16-wide MFMA chain: 16 products summed exactly, then one fp32 rounding per K step
    acc = np.zeros((M, N), np.float32)
    for k0 in range(0, K, 16):
        acc = (acc.astype(np.float64) + V[:, k0:k0 + 16].astype(np.float64) @ U[k0:k0 + 16].astype(np.float64)).astype(np.float32)
    e_acc = np.abs(acc - ref).max() / ymax
    vp, up = split_bf3(V), split_bf3(U)
    m6 = sum(vp[i].astype(np.float64) @ up[j].astype(np.float64) for i in range(3) for j in range(3) if i + j <= 2)
    e_b3 = np.abs(m6 - ref).max() / ymax
    errs = []
    for head in (1.0, 225.0):                                                        # the scale's bound may overshoot the true maximum by the transform's growth
        sv, su = h2_scale(np.abs(V).max(), head), h2_scale(np.abs(U).max(), 1.0)
        (v0, v1), (u0, u1) = split_h2(V, sv), split_h2(U, su)
        v0, v1, u0, u1 = (a.astype(np.float64) for a in (v0, v1, u0, u1))
        errs.append(np.abs((v0 @ u0 + v0 @ u1 + v1 @ u0) * sv * su - ref).max() / ymax)
    print("%-18s fp32 accumulation %.1e   bf16x3 %.1e   fp16x2 %.1e / %.1e (tight / 225x headroom)" % (name, e_acc, e_b3, errs[0], errs[1]))
    assert e_b3 <= 5e-8 and max(errs) <= 4e-7
    assert e_b3 < e_acc and max(errs) < e_acc * 2.0      # the representation never dominates (fp16x2: at most on par in the mildest case)
