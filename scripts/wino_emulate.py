"""Lane-level NumPy emulation of csrc/conv_wino.hip (Winograd F(2x2,3x3), 16x16x4 fp32 MFMA).

A development tool, not a product path: it restates the kernel's INDEX arithmetic -- block id -> (m-block,
n-block), the LDS-DMA lane -> (pixel, chunk) map with its swizzle, the fragment-read addresses, the MFMA
operand / accumulator lane maps, the packed filter layout of RN_PACK_CONV_WINO and the epilogue's output
addressing -- so that all of it can be checked against a plain convolution on a CPU-only box before a GPU
minute is spent.  tests/test_wino_layout.py runs it on small ragged shapes.
"""
import numpy as np

WPW, WPH = 34, 18
WNPIX = WPW * WPH
WRAW_PIECES = 40
WRAW_B = WRAW_PIECES * 1024
WU_B = 16 * 4 * 32 * 16

G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float32)


def pack_wino(w_tf, transposed=False):
    """TF conv filter [3,3,Cin,Cout] (or, transposed=True, a conv_transpose-style [3,3,Cout,Cin] read with
    flipped taps) -> U = G g G^T packed [Cout/32][Cin/16][16 xi][4 kq][32 n][4 r], channel c = step*16 + kq*4 + r.
    3-D filters [3,3,3,Cin,Cout]: the depth tap joins the channel, c' = t2*Cin + c (transform over the first two dims)."""
    w = np.asarray(w_tf, np.float32)
    if w.ndim == 5:
        if transposed:
            w = w[::-1, ::-1, ::-1].transpose(0, 1, 2, 4, 3)
        w = np.ascontiguousarray(w).reshape(3, 3, 3 * w.shape[3], w.shape[4])
    elif transposed:
        w = w[::-1, ::-1].transpose(0, 1, 3, 2)
    cin, cout = w.shape[2], w.shape[3]
    assert cin % 16 == 0 and cout % 16 == 0
    NB = 32 if cout % 32 == 0 else 16                                      # n-block width (one or two MFMA n-tiles per wave)
    u = np.einsum("ip,pqcn,jq->ijcn", G, w, G).astype(np.float32)          # [4,4,Cin,Cout]
    u = u.reshape(16, cin // 16, 4, 4, cout // NB, NB)                    # xi, step, kq, r, nb, n
    return np.ascontiguousarray(u.transpose(4, 1, 0, 2, 5, 3)).reshape(-1)


G2 = np.array([[1, 0], [1, 1], [0, 1]], np.float32)          # F(2,2): y0 = m0 + m1, y1 = m1 - m2 with m = (G h) * (B^T d)


def pack_wino4(w_tf, transposed=False):
    """4x4 filters: [4,4,Cin,Cout] (or, transposed=True, the conv_transpose layout [4,4,Cout,Cin] with flipped taps) as FOUR
    2x2 sub-filters h_ab = g[2a:2a+2, 2b:2b+2], each transformed with F(2x2,2x2): U_ab = G2 h_ab G2^T (9 planes).
    Packed [Cout/NB][(Cin/16)*4][9 xi][4 kq][NB n][4 r]: K step s = cstep*4 + (2a+b), channel c = cstep*16 + kq*4 + r;
    NB = 64 when Cout % 64 == 0, else 32."""
    w = np.asarray(w_tf, np.float32)
    if transposed:
        w = w[::-1, ::-1].transpose(0, 1, 3, 2)
    cin, cout = w.shape[2], w.shape[3]
    assert w.shape[:2] == (4, 4) and cin % 16 == 0 and cout % 16 == 0
    NB = 64 if cout % 64 == 0 else 32 if cout % 32 == 0 else 16
    subs = []
    for a in range(2):
        for b in range(2):
            h = w[2 * a:2 * a + 2, 2 * b:2 * b + 2]
            subs.append(np.einsum("ip,pqcn,jq->ijcn", G2, h, G2).reshape(9, cin, cout))
    u = np.stack(subs)                                                     # [sub, xi, Cin, Cout]
    u = u.reshape(4, 9, cin // 16, 4, 4, cout // NB, NB)                   # sub, xi, cstep, kq, r, nb, n
    return np.ascontiguousarray(u.transpose(5, 2, 0, 1, 3, 6, 4)).reshape(-1).astype(np.float32)


def conv_wino4_emulated(x, u_packed, cout, pad_lo=1, bias=None):
    """4x4 stride-1 conv, pad_lo 1 (SAME conv) or 2 (the flipped conv of a stride-1 transposed conv), x [B,H,W,Cin] ->
    [B,H,W,Cout], lane by lane like conv_wino_kernel<.., MODE 1>: K steps run over (channel step, sub-filter); sub-filter
    (a, b) reads the patch shifted by (2a, 2b) pixels and its own filter piece; 3x3 input tiles, 9 accumulators."""
    B, H, W, Cin = x.shape
    xf = np.ascontiguousarray(x, np.float32).reshape(-1)
    y = np.full((B, H, W, cout), np.nan, np.float32)
    bh, bw = (H + 15) // 16, (W + 31) // 32
    NT = 4 if cout % 64 == 0 else 2 if cout % 32 == 0 else 1
    NPIECE = 9 * NT                                                        # 1-KiB filter pieces per step: 9 xi x (4 kq x 16NT n x 16 B = NT KiB)
    UPW = -(-NPIECE // 8)                                                  # per wave (piece p = wave + 8 i; p >= NPIECE: nothing)
    USTEP = NPIECE * 1024
    USTAGE = UPW * 8 * 1024
    mblocks, nblocks, nstep = B * bh * bw, cout // (16 * NT), (Cin // 16) * 4
    lane = np.arange(64)
    l16, kq = lane & 15, lane >> 4
    seen = set()
    for bid in range(mblocks * nblocks):
        mb, nb = block_map(bid, mblocks, nblocks)
        assert (mb, nb) not in seen
        seen.add((mb, nb))
        bx, by, b = mb % bw, (mb // bw) % bh, mb // (bw * bh)
        acc = np.zeros((8, 9, NT, 64, 4), np.float32)
        for s in range(nstep):
            cstep, sub = s >> 2, s & 3
            y0, x0 = by * 16 - pad_lo + 2 * (sub >> 1), bx * 32 - pad_lo + 2 * (sub & 1)
            lds = np.zeros((WRAW_B + USTAGE) // 4, np.float32)
            for wave in range(8):
                for i in range(5):
                    p = wave + 8 * i
                    q = p * 16 + (lane >> 2)
                    py, px = q // WPW, q % WPW
                    iy, ix = y0 + py, x0 + px
                    ok = (q < WNPIX) & (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
                    off = ((b * H + iy) * W + ix) * Cin * 4 + (((lane & 3) ^ ((px >> 1) & 3)) * 16) + cstep * 64
                    for ln in range(64):
                        dst = (p * 1024 + ln * 16) // 4
                        lds[dst:dst + 4] = xf[off[ln] // 4: off[ln] // 4 + 4] if ok[ln] else 0.0
                for i in range(UPW):
                    p = wave + 8 * i
                    if p >= NPIECE:
                        continue                                           # the kernel issues these with an out-of-range offset
                    g = (nb * nstep + s) * USTEP + p * 1024
                    dst = (WRAW_B + p * 1024) // 4
                    lds[dst:dst + 256] = u_packed[g // 4: g // 4 + 256]
            for wave in range(8):
                raddr = [(2 * wave * WPW + 2 * l16) * 64 + ((kq ^ ((l16 + hj) & 3)) << 4) for hj in range(2)]
                uaddr = WRAW_B + kq * 256 * NT + l16 * 16
                d = np.zeros((3, 3, 64, 4), np.float32)
                for ai in range(3):
                    for bi in range(3):
                        ad = (raddr[bi >> 1] + (ai * WPW + bi) * 64) // 4
                        d[ai, bi] = lds[ad[:, None] + np.arange(4)]
                t = np.stack([d[0] - d[1], d[1], d[1] - d[2]])                                # B^T d, B^T = [[1,-1,0],[0,1,0],[0,1,-1]]
                v = np.stack([t[:, 0] - t[:, 1], t[:, 1], t[:, 1] - t[:, 2]], 1)              # (B^T d) B
                for xi in range(9):
                    vv = v[xi // 3, xi % 3]
                    for nt in range(NT):
                        ad = (uaddr + xi * 1024 * NT + nt * 256) // 4
                        bb = lds[ad[:, None] + np.arange(4)]
                        A = np.zeros((16, 16), np.float32)
                        Bm = np.zeros((16, 16), np.float32)
                        A[l16[:, None], (4 * kq)[:, None] + np.arange(4)] = bb
                        Bm[(4 * kq)[:, None] + np.arange(4), l16[:, None]] = vv
                        Dm = A @ Bm
                        acc[wave, xi, nt] += Dm[(4 * kq)[:, None] + np.arange(4), l16[:, None]]
        for wave in range(8):
            ty = wave
            for nt in range(NT):
                for r in range(4):
                    n = nb * 16 * NT + nt * 16 + 4 * kq + r
                    M = acc[wave, :, nt, :, r].reshape(3, 3, 64)
                    sc = np.stack([M[:, 0] + M[:, 1], M[:, 1] - M[:, 2]], 1)                  # A^T = [[1,1,0],[0,1,-1]]
                    Y = np.stack([sc[0] + sc[1], sc[1] - sc[2]])
                    tx = l16
                    for dy in range(2):
                        for dx in range(2):
                            oy, ox = by * 16 + 2 * ty + dy, bx * 32 + 2 * tx + dx
                            if oy >= H:
                                continue
                            ok = ox < W
                            val = Y[dy, dx] + (bias[n] if bias is not None else 0.0)
                            y[b, oy, ox[ok], n[ok]] = val[ok]
    return y


def block_map(bid, mblocks, nblocks):
    T = mblocks * nblocks
    e = bid
    if bid < (T & ~255):
        s = bid >> 3
        e = (s >> 5) * 256 + (bid & 7) * 32 + (s & 31)
    per = 8 * nblocks
    g, full = e // per, mblocks >> 3
    rem, gs, g0 = e - g * per, 8, g
    if g >= full:
        rem, gs, g0 = e - full * per, mblocks - full * 8, full
    return g0 * 8 + rem % gs, rem // gs


def conv_wino_emulated(x, u_packed, cout, bias=None):
    """x [B,H,W,Cin] (2-D) or [B,H,W,D,Cin] (3x3x3 conv) float32 -> y [B,H,W,(D,)Cout]; follows the kernel thread by
    thread (vectorised over lanes)."""
    three_d = x.ndim == 5
    if three_d:
        B, H, W, D, Cin = x.shape
        KD = 3
    else:
        B, H, W, Cin = x.shape
        D, KD = 1, 1
    xf = np.ascontiguousarray(x, np.float32).reshape(-1)
    y = np.full((B, H, W, D, cout), np.nan, np.float32)
    bh, bw = (H + 15) // 16, (W + 31) // 32
    spt = Cin // 16
    NT = 2 if cout % 32 == 0 else 1
    USTEP = 16384 * NT
    mblocks, nblocks, nstep = B * D * bh * bw, cout // (16 * NT), KD * spt
    lane = np.arange(64)
    l16, kq = lane & 15, lane >> 4
    seen = set()
    for bid in range(mblocks * nblocks):
        mb, nb = block_map(bid, mblocks, nblocks)
        assert (mb, nb) not in seen
        seen.add((mb, nb))
        bx, by, dz, b = mb % bw, (mb // bw) % bh, (mb // (bw * bh)) % D, mb // (bw * bh * D)
        y0, x0 = by * 16 - 1, bx * 32 - 1
        s_begin = spt if (KD == 3 and dz == 0) else 0
        s_end = nstep - (spt if (KD == 3 and dz == D - 1) else 0)
        win_off = (dz - (1 if KD == 3 else 0)) * Cin * 4
        acc = np.zeros((8, 16, 2, 64, 4), np.float32)                     # wave, xi, nt, lane, r
        for s in range(s_begin, s_end):
            lds = np.zeros((WRAW_B + WU_B) // 4, np.float32)
            for wave in range(8):
                for i in range(5):                                        # raw patch pieces
                    p = wave + 8 * i
                    q = p * 16 + (lane >> 2)
                    py, px = q // WPW, q % WPW
                    iy, ix = y0 + py, x0 + px
                    ok = (q < WNPIX) & (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
                    off = ((b * H + iy) * W + ix) * D * Cin * 4 + win_off + (((lane & 3) ^ ((px >> 1) & 3)) * 16) + s * 64
                    for ln in range(64):
                        dst = (p * 1024 + ln * 16) // 4
                        lds[dst:dst + 4] = xf[off[ln] // 4: off[ln] // 4 + 4] if ok[ln] else 0.0
                for i in range(2 * NT):                                   # filter pieces
                    g = (nb * nstep + s) * USTEP + wave * 2048 * NT + i * 1024
                    dst = (WRAW_B + (wave * 2 * NT + i) * 1024) // 4
                    lds[dst:dst + 256] = u_packed[g // 4: g // 4 + 256]
            for wave in range(8):
                raddr = [(2 * wave * WPW + 2 * l16) * 64 + ((kq ^ ((l16 + hj) & 3)) << 4) for hj in range(2)]
                uaddr = WRAW_B + kq * 256 * NT + l16 * 16
                d = np.zeros((4, 4, 64, 4), np.float32)
                for ai in range(4):
                    for bi in range(4):
                        ad = (raddr[bi >> 1] + (ai * WPW + bi) * 64) // 4
                        d[ai, bi] = lds[ad[:, None] + np.arange(4)]
                t = np.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]])           # [i][bi]
                v = np.stack([t[:, 0] - t[:, 2], t[:, 1] + t[:, 2], t[:, 2] - t[:, 1], t[:, 1] - t[:, 3]], 1)  # [i][j]
                for xi in range(16):
                    vv = v[xi // 4, xi % 4]                                                   # [lane, s]
                    for nt in range(NT):
                        ad = (uaddr + xi * 1024 * NT + nt * 256) // 4
                        bb = lds[ad[:, None] + np.arange(4)]                                  # [lane, s]
                        # MFMA 16x16x4 x4 with the FILTER as the A operand: A[row=l16 (channel)][k=4kq+s],
                        # B[k=4kq+s][col=l16 (tile)]
                        A = np.zeros((16, 16), np.float32)
                        Bm = np.zeros((16, 16), np.float32)
                        A[l16[:, None], (4 * kq)[:, None] + np.arange(4)] = bb
                        Bm[(4 * kq)[:, None] + np.arange(4), l16[:, None]] = vv
                        Dm = A @ Bm                                                           # [row = channel, col = tile]
                        # D -> lane (col = l16, rows 4kq..4kq+3)
                        acc[wave, xi, nt] += Dm[(4 * kq)[:, None] + np.arange(4), l16[:, None]]
        for wave in range(8):
            ty = wave
            for nt in range(NT):
                for r in range(4):
                    n = nb * 16 * NT + nt * 16 + 4 * kq + r        # a lane holds channels 4kq..4kq+3 of tile tx = l16
                    M = acc[wave, :, nt, :, r].reshape(4, 4, 64)
                    sc = np.stack([M[:, 0] + M[:, 1] + M[:, 2], M[:, 1] - M[:, 2] - M[:, 3]], 1)   # [i][dx][lane]
                    Y = np.stack([sc[0] + sc[1] + sc[2], sc[1] - sc[2] - sc[3]])                   # [dy][dx][lane]
                    tx = l16
                    for dy in range(2):
                        for dx in range(2):
                            oy, ox = by * 16 + 2 * ty + dy, bx * 32 + 2 * tx + dx
                            if oy >= H:
                                continue
                            ok = ox < W
                            val = Y[dy, dx] + (bias[n] if bias is not None else 0.0)
                            y[b, oy, ox[ok], dz, n[ok]] = val[ok]
    assert len(seen) == mblocks * nblocks
    return y if three_d else y[:, :, :, 0]
