// The 3x3x3, 32 -> 32 channel convs of the 3-D encoder (res_block_3d / res1_skip: tools/layer_util.py:60-73, RenderNet_Shader.py:51-64;
// 21 launches per render, and their input gradients) on the bf16 matrix pipe at fp32 accuracy -- the fused counterpart of
// conv_wino_bf3.hip for a layer that is too narrow for the three-launch path (32 channels: V / M round trips would cost more
// than the multiplies they save).
//
// Algorithm: as conv_wino.hip -- in channels-last [B,H,W,D,C] a depth slice of the output is a 2-D conv over (H,W) with 3*C
// input channels (the three depth taps), done as Winograd F(2x2,3x3) over (H,W): Y = A^T [ sum_{dz,c} U[xi][dz][c][n] .* V[xi][dz][c] ] A,
// 16 xi planes, V = B^T d B of the 4x4 input tile.  What changes is the multiply: every fp32 V and U value is the exact sum of
// three bf16 pieces and a product is the six piece products with i + j <= 2 on v_mfma_f32_16x16x32_bf16, fp32 accumulate
// (conv_wino_bf3.hip explains the arithmetic; same error class as an fp32 FMA chain).
//
// Design for gfx950 -- what is different from the fp32 kernel and why:
//   * K is only 96 and N only 32, so per MFMA the FILTER fragments are as large as the data fragments: streaming U through
//     LDS (295 KB per item in split form) would saturate the L2 -> LDS path.  U therefore lives in REGISTERS for the whole
//     persistent kernel: a workgroup is 4 waves, ONE per SIMD (512 registers each), wave r owns the four xi of row r of the
//     4x4 xi grid for every tile of the block -- 4 xi x 3 taps x 3 pieces x 2 channel tiles = 72 fragments = 288 VGPRs, read once.
//   * a wave computes exactly the part of the input transform it needs: row r of B^T d (one signed sum of two pixel rows),
//     then the four column combinations; the split into bf16 pieces happens in registers right behind it (v_cvt_pk_bf16_f32 +
//     one subtraction per piece) and is issued in the shadow of the MFMAs of the previous xi: bf16 MFMA and VALU are separate
//     pipes (unlike fp32 MFMA, which conv_wino.hip found to share the vector datapath).
//   * the output transform needs all 16 xi of a (tile, channel): the column half (M[r][*] A) is wave-local, the row half goes
//     through a 32-KiB LDS exchange, after which wave w owns (tile row w / 2, channel tile w % 2) and runs the epilogue on
//     16-byte pixels exactly like conv_wino.hip.
//   * block = 16 x 2 tiles (32 x 4 outputs) of one depth slice, raw patch 34 x 6 pixels x 32 channels = 26 KiB per depth tap,
//     global -> LDS by DMA (8 pixels x 128 B per wave instruction, SAME padding = out-of-range offsets = hardware zero fill),
//     three stages (stage = tap), fetched two steps ahead with counted vmcnt.  A depth tap outside the volume is fetched as
//     zeros and multiplied like the others (2 % of the taps; keeps every item at three identical steps).
//   * persistent grid, XCD k takes 32 consecutive items = the 16 x 2 blocks of one (image, depth slice): halos and the two
//     shared depth taps of neighbouring slices stay in that XCD's L2.
#include "rn_common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int C3 = 32;                          // channels in and out
constexpr int C3PW = 34, C3PH = 4;              // raw patch of one input depth slice: 34 x 4 pixels (one row of 16 tiles + halo)
constexpr int C3NPIX = C3PW * C3PH;             // 136
constexpr int C3NPIECE = 17;                    // DMA pieces of 8 pixels x 128 B
constexpr int C3STAGE = 18 * 1024;              // bytes per patch stage (17 pieces, padded)
constexpr int C3NSTG = 3;
constexpr int C3XCH = 4 * 2 * 2 * 1024;         // one exchange buffer: c[i][q][channel tile] x 1 KiB = 16 KiB (two of them)
constexpr unsigned C3OOB = 0x80000000u;
}

struct C3Args {
    const float* x; const char* u; const float* bias; const float* alpha; const float* res; float* y; float* z;
    unsigned x_bytes;
    int B, H, W, D;
    int bh, bw;                 // blocks per image along H (2 output rows each) and W (32 output columns each)
    int nitems;                 // B * bh * bw: an item is a row of 16 tiles through ALL depth slices
    int act;
    int probe;                  // RN_C3_PROBE (timing experiments, wrong results): 1 no DMA in the loop, 2 no compute, 4 no epilogue, 8 no barriers
};

// filter transform U = G g G^T over (k1, k2) per depth tap (double), rounded to fp32, split into three bf16 pieces and stored
// in MFMA A-fragment order: [xi row i][xi column j][depth tap][piece][channel tile nt][lane = (n % 16) + 16 (c / 8)][c % 8],
// n = output channel, c = input channel.  transposed: the filter of the input gradient (taps flipped in all three axes,
// channel roles swapped) -- the same TF tensor read the other way round.
__global__ __launch_bounds__(256)
void conv3d_wino_pack_bf3_kernel(const float* __restrict__ w_tf, unsigned short* __restrict__ us, int transposed)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;            // (xi, dz, c, n)
    if (idx >= 16 * 3 * C3 * C3) return;
    const int n = idx % C3, c = (idx / C3) % C3, dz = (idx / (C3 * C3)) % 3, xi = idx / (C3 * C3 * 3);
    const int i = xi >> 2, j = xi & 3;
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    double acc = 0.0;
    for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) {
            const float w = transposed ? w_tf[((((size_t)(2 - p) * 3 + (2 - q)) * 3 + (2 - dz)) * C3 + n) * C3 + c]
                                       : w_tf[((((size_t)p * 3 + q) * 3 + dz) * C3 + c) * C3 + n];
            acc = __builtin_fma(G[i][p] * G[j][q], (double)w, acc);
        }
    const float u = (float)acc;
    const __bf16 h0 = (__bf16)u;
    const float r1 = u - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    const float r2 = r1 - (float)h1;
    const __bf16 h2 = (__bf16)r2;
    const __bf16 h[3] = {h0, h1, h2};
    const int lane = (n & 15) + 16 * (c >> 3), nt = n >> 4, e = c & 7;
    for (int p = 0; p < 3; ++p)
        us[((((((size_t)i * 4 + j) * 3 + dz) * 3 + p) * 2 + nt) * 64 + lane) * 8 + e] = __builtin_bit_cast(unsigned short, h[p]);
}

// 8 fp32 values (two registers quads) -> three bf16x8 pieces, round to nearest even, remainders exact
__device__ __forceinline__ void c3_split8(const f32x4 lo, const f32x4 hi, bf16x8 (&p)[3])
{
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        f32x2 v = e < 4 ? f32x2{lo[e], lo[e + 1]} : f32x2{hi[e - 4], hi[e - 3]};
        const bf16x2 h0 = __builtin_convertvector(v, bf16x2);
        v -= __builtin_convertvector(h0, f32x2);
        const bf16x2 h1 = __builtin_convertvector(v, bf16x2);
        v -= __builtin_convertvector(h1, f32x2);
        const bf16x2 h2 = __builtin_convertvector(v, bf16x2);
        p[0][e] = h0[0]; p[0][e + 1] = h0[1];
        p[1][e] = h1[0]; p[1][e + 1] = h1[1];
        p[2][e] = h2[0]; p[2][e + 1] = h2[1];
    }
}

// MFMA with the filter fragment pinned to the ACCUMULATOR half of the register file ("a") or to the vector half ("v"): the 72
// resident fragments are 288 registers, more than either half (256) holds, and hipcc, left to itself, shuffles them between
// the halves around every use (v_accvgpr_read / _mov: 2.3 per MFMA, plus 170 spilled registers).  With the classes fixed at
// the use -- 64 fragments in AGPRs, 8 in VGPRs, accumulators and data fragments in VGPRs -- nothing moves.  The compiler
// inserts no hazard padding around inline asm: the accumulators rotate over six independent registers (a dependent MFMA
// is five instructions away) and the flush that reads them with vector instructions waits explicitly.
__device__ __forceinline__ void c3_mfma_a(f32x4& c, const bf16x8& u, const bf16x8& p)
{
    asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(u), "v"(p));
}
__device__ __forceinline__ void c3_mfma_v(f32x4& c, const bf16x8& u, const bf16x8& p)
{
    asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(u), "v"(p));
}

#define C3_WAIT_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct C3Item { int bx, by, b; unsigned roff[5]; unsigned okmask; };

// 256 threads = 4 waves, ONE per SIMD (512 registers each).  Wave r owns the four xi of row r of the 4x4 xi grid:
// 4 xi x 3 taps x 3 pieces x 2 channel tiles = 72 filter fragments = 288 registers for the whole kernel.
// Depth run: an item is one row of 16 tiles (32 x 2 outputs) of one image, walked through ALL depth slices.  Step d fetches
// input slice d (two steps ahead), transforms and splits it ONCE and feeds it to the three output slices it belongs to -- tap 0
// of slice d + 1, tap 1 of slice d, tap 2 of slice d - 1 -- i.e. three accumulator sets of 4 xi x 2 channel tiles = 96 registers
// (the ring index is the output slice modulo 3, compile-time in the three-fold unrolled loop).  That is 144 MFMAs for about
// 300 vector instructions of transform and splitting per wave and step: two per MFMA, which fit in the issue shadow of a
// 16-cycle MFMA, so one wave per SIMD keeps the matrix pipe busy (re-transforming a slice for each of its three taps, as the
// fp32 kernel does, would be six per MFMA: vector-bound).  After step d output slice d - 1 is complete and is flushed: the
// column half of A^T M A is wave-local, the row half goes through a 16-KiB LDS exchange.  One workgroup barrier per step.
__global__ __launch_bounds__(256, 1)
void conv3d_wino_bf3_kernel(const C3Args a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [patch stage 0 | 1 | 2][exchange 0 | 1]
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);

    // ---- this wave's filter fragments [j][tap][piece][channel tile], xi = (wave, j)
    bf16x8 U[4][3][3][2];
    {
        const bf16x8* up = reinterpret_cast<const bf16x8*>(a.u) + (size_t)wave * (4 * 3 * 3 * 2 * 64) + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int dz = 0; dz < 3; ++dz)
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) U[j][dz][p][nt] = up[(((j * 3 + dz) * 3 + p) * 2 + nt) * 64];
    }
    // ---- row `wave` of B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]: t = d[rowA] + sgn * d[rowB]
    const int rowA = (0x1210 >> (4 * wave)) & 3, rowB = (0x3122 >> (4 * wave)) & 3;
    const float sgn = wave == 1 ? 1.f : -1.f;
    // fragment-read offsets inside a stage: pixel (py, px = 2 tx + b) at (py * 34 + px) * 128, the lane's 8 channels = logical
    // chunks 2 kq, 2 kq + 1, physical chunk = logical ^ ((px >> 1) & 7) = logical ^ ((tx + (b >> 1)) & 7)
    unsigned foffA[2][2], foffB[2][2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned f = (unsigned)(tx * 256) + ((((unsigned)(2 * kq + h)) ^ (unsigned)((tx + sb) & 7)) << 4);
            foffA[sb][h] = f + (unsigned)(rowA * C3PW * 128);
            foffB[sb][h] = f + (unsigned)(rowB * C3PW * 128);
        }

    const int G = gridDim.x;
    const unsigned pix_bytes = (unsigned)a.D * (C3 * 4);
    // DMA piece wave + 4 i of a stage (17 pieces: i = 4 only on wave 0) holds pixels q = 8 piece + lane / 8 (q = py * 34 + px) in
    // 128-byte slots; the lane fetches LOGICAL chunk (lane % 8) ^ ((px >> 1) & 7) into slot lane % 8
    const bool five = wave == 0;
    auto decode = [&](int id, C3Item& it) {
        it.bx = id % a.bw;
        it.by = (id / a.bw) % a.bh;
        it.b = id / (a.bw * a.bh);
        const int y0 = it.by * 2 - 1, x0 = it.bx * 32 - 1;                             // patch origin (may be -1: SAME padding)
        // byte offset of depth slice 0 of the patch origin; wraps below zero for the first row / column: all sums are modulo
        // 2^32 and only used where the pixel is inside
        const unsigned base = (unsigned)((it.b * a.H + y0) * a.W + x0) * pix_bytes;
        it.okmask = 0u;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = (wave + 4 * i) * 8 + (lane >> 3);
            const int py = (q * 241) >> 13, px = q - py * C3PW;                        // q / 34 for q < 256
            const bool ok = q < C3NPIX && (unsigned)(y0 + py) < (unsigned)a.H && (unsigned)(x0 + px) < (unsigned)a.W;
            it.roff[i] = base + (unsigned)(py * a.W + px) * pix_bytes + (unsigned)((((lane & 7) ^ ((px >> 1) & 7))) << 4);
            it.okmask |= ok ? 1u << i : 0u;
        }
    };
    // input slice d of the item -> stage d % 3
    auto issue = [&](const C3Item& it, int d, int stage) {
        if (a.probe & 1) return;
        char* sb = smem + stage * C3STAGE;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (i == 4 && !five) break;
            const unsigned o = ((it.okmask >> i) & 1u) ? it.roff[i] + (unsigned)(d * (C3 * 4)) : C3OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)(sb + (wave + 4 * i) * 1024), 16, o, 0, 0, 0);
        }
    };
    // wait until only the newest fetch of this wave (5 | 4 DMAs) may be outstanding, then the workgroup barrier
    auto wait_newest = [&](bool issued) {
        if (a.probe & 8) return;
        if (!issued) C3_WAIT_BARRIER(0);
        else if (five) C3_WAIT_BARRIER(5);
        else C3_WAIT_BARRIER(4);
    };

    f32x4 acc[3][4][2];                                               // [output slice % 3][j][channel tile]
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[o][j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this wave's share of every flush: channel tile wave >> 1, output row dy = wave & 1 of the 2x2 tile (both columns);
    // C/D layout of the 16x16 MFMA with the filter as A: a lane holds channels 4 kq .. 4 kq + 3 of tile tx -> 16-byte pixels
    const int ent = wave >> 1, dy = wave & 1;
    const int nch = ent * 16 + 4 * kq;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + nch) : zero4;
    const f32x4 av = a.alpha ? *reinterpret_cast<const f32x4*>(a.alpha + nch) : zero4;

    C3Item cur;
    const int perm = (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);       // XCD-contiguous slot in a round (G % 8 == 0)
    for (int id = perm; id < a.nitems; id += G) {
        decode(id, cur);
        const int oy = cur.by * 2 + dy, ox0 = cur.bx * 32 + 2 * tx;
        const bool inb0 = oy < a.H && ox0 < a.W, inb1 = oy < a.H && ox0 + 1 < a.W;
        const size_t obase = (((size_t)(cur.b * a.H + oy) * a.W + ox0) * a.D) * C3 + nch;      // + dx * D * 32 + slice * 32
        const size_t ostep = (size_t)a.D * C3;
        issue(cur, 0, 0);
        if (a.D > 1) issue(cur, 1, 1);
        wait_newest(a.D > 1);                                         // slice 0 has landed (slice 1 may still be in flight)

        // step d (S = d % 3 at compile time): see the kernel comment.  `fl` = output slice d - 1 exists and is flushed.
        auto step = [&](auto sc, int d) {
            constexpr int S = decltype(sc)::value;                    // stage of slice d; accumulator set of OUTPUT slice d
            constexpr int SP = (S + 2) % 3, SN = (S + 1) % 3;         // ... of output slices d - 1 and d + 1
            const bool fl = d >= 1;
            const bool arith = d < a.D && !(a.probe & 2);             // d == D: the pass that only flushes output slice D - 1
            // the flush's own loads (residual of slice d - 1) go out FIRST, then the fetch of slice d + 2: the flush can then wait
            // for everything but that fetch
            f32x4 rv0 = zero4, rv1 = zero4;
            if (fl && a.res && !(a.probe & 4)) {
                if (inb0) rv0 = *reinterpret_cast<const f32x4*>(a.res + obase + (size_t)(d - 1) * C3);
                if (inb1) rv1 = *reinterpret_cast<const f32x4*>(a.res + obase + ostep + (size_t)(d - 1) * C3);
            }
            asm volatile("" ::: "memory");
            const bool issued = d + 2 < a.D;
            if (issued) issue(cur, d + 2, SP);                        // stage (d + 2) % 3 = (d - 1) % 3: read during step d - 1, free since its barrier
            if (arith) {
                const char* sb = smem + S * C3STAGE;
                f32x4 v[4][2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {                         // the lane's channels 4 h .. 4 h + 3 of its 8
                    f32x4 t[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const f32x4 dA = *reinterpret_cast<const f32x4*>(sb + b * 128 + foffA[b >> 1][h]);
                        const f32x4 dB = *reinterpret_cast<const f32x4*>(sb + b * 128 + foffB[b >> 1][h]);
                        t[b] = dA + sgn * dB;
                    }
                    v[0][h] = t[0] - t[2];
                    v[1][h] = t[1] + t[2];
                    v[2][h] = t[2] - t[1];
                    v[3][h] = t[1] - t[3];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16x8 p[3];
                    c3_split8(v[j][0], v[j][1], p);
                    constexpr int PU[6] = {2, 1, 0, 1, 0, 0}, PV[6] = {0, 1, 2, 0, 1, 0};     // i + j <= 2, smallest terms first
                    // (vector write -> MFMA operand read: the pieces are a few cycles old; nothing pads inline asm)
                    asm volatile("s_nop 7" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]));
                    // input slice d is tap 2 of output d - 1, tap 1 of output d, tap 0 of output d + 1
#pragma unroll
                    for (int k = 0; k < 6; ++k)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            c3_mfma_a(acc[SP][j][nt], U[j][2][PU[k]][nt], p[PV[k]]);
                            c3_mfma_a(acc[S][j][nt], U[j][1][PU[k]][nt], p[PV[k]]);
                            if (PU[k] == 2) c3_mfma_v(acc[SN][j][nt], U[j][0][2][nt], p[PV[k]]);       // the 8 fragments that live in VGPRs
                            else c3_mfma_a(acc[SN][j][nt], U[j][0][PU[k]][nt], p[PV[k]]);
                        }
                }
                // MFMA results -> vector reads in the flush below (the operands order the MFMAs before this)
                asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[SP][0][0]), "+v"(acc[SP][0][1]), "+v"(acc[SP][1][0]), "+v"(acc[SP][1][1]),
                             "+v"(acc[SP][2][0]), "+v"(acc[SP][2][1]), "+v"(acc[SP][3][0]), "+v"(acc[SP][3][1]) :: "memory");
            }
            // flush of output slice d - 1 (set SP).  Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]].  Column half, wave-local:
            // c[i][q] = (M[i][*] A)[q] -> exchange [i][q][channel tile]
            char* xch = smem + C3NSTG * C3STAGE + (d & 1) * C3XCH;
            if (fl && !(a.probe & 4)) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const f32x4 c0 = (acc[SP][0][nt] + acc[SP][1][nt]) + acc[SP][2][nt];
                    const f32x4 c1 = (acc[SP][1][nt] - acc[SP][2][nt]) - acc[SP][3][nt];
                    *reinterpret_cast<f32x4*>(xch + (((wave * 2 + 0) * 2 + nt) * 64 + lane) * 16) = c0;
                    *reinterpret_cast<f32x4*>(xch + (((wave * 2 + 1) * 2 + nt) * 64 + lane) * 16) = c1;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[SP][j][nt] = zero4;                                             // becomes the set of output slice d + 2
                    // the last pass also clears the set of "output slice D", which took the tap-0 products of slice D - 1
                    if (d >= a.D) acc[S][j][nt] = zero4;
                }
            // one barrier: the exchange is written, the residual is in, slice d + 1 has landed (older than the fetch just issued),
            // and every wave is done reading stage S
            wait_newest(issued);
            if (fl && !(a.probe & 4)) {
                // rows of A^T: dy = 0: c0 + c1 + c2;  dy = 1: c1 - c2 - c3
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    f32x4 ci[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        ci[k] = *reinterpret_cast<const f32x4*>(xch + ((((dy + k) * 2 + dx) * 2 + ent) * 64 + lane) * 16);
                    if (!(dx ? inb1 : inb0)) continue;
                    f32x4 o = (dy == 0 ? (ci[0] + ci[1]) + ci[2] : (ci[0] - ci[1]) - ci[2]) + bv;
                    const size_t off = obase + dx * ostep + (size_t)(d - 1) * C3;
                    if (a.z) *reinterpret_cast<f32x4*>(a.z + off) = o;
                    if (a.act & RN_ACT_PRELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f) + av[e] * fminf(o[e], 0.f);
                    }
                    if (a.act & RN_ACT_ELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : expf(o[e]) - 1.f;
                    }
                    if (a.res) o += dx ? rv1 : rv0;
                    if (a.act & RN_ACT_SIGMOID) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = 1.f / (1.f + __expf(-o[e]));
                    }
                    *reinterpret_cast<f32x4*>(a.y + off) = o;
                }
            }
        };
        // input slices 0 .. D-1, then one more pass (d = D) that only flushes output slice D - 1
        for (int d = 0; d <= a.D; d += 3) {
            step(std::integral_constant<int, 0>{}, d);
            if (d + 1 <= a.D) step(std::integral_constant<int, 1>{}, d + 1);
            if (d + 2 <= a.D) step(std::integral_constant<int, 2>{}, d + 2);
        }
        // (the accumulator sets are all zero again: every set was zeroed when its slice was flushed; the sets that took the products
        // of taps beyond the volume -- "output slices" -1 and D -- were zeroed unflushed)
    }
#endif
}

bool rn_conv3d_wino_bf3_supported(int Cin, int Cout)
{
    static const bool off = getenv("RN_NO_WINOGRAD3D_BF3") != nullptr || getenv("RN_NO_WINOGRAD3D") != nullptr || getenv("RN_NO_WINOGRAD") != nullptr;
    return !off && Cin == C3 && Cout == C3;
}

size_t rn_conv3d_wino_bf3_packed_bytes() { return (size_t)16 * 3 * C3 * C3 * 3 * 2; }

int rn_launch_conv3d_wino_pack_bf3(const float* w_tf, void* us, int transposed, hipStream_t st)
{
    hipLaunchKernelGGL(conv3d_wino_pack_bf3_kernel, dim3((16 * 3 * C3 * C3 + 255) / 256), dim3(256), 0, st, w_tf,
                       static_cast<unsigned short*>(us), transposed);
    return rn_check_launch("conv3d_wino_pack_bf3");
}

int rn_launch_conv3d_wino_bf3(const float* x, const void* us, const float* bias, const float* alpha, const float* residual,
                              float* y, float* preact, int B, int H, int W, int D, int act, hipStream_t st)
{
    if (B < 1 || H < 1 || W < 1 || D < 1) return rn_set_error(RN_E_INVALID, "conv3d_wino_bf3: bad sizes");
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "conv3d_wino_bf3: PReLU needs alpha");
    const size_t per_image = (size_t)H * W * D * C3 * 4;
    if (per_image >= 0x7fffff00ULL) return rn_set_error(RN_E_UNSUPPORTED, "conv3d_wino_bf3: one image exceeds the 2 GiB buffer window");
    const int chunk = (int)(0x7fffff00ULL / per_image);              // images per launch: byte offsets stay below 2^31 (the top bit = zero fill)
    const size_t lds = (size_t)C3NSTG * C3STAGE + 2 * C3XCH;
    auto kern = conv3d_wino_bf3_kernel;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc_ != RN_OK) return rc_; }
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = B - b0 < chunk ? B - b0 : chunk;
        const size_t off = (size_t)b0 * H * W * D * C3;
        C3Args a;
        a.x = x + off; a.u = static_cast<const char*>(us); a.bias = bias; a.alpha = alpha; a.res = residual ? residual + off : nullptr;
        a.y = y + off; a.z = preact ? preact + off : nullptr;
        a.x_bytes = (unsigned)(nb * per_image);
        a.B = nb; a.H = H; a.W = W; a.D = D;
        a.bh = (H + 1) / 2; a.bw = (W + 31) / 32;
        const long long nitems = (long long)nb * a.bh * a.bw;
        if (nitems > 0x7fffffff) return rn_set_error(RN_E_UNSUPPORTED, "conv3d_wino_bf3: too many blocks");
        a.nitems = (int)nitems; a.act = act;
        { static const int probe = getenv("RN_C3_PROBE") ? atoi(getenv("RN_C3_PROBE")) : 0; a.probe = probe; }
        const unsigned grid = nitems < 256 ? (unsigned)((nitems + 7) / 8 * 8) : 256u;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
        const int rc = rn_check_launch("conv3d_wino_bf3");
        if (rc != RN_OK) return rc;
    }
    return RN_OK;
}
