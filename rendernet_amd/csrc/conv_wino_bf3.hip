// The multiply stage of the three-launch Winograd path (conv_wino43.hip) on the bf16 matrix pipe, at fp32 accuracy.
//
// gfx950 runs f32-input MFMA at the fp32 VECTOR rate (64 FLOP/clk/SIMD, 157 TFLOP/s) and bf16-input MFMA at 16x that, both
// with fp32 accumulation.  Every fp32 operand is therefore written as the exact sum of three bf16 pieces
//     x = x0 + x1 + x2,   x0 = bf16(x),  x1 = bf16(x - x0),  x2 = bf16(x - x0 - x1)        (round to nearest even)
// (3 x 8 significand bits + the signs of the remainders cover fp32's 24; bf16 has fp32's exponent range, so nothing is
// scaled) and a product x.y is taken as the six piece products with i + j <= 2:
//     x0y0 + (x0y1 + x1y0) + (x0y2 + x1y1 + x2y0);          dropped: x1y2 + x2y1 + x2y2 <= 3 * 2^-25 |x||y|,
// i.e. below half an fp32 ulp of the product -- the class of error an fp32 FMA chain makes per term.  Six
// v_mfma_f32_32x32x16_bf16 (6 x 32 cycles per SIMD) replace eight v_mfma_f32_32x32x2_f32 (8 x 64) per 16 channels: 2.67x
// fewer matrix-pipe cycles for 1.5x the operand bytes.  The split is done where the values are produced -- never beside the
// MFMAs: the input transform writes V pre-split (wino_input_bf3_kernel), the filter transform writes U pre-split
// (wino_pack_bf3_kernel); M stays fp32 and the output transform is the one of conv_wino43.hip.
// The layers: the wide 3x3 / 4x4 stride-1 2-D convs (res_block_2d, *_skip, e_conv5, e_conv6: tools/layer_util.py:91-105,
// RenderNet_Shader.py:71-103).
//
// Layouts (K step = 16 channels; a "row" = the 3 x 16 bf16 of one tile / one output channel for one K step = 96 bytes):
//     Vs [nxi][Cin/16][T][3 planes][2 chunks of 8 bf16]           Us [nxi][Cout/256][Cin/16][256][3][2 x 8]
// so that what a 256 x 256 GEMM block needs for one K step is two contiguous 24-KiB pieces that go global -> LDS by DMA
// verbatim (48 wave instructions of 1 KiB, lane-linear on both sides).  The two 16-byte chunks of a plane are stored
// swapped when bit 3 of the row index is set: with 96-byte rows that makes the ds_read_b128 fragment reads (lane = row,
// 16 lanes per LDS cycle) conflict-free.
// GEMM: 512 threads = 8 waves (4 along tiles x 2 along channels), a wave = 64 tiles x 128 channels (128 accumulators),
// U is the A operand (accumulator registers = 4 consecutive output channels: 16-byte stores), three LDS stages of 48 KiB,
// DMA two K steps ahead with counted vmcnt (the loads stay in flight across the one barrier of a step), persistent grid
// with the XCD-contiguous item order of the fp32 kernel.  Format B3 multiplies on v_mfma_f32_16x16x32_bf16 with the K = 32 of an
// instruction = 16 channels x two pieces (three instructions per 16 x 16 tile and K step, round 6: the shape the chip sustains 13 %
// faster at its power limit); format H2 and RN_WINO_BF3_P16=0 on v_mfma_f32_32x32x16_bf16, one product per instruction.
#include "rn_common.h"
#include "wino_mats.h"
#include <stdlib.h>

// Cache-policy switches of the stage's streams, for same-box A/B builds (scripts/build_variant.py -D...; the product builds with the
// defaults).  aux of the buffer instructions: 0 default, 2 = nt, 16 = sc1 (write-through / L1-bypassing), 17 = sc0 sc1.
#ifndef RN_BF3_M_AUX
#define RN_BF3_M_AUX 0          // the GEMM's stores of M (written once, read once by the output transform of another launch)
#endif
#ifndef RN_BF3_VLOAD_AUX
#define RN_BF3_VLOAD_AUX 0      // the GEMM's LDS-DMA loads of V (a slab is read by the 4 CUs of one XCD that share the row block)
#endif
#ifndef RN_BF3_ULOAD_AUX
#define RN_BF3_ULOAD_AUX 0      // ... of U (read by the 8 CUs of an XCD that share the channel block, and again every round)
#endif
#ifndef RN_P16_SCHED
#define RN_P16_SCHED 2          // 16x16x32 form, placement of a channel group's U reads + DMA piece: 0 ahead of its MFMAs (pinned), 1 compiler's choice, 2 behind its first four MFMAs
#endif
#ifndef RN_BF3_VSTORE_NT
#define RN_BF3_VSTORE_NT 0      // the input transform's stores of V as non-temporal stores
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Operand formats of the GEMM stage.  A row of an operand panel holds, per K step of 16, NP planes of 16 values (32 bytes each).
//   B3: three bf16 pieces (exact sum = the fp32 value), six piece products i + j <= 2.  Rows of 96 bytes; the two 16-byte chunks
//       of a plane are swapped in rows with bit 3 set.
//   H2: the fp32 value divided by a power-of-two scale of its tensor, as two fp16 pieces (22-bit mantissa; values below 2^-18 of
//       the scaled maximum lose relative, not absolute, precision), three products (h0 h0, h0 h1, h1 h0); the accumulators are
//       multiplied by the two scales on the way out.  Rows of 64 bytes; the four chunks of a row are XORed with bits 2..3 of the
//       row index.  Either way the 16 lanes of a ds_read_b128 group (16 consecutive rows) hit 16 different bank groups.
// (a named namespace: profiler tables then show wino_gemm_bf3_kernel<rnf::FmtH2, 4, 2>)
namespace rnf {
struct FmtB3 {
    static constexpr int NP = 3, ROW = 96, NPROD = 6, ID = 0;
    typedef bf16x8 frag;
    static constexpr int PU[6] = {2, 1, 0, 1, 0, 0}, PV[6] = {0, 1, 2, 0, 1, 0};        // smallest terms first
    __device__ static __forceinline__ unsigned chunk(int row, int p, int hb) { return (unsigned)(p * 32) + ((((unsigned)hb) ^ (unsigned)((row >> 3) & 1)) << 4); }
    __device__ static __forceinline__ f32x16 mfma(const frag& u, const frag& v, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(u, v, c, 0, 0, 0); }
};
struct FmtH2 {
    static constexpr int NP = 2, ROW = 64, NPROD = 3, ID = 1;
    typedef f16x8 frag;
    static constexpr int PU[6] = {1, 0, 0, 0, 0, 0}, PV[6] = {0, 1, 0, 0, 0, 0};
    __device__ static __forceinline__ unsigned chunk(int row, int p, int hb) { return (((unsigned)(p * 2 + hb)) ^ (unsigned)((row >> 2) & 3)) << 4; }
    __device__ static __forceinline__ f32x16 mfma(const frag& u, const frag& v, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(u, v, c, 0, 0, 0); }
};
}  // namespace rnf
using rnf::FmtB3;
using rnf::FmtH2;

namespace {

// two values at once: one v_cvt_pk_bf16_f32 per piece pair gives the packed word that is stored, its two halves widened again (a shift, a
// mask) feed one packed subtraction -- 9 instructions per pair where split3<2> compiles to 15 (a conversion per value AND the packed one)
__device__ __forceinline__ void split3_pair(f32x2 x, unsigned (&w)[3])
{
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        w[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
        if (q < 2) {
            f32x2 h;
            h[0] = __builtin_bit_cast(float, w[q] << 16);
            h[1] = __builtin_bit_cast(float, w[q] & 0xffff0000u);
            x -= h;
        }
    }
}

// the power-of-two scale of a tensor in format H2 from the largest magnitude of its UNtransformed values (0 -> 1)
__host__ __device__ inline float h2_scale(float amax, float bound)
{
    const float t = amax * bound * (1.0f / 32768.0f);
    if (!(t > 0.f)) return 1.f;                     // 0, and NaN (the max reduction drops NaNs; a NaN word itself lands here too)
    // max|x| = inf (an overflowed activation) or a product beyond the fp32 range: frexpf(inf) leaves the exponent unspecified.  The largest
    // power-of-two scale is used instead -- finite values then shrink towards 0, the inf itself stays inf in the fp16 piece and the output
    // is inf / NaN like the fp32 route's: deterministic, never an arbitrary scale.
    if (!(t <= 3.0e38f)) return 8.507059e37f;       // 2^126
    int e;
    const float m = frexpf(t, &e);                  // t = m * 2^e, 0.5 <= m < 1
    return ldexpf(1.f, m == 0.5f ? e - 1 : e);
}

constexpr int SB_ROW = 96;                                   // bytes per row and K step
constexpr int SB_BM = 256, SB_BN = 256;
constexpr int SB_UB = SB_BN * SB_ROW;                        // U part of a stage: 24 KiB (V: 24 | 12 KiB)
constexpr int SB_NSTAGE = 3;

__device__ __forceinline__ unsigned xcd_contiguous(unsigned blk, unsigned nblk8) { return (blk & 7u) * (nblk8 >> 3) + (blk >> 3); }

// x -> three bf16 pieces (v_cvt_pk_bf16_f32 rounds to nearest even; the remainders are exact in fp32)
template <int VW>
__device__ __forceinline__ void split3(const float (&x)[VW], unsigned short (&p)[3][VW])
{
#pragma unroll
    for (int e = 0; e < VW; ++e) {
        const __bf16 h0 = (__bf16)x[e];
        const float r1 = x[e] - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const float r2 = r1 - (float)h1;
        const __bf16 h2 = (__bf16)r2;
        p[0][e] = __builtin_bit_cast(unsigned short, h0);
        p[1][e] = __builtin_bit_cast(unsigned short, h1);
        p[2][e] = __builtin_bit_cast(unsigned short, h2);
    }
}

// B^T applied to one 8-vector, F(6x6,3x3): rows 1..6 come in +/- pairs over the even and the odd inputs, rows 0 and 7 are a
// difference pair each -- 26 operations where the dense row-by-row form (44 non-zeros) takes 44.  The input transform is
// issue-bound (4.7 TB/s of a 7.0 TB/s pure-write stream), so the count matters; the sums are the same to rounding order.
template <class S, class V>
__device__ __forceinline__ void bt_apply(const V (&d)[S::TA], V (&o)[S::TA])
{
    if constexpr (S::TA == 8 && S::R == 3) {
        o[0] = (d[6] - d[0]) + 5.25f * (d[2] - d[4]);
        o[7] = (d[7] - d[1]) + 5.25f * (d[3] - d[5]);
        const V e1 = (d[2] + d[6]) - 4.25f * d[4], f1 = (d[1] + d[5]) - 4.25f * d[3];
        o[1] = e1 + f1;
        o[2] = e1 - f1;
        const V e2 = (d[6] + 0.25f * d[2]) - 1.25f * d[4], f2 = (0.5f * d[1] - 2.5f * d[3]) + 2.f * d[5];
        o[3] = e2 + f2;
        o[4] = e2 - f2;
        const V e3 = (d[6] + 4.f * d[2]) - 5.f * d[4], f3 = (2.f * d[1] - 2.5f * d[3]) + 0.5f * d[5];
        o[5] = e3 + f3;
        o[6] = e3 - f3;
    } else {
#pragma unroll
        for (int i = 0; i < S::TA; ++i) {
            V acc = V(0.f);
#pragma unroll
            for (int k = 0; k < S::TA; ++k) {
                const float cf = S::BT(i, k);
                if (cf != 0.f) acc += cf * d[k];
            }
            o[i] = acc;
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// 1. input transform V = B^T d B, written as three bf16 planes in the GEMM's row layout: the fp32 V of wino_input_kernel
// (conv_wino43.hip) to rounding order (F(6x6,3x3) applies B^T in its factored form, bt_apply), only its representation changes.
// thread = (tile, 2 channels); workgroup = 8 consecutive tiles x 64 channels (4 K-step groups), 32 lanes per tile: a load
// instruction reads two 256-byte runs.  What the workgroup produces for one (xi, K-step group) is 8 rows x 96 bytes =
// 768 CONTIGUOUS bytes of Vs, but a thread holds only 4 bytes per plane of it -- measured on the res2 shape (B = 24): three
// 4-byte stores per thread and xi 0.50 ms, the same bytes as whole-line 16-byte stores 0.3 ms.  So the pieces of one
// output row i (A xi planes) are exchanged through LDS: every thread writes its 3 x A words, one barrier, then the
// workgroup streams the A x 4 segments out in 16-byte stores, 48 consecutive lanes per segment (two LDS buffers: the
// next row's writes need no second barrier).
constexpr int IB_TILES = 8, IB_SEG = IB_TILES * SB_ROW + 32;  // LDS bytes per (xi, K-step group) segment: 768 + 32 (bank spread)

template <class S>
__global__ __launch_bounds__(256)
void wino_input_bf3_kernel(const float* __restrict__ x, char* __restrict__ Vs, int H, int W, int C, int th, int tw,
                           long long T, unsigned ncb, unsigned nwg, unsigned nblk8, int pad_lo)
{
    typedef float vec __attribute__((ext_vector_type(2)));
    constexpr int A = S::TA;
    constexpr int NSEG = A * 4, BUF = NSEG * IB_SEG;
    __shared__ __attribute__((aligned(16))) char xch[2 * BUF];
    const unsigned blk = xcd_contiguous(blockIdx.x, nblk8);
    if (blk >= nwg) return;                                    // (whole workgroups only: no barrier is skipped)
    const unsigned cb = blk % ncb;
    const long long tg = blk / ncb;
    const int tid = threadIdx.x, l32 = tid & 31, tl = tid >> 5;
    const int c = (int)cb * 64 + l32 * 2;
    const long long t0 = tg * IB_TILES, t = t0 + tl;
    const bool live = t < T && c < C;
    vec tt[A][A];                                              // (B^T d)[i][col]
    {
        const long long tc = live ? t : 0;
        const int tx = (int)(tc % tw), ty = (int)((tc / tw) % th);
        const long long b = tc / ((long long)tw * th);
        const int y0 = S::M * ty - pad_lo, x0 = S::M * tx - pad_lo;
        // one 64-bit multiply per thread: the A x A addresses are the tile's corner (possibly outside the plane -- then never
        // dereferenced) plus offsets r W C + col C that are the same for every lane, i.e. scalar arithmetic
        const float* p0 = x + (((long long)b * H + y0) * W + x0) * (long long)C + (live ? c : 0);
        const long long rs = (long long)W * C;
        const float* prow[A];                                  // (A row pointers: the column step col C is then one scalar-offset add per load)
#pragma unroll
        for (int r = 0; r < A; ++r) prow[r] = r == 0 ? p0 : prow[r - 1] + rs;
#pragma unroll
        for (int col = 0; col < A; ++col) {
            vec d[A];
            const bool cok = live && (unsigned)(x0 + col) < (unsigned)W;
#pragma unroll
            for (int r = 0; r < A; ++r) {
                const bool ok = cok && (unsigned)(y0 + r) < (unsigned)H;
                d[r] = ok ? *reinterpret_cast<const vec*>(prow[r] + col * C) : vec(0.f);
            }
            vec o[A];
            bt_apply<S>(d, o);
#pragma unroll
            for (int i = 0; i < A; ++i) tt[i][col] = o[i];
        }
    }
    // LDS position of this thread's word of plane 0 in segment (j = 0, its K-step group): row tl, chunk (l32 % 8) / 4 swapped
    // when bit 3 of the tile index is set (t0 is a multiple of 8: the bit is that of tg), word l32 % 4
    const unsigned sw = (unsigned)(tg & 1);
    const unsigned wofs = (unsigned)((l32 >> 3) * IB_SEG + tl * SB_ROW) + ((((unsigned)(l32 >> 2) & 1u) ^ sw) << 4) + (unsigned)(l32 & 3) * 4;
    const size_t xi_stride = (size_t)T * C * 6;                // bytes per xi: (C / 16) K steps x T rows x 96
    const size_t step_stride = (size_t)T * SB_ROW;
    char* vbase = Vs + ((size_t)cb * 4 * T + t0) * SB_ROW;     // segment (xi = 0, K-step group 4 cb) of this tile group
    const int tiles_here = (int)((T - t0) < IB_TILES ? (T - t0) : IB_TILES);
    // the way out: the four K-step-group segments of one xi are 4 x 48 = 192 16-byte chunks = three store instructions of a whole wave;
    // wave w takes xi columns j = w, w + 4.  What depends on the lane (segment, chunk, the ragged-edge predicate) is fixed for the kernel.
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned loff[3];
    size_t goff[3];
    bool cok[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int q = (tid & 63) + 64 * k, sg = q / 48, r = q - sg * 48;
        loff[k] = (unsigned)(sg * IB_SEG + r * 16);
        goff[k] = sg * step_stride + (size_t)(r * 16);
        cok[k] = r < tiles_here * 6 && (int)cb * 4 + sg < (C >> 4);
    }
#pragma unroll
    for (int i = 0; i < A; ++i) {
        char* buf = xch + (i & 1) * BUF;
        vec vrow[A];
        bt_apply<S>(tt[i], vrow);
#pragma unroll
        for (int j = 0; j < A; ++j) {
            unsigned pw[3];
            split3_pair(vrow[j], pw);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                *reinterpret_cast<unsigned*>(buf + j * (4 * IB_SEG) + wofs + q * 32) = pw[q];
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < (A + 3) / 4; ++jj) {
            const int j = wv + 4 * jj;                         // wave-uniform: the xi's address is scalar arithmetic
            if (j >= A) break;
            char* gb = vbase + (size_t)(i * A + j) * xi_stride;
            const char* lb = buf + j * (4 * IB_SEG);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (cok[k]) {
                    const u32x4 vv = *reinterpret_cast<const u32x4*>(lb + loff[k]);
                    if (RN_BF3_VSTORE_NT) __builtin_nontemporal_store(vv, reinterpret_cast<u32x4*>(gb + goff[k]));
                    else *reinterpret_cast<u32x4*>(gb + goff[k]) = vv;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// filter transform U = G g G^T (double, as wino_pack_kernel), rounded to fp32 and split: Us [nxi][Cout/256][Cin/16][256][3][16].
// Once per weight update -- every step when training, for the forward AND the input-gradient pack of every layer, so its
// stores matter: workgroup = 64 output channels x one 16-channel K step (thread = (channel, 4 input channels)); what it
// produces per xi is 64 rows x 96 bytes = 6 KiB CONTIGUOUS, exchanged through LDS four xi at a time and written as 16-byte
// stores (the straight form -- 8-byte stores at a 96-byte stride -- ran at 1 TB/s: 0.38 ms for the res2 filter).
constexpr int PK_CO = 64, PK_XB = 4, PK_PITCH = 112;          // LDS row pitch (96 + 16: 16-byte aligned rows, 2-way write conflicts at worst)

template <class S>
__global__ __launch_bounds__(256)
void wino_pack_bf3_kernel(const float* __restrict__ w_tf, char* __restrict__ us, int Cin, int Cout, int transposed)
{
    constexpr int A = S::TA, R = S::R, NXI = A * A;
    __shared__ __attribute__((aligned(16))) char xch[PK_XB * PK_CO * PK_PITCH];
    const int ksteps = Cin / 16, nblocks = Cout / 256, cgroups = Cout / PK_CO;
    const int cg = blockIdx.x % cgroups, s = blockIdx.x / cgroups;          // consecutive workgroups: neighbouring channel groups of one K step
    const int tid = threadIdx.x, col = tid & 63, kgl = tid >> 6;            // a wave = 64 channels x one group of 4 input channels
    const int co = cg * PK_CO + col, kg = s * 4 + kgl;
    float g[R][R][4];
#pragma unroll
    for (int p_ = 0; p_ < R; ++p_)
#pragma unroll
        for (int q = 0; q < R; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = kg * 4 + r;
                g[p_][q][r] = transposed ? w_tf[((size_t)((R - 1 - p_) * R + (R - 1 - q)) * Cout + co) * Cin + c]
                                         : w_tf[((size_t)(p_ * R + q) * Cin + c) * Cout + co];
            }
    double gg[A][R][4];                                     // (G g)[i][q]
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int q = 0; q < R; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int p_ = 0; p_ < R; ++p_) acc = __builtin_fma(S::G(i, p_), (double)g[p_][q][r], acc);      // explicit: the fp32 and the split pack must round alike
                gg[i][q][r] = acc;
            }
    // this thread's 8 bytes of plane 0 inside its row: chunk (4 kgl) / 8 swapped when bit 3 of the row (= channel within the
    // 256-block) is set, then the second half of the chunk for odd kgl
    const int slot = co & 255, nb = co >> 8;
    const unsigned wofs = (unsigned)(col * PK_PITCH) + ((((unsigned)kgl >> 1) ^ (unsigned)((slot >> 3) & 1)) << 4) + (unsigned)(kgl & 1) * 8;
    const size_t plane = (size_t)nblocks * ksteps * 256 * SB_ROW;                                   // bytes per xi
    char* ubase = us + (((size_t)nb * ksteps + s) * 256 + (slot - col)) * SB_ROW;                   // row of this group's first channel, xi = 0
    constexpr int NB = (NXI + PK_XB - 1) / PK_XB;
#pragma unroll
    for (int xb = 0; xb < NB; ++xb) {
#pragma unroll
        for (int e = 0; e < PK_XB; ++e) {
            const int xi = xb * PK_XB + e;                      // compile-time after unrolling: the matrix entries fold
            if (xi >= NXI) continue;
            const int i = xi / A, j = xi % A;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < R; ++q) acc = __builtin_fma(gg[i][q][r], S::G(j, q), acc);
                o[r] = (float)acc;
            }
            unsigned short p[3][4];
            split3<4>(o, p);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                uint2 v;
                v.x = (unsigned)p[q][0] | ((unsigned)p[q][1] << 16);
                v.y = (unsigned)p[q][2] | ((unsigned)p[q][3] << 16);
                *reinterpret_cast<uint2*>(xch + e * (PK_CO * PK_PITCH) + wofs + q * 32) = v;
            }
        }
        __syncthreads();
        // 4 xi x 64 rows x 6 chunks of 16 bytes = 1536 chunks: 6 per thread, 384 consecutive lanes per xi
        for (int qd = tid; qd < PK_XB * PK_CO * 6; qd += 256) {
            const int e = qd / (PK_CO * 6), rr = qd - e * (PK_CO * 6);
            const int xi = xb * PK_XB + e;
            if (xi >= NXI) continue;
            const int row = rr / 6, ch = rr - row * 6;
            const u32x4 v = *reinterpret_cast<const u32x4*>(xch + e * (PK_CO * PK_PITCH) + row * PK_PITCH + ch * 16);
            *reinterpret_cast<u32x4*>(ubase + (size_t)xi * plane + rr * 16) = v;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Format H2: the two transforms again, writing two fp16 pieces of value / scale in rows of 64 bytes ([2 planes][16]; the four
// chunks of a row XORed with bits 2..3 of the row index).  The scale is a power of two derived from max|x| of the UNtransformed
// tensor (absmax_kernel -> a device word) times the factor by which the transform can grow a value (the squared largest absolute
// row sum of B^T, resp. G), so that every transformed value / scale is below 2^15: no overflow, and everything above 2^-18 of that
// keeps 22 mantissa bits.
template <class S> struct H2Bound {
    static constexpr float bt()
    {
        float m = 0.f;
        for (int i = 0; i < S::TA; ++i) { float r = 0.f; for (int k = 0; k < S::TA; ++k) r += S::BT(i, k) < 0.f ? -S::BT(i, k) : S::BT(i, k); m = r > m ? r : m; }
        return m * m;
    }
    static constexpr float g()
    {
        double m = 0.0;
        for (int i = 0; i < S::TA; ++i) { double r = 0.0; for (int k = 0; k < S::R; ++k) r += S::G(i, k) < 0.0 ? -S::G(i, k) : S::G(i, k); m = r > m ? r : m; }
        return (float)(m * m);
    }
};

// *dst = src ? *src : 0.  The hand-over words are set and copied by a one-thread KERNEL, not by hipMemsetAsync / hipMemcpyAsync:
// inside a captured hipGraph the memset / memcpy nodes of this ROCm were seen to run out of order with the kernels around them
// (replays with new inputs used the previous replay's maxima: tests/test_gpu_wino_split.py::test_hipgraph_replay_of_the_split_routes).
__global__ void word_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ src) { *dst = src ? *src : 0u; }

// max |x| over n floats (n % 4 == 0, 16-byte aligned) into *out (bit pattern of a non-negative float: unsigned order = float order;
// *out must be zero before the launch)
__global__ __launch_bounds__(256)
void absmax_kernel(const float* __restrict__ x, size_t n4, unsigned* __restrict__ out)
{
    __shared__ float part[4];
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        // one candidate per workgroup, and an atomic only when it beats the running maximum (a plain read first): thousands of atomics
        // on ONE address serialise in the L2 -- 0.10 ms for a 100-MB tensor with one atomic per wave, where the read takes 0.02 ms
        const unsigned bits = __builtin_bit_cast(unsigned, fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])));
        if (bits > *reinterpret_cast<volatile unsigned*>(out)) atomicMax(out, bits);
    }
}

__device__ __forceinline__ unsigned h2_word(float a, float b, unsigned& lo)
{
    // two scaled values -> the word of their first pieces (return) and of their second pieces (lo)
    const _Float16 a0 = (_Float16)a, b0 = (_Float16)b;
    const _Float16 a1 = (_Float16)(a - (float)a0), b1 = (_Float16)(b - (float)b0);
    lo = (unsigned)__builtin_bit_cast(unsigned short, a1) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
    return (unsigned)__builtin_bit_cast(unsigned short, a0) | ((unsigned)__builtin_bit_cast(unsigned short, b0) << 16);
}

constexpr int IH_ROW = 64, IH_SEG = IB_TILES * IH_ROW + 32;   // LDS bytes per (xi, K-step group) segment: 512 + 32

template <class S>
__global__ __launch_bounds__(256)
void wino_input_h2_kernel(const float* __restrict__ x, char* __restrict__ Vs, const unsigned* __restrict__ amax, int H, int W, int C, int th, int tw,
                          long long T, unsigned ncb, unsigned nwg, unsigned nblk8, int pad_lo)
{
    typedef float vec __attribute__((ext_vector_type(2)));
    constexpr int A = S::TA;
    constexpr int NSEG = A * 4, BUF = NSEG * IH_SEG;
    __shared__ __attribute__((aligned(16))) char xch[2 * BUF];
    const unsigned blk = xcd_contiguous(blockIdx.x, nblk8);
    if (blk >= nwg) return;
    const float inv = 1.f / h2_scale(__builtin_bit_cast(float, *amax), H2Bound<S>::bt());
    const unsigned cb = blk % ncb;
    const long long tg = blk / ncb;
    const int tid = threadIdx.x, l32 = tid & 31, tl = tid >> 5;
    const int c = (int)cb * 64 + l32 * 2;
    const long long t0 = tg * IB_TILES, t = t0 + tl;
    const bool live = t < T && c < C;
    vec tt[A][A];                                              // (B^T d)[i][col]
    {
        const long long tc = live ? t : 0;
        const int tx = (int)(tc % tw), ty = (int)((tc / tw) % th);
        const long long b = tc / ((long long)tw * th);
        const int y0 = S::M * ty - pad_lo, x0 = S::M * tx - pad_lo;
        const float* p0 = x + (((long long)b * H + y0) * W + x0) * (long long)C + (live ? c : 0);   // (addresses as in wino_input_bf3_kernel)
        const long long rs = (long long)W * C;
        const float* prow[A];
#pragma unroll
        for (int r = 0; r < A; ++r) prow[r] = r == 0 ? p0 : prow[r - 1] + rs;
#pragma unroll
        for (int col = 0; col < A; ++col) {
            vec d[A];
            const bool cok = live && (unsigned)(x0 + col) < (unsigned)W;
#pragma unroll
            for (int r = 0; r < A; ++r) {
                const bool ok = cok && (unsigned)(y0 + r) < (unsigned)H;
                d[r] = ok ? *reinterpret_cast<const vec*>(prow[r] + col * C) : vec(0.f);
            }
            vec o[A];
            bt_apply<S>(d, o);
#pragma unroll
            for (int i = 0; i < A; ++i) tt[i][col] = o[i];
        }
    }
    // LDS position of this thread's word of plane q in segment (j = 0, its K-step group): row tl, logical chunk 2 q + (l32 % 8) / 4
    // XORed with bits 2..3 of the tile index, word l32 % 4
    const unsigned swz = (unsigned)((t >> 2) & 3);
    const unsigned wbase = (unsigned)((l32 >> 3) * IH_SEG + tl * IH_ROW) + (unsigned)(l32 & 3) * 4;
    const unsigned hbit = (unsigned)(l32 >> 2) & 1u;
    const size_t xi_stride = (size_t)T * C * 4;                // bytes per xi: (C / 16) K steps x T rows x 64
    const size_t step_stride = (size_t)T * IH_ROW;
    char* vbase = Vs + ((size_t)cb * 4 * T + t0) * IH_ROW;
    const int tiles_here = (int)((T - t0) < IB_TILES ? (T - t0) : IB_TILES);
    // the way out as in wino_input_bf3_kernel: the four segments of one xi are 4 x 32 chunks = two store instructions of a whole wave
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned loff[2];
    size_t goff[2];
    bool cok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = (tid & 63) + 64 * k, sg = q >> 5, r = q & 31;
        loff[k] = (unsigned)(sg * IH_SEG + r * 16);
        goff[k] = sg * step_stride + (size_t)(r * 16);
        cok[k] = r < tiles_here * 4 && (int)cb * 4 + sg < (C >> 4);
    }
#pragma unroll
    for (int i = 0; i < A; ++i) {
        char* buf = xch + (i & 1) * BUF;
        vec vrow[A];
        bt_apply<S>(tt[i], vrow);
#pragma unroll
        for (int j = 0; j < A; ++j) {
            unsigned lo;
            const unsigned hi = h2_word(vrow[j][0] * inv, vrow[j][1] * inv, lo);
            *reinterpret_cast<unsigned*>(buf + j * (4 * IH_SEG) + wbase + (((0u + hbit) ^ swz) << 4)) = hi;
            *reinterpret_cast<unsigned*>(buf + j * (4 * IH_SEG) + wbase + (((2u + hbit) ^ swz) << 4)) = lo;
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < (A + 3) / 4; ++jj) {
            const int j = wv + 4 * jj;
            if (j >= A) break;
            char* gb = vbase + (size_t)(i * A + j) * xi_stride;
            const char* lb = buf + j * (4 * IH_SEG);
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (cok[k]) {
                    const u32x4 vv = *reinterpret_cast<const u32x4*>(lb + loff[k]);
                    if (RN_BF3_VSTORE_NT) __builtin_nontemporal_store(vv, reinterpret_cast<u32x4*>(gb + goff[k]));
                    else *reinterpret_cast<u32x4*>(gb + goff[k]) = vv;
                }
        }
    }
}

// filter transform in format H2: Us [nxi][Cout/256][Cin/16][256][2][16] fp16 of U / scale; same workgroup shape and LDS exchange
// as wino_pack_bf3_kernel (64 rows x 64 bytes = 4 KiB contiguous per xi)
constexpr int PH_PITCH = 80;                                  // 64 + 16

template <class S>
__global__ __launch_bounds__(256)
void wino_pack_h2_kernel(const float* __restrict__ w_tf, char* __restrict__ us, const unsigned* __restrict__ amax, int Cin, int Cout, int transposed)
{
    constexpr int A = S::TA, R = S::R, NXI = A * A;
    __shared__ __attribute__((aligned(16))) char xch[PK_XB * PK_CO * PH_PITCH];
    const float inv = 1.f / h2_scale(__builtin_bit_cast(float, *amax), H2Bound<S>::g());
    const int ksteps = Cin / 16, nblocks = Cout / 256, cgroups = Cout / PK_CO;
    const int cg = blockIdx.x % cgroups, s = blockIdx.x / cgroups;
    const int tid = threadIdx.x, col = tid & 63, kgl = tid >> 6;            // a wave = 64 channels x one group of 4 input channels
    const int co = cg * PK_CO + col, kg = s * 4 + kgl;
    float g[R][R][4];
#pragma unroll
    for (int p_ = 0; p_ < R; ++p_)
#pragma unroll
        for (int q = 0; q < R; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = kg * 4 + r;
                g[p_][q][r] = transposed ? w_tf[((size_t)((R - 1 - p_) * R + (R - 1 - q)) * Cout + co) * Cin + c]
                                         : w_tf[((size_t)(p_ * R + q) * Cin + c) * Cout + co];
            }
    double gg[A][R][4];                                     // (G g)[i][q]
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int q = 0; q < R; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int p_ = 0; p_ < R; ++p_) acc = __builtin_fma(S::G(i, p_), (double)g[p_][q][r], acc);
                gg[i][q][r] = acc;
            }
    // this thread's 8 bytes of plane q inside its row: logical chunk 2 q + (4 kgl) / 8, XORed with bits 2..3 of the row (= channel within
    // the 256-block), second half of the chunk for odd kgl
    const int slot = co & 255, nb = co >> 8;
    const unsigned swz = (unsigned)((slot >> 2) & 3), hbit = (unsigned)kgl >> 1;
    const unsigned wbase = (unsigned)(col * PH_PITCH) + (unsigned)(kgl & 1) * 8;
    const size_t plane = (size_t)nblocks * ksteps * 256 * IH_ROW;                                   // bytes per xi
    char* ubase = us + (((size_t)nb * ksteps + s) * 256 + (slot - col)) * IH_ROW;
    constexpr int NB = (NXI + PK_XB - 1) / PK_XB;
#pragma unroll
    for (int xb = 0; xb < NB; ++xb) {
#pragma unroll
        for (int e = 0; e < PK_XB; ++e) {
            const int xi = xb * PK_XB + e;
            if (xi >= NXI) continue;
            const int i = xi / A, j = xi % A;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < R; ++q) acc = __builtin_fma(gg[i][q][r], S::G(j, q), acc);
                o[r] = (float)acc * inv;
            }
            uint2 hi, lo;
            hi.x = h2_word(o[0], o[1], lo.x);
            hi.y = h2_word(o[2], o[3], lo.y);
            *reinterpret_cast<uint2*>(xch + e * (PK_CO * PH_PITCH) + wbase + (((0u + hbit) ^ swz) << 4)) = hi;
            *reinterpret_cast<uint2*>(xch + e * (PK_CO * PH_PITCH) + wbase + (((2u + hbit) ^ swz) << 4)) = lo;
        }
        __syncthreads();
        // 4 xi x 64 rows x 4 chunks of 16 bytes = 1024 chunks: 4 per thread
        for (int qd = tid; qd < PK_XB * PK_CO * 4; qd += 256) {
            const int e = qd / (PK_CO * 4), rr = qd - e * (PK_CO * 4);
            const int xi = xb * PK_XB + e;
            if (xi >= NXI) continue;
            const int row = rr / 4, ch = rr - row * 4;
            const u32x4 v = *reinterpret_cast<const u32x4*>(xch + e * (PK_CO * PH_PITCH) + row * PH_PITCH + ch * 16);
            *reinterpret_cast<u32x4*>(ubase + (size_t)xi * plane + rr * 16) = v;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2. the GEMMs  M[xi] = V[xi] (T x Cin) . U[xi] (Cin x Cout) on split operands
struct Bf3GemmArgs {
    const char* V; const char* U; float* M;
    long long T;
    int Cin, Cout;
    int mblocks, nblocks, ksteps;   // row blocks of T this launch walks (from mb_begin), 256-channel blocks, K steps of 16
    int mb_begin, parts;            // parts: BM-row parts of a block that are enumerated (4 / WM; fewer for the ragged last block)
    int mrows;                      // rows between consecutive row blocks: 256, or BM itself when all of T is walked in BM-row items
    int item_begin, item_end;       // this launch's range of the items L = (xi*mblocks + mb - mb_begin)*nblocks + nb
    unsigned v_step_bytes;          // T * 96: one (xi, K step) sub-plane of Vs
    unsigned m_bytes;               // one xi plane of M
    int probe;                      // RN_WINO_BF3_PROBE (timing experiments; results are wrong when set): 1 no DMA in the loop, 2 no stores
    const unsigned* amax_v; const unsigned* amax_u;   // format H2: bit patterns of max|x| of the two tensors (device), and the factors that
    float bound_v, bound_u;                           // bound the transformed values by them: scale = 2^ceil(log2(bound * max / 2^15))
};


#define BF3_WAIT_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")

// WM = waves along the tile rows: 4 -> block 256 rows x 256 channels (waves 4 x 2, wave tile 64 x 128 = 2 x 4 MFMA tiles, 128
// accumulators); 2 -> block 128 x 256 (waves 2 x 4, wave tile 64 x 64 = 2 x 2 tiles): the launcher runs ragged row blocks, the
// items of a last partial round and small batches as half items.
// TAG only names the kernel per layer class in profiler tables (0: F43, 1: F44, 2: F63 Cin >= 1024, 3: F63 narrower, 4: filter gradient, 5: 1x1 filter)
template <class F, int WM, int TAG, bool P16 = false>
__global__ __launch_bounds__(512, 2)
void wino_gemm_bf3_kernel(const Bf3GemmArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename F::frag frag;
    constexpr int NP = F::NP, SB_ROW = F::ROW, SB_UB = SB_BN * SB_ROW;  // (shadow the file constants: they are format B3's)
    constexpr int WN = 8 / WM, NT2 = 8 / WN;                          // 32-channel MFMA tiles per wave along channels (4 | 2)
    constexpr int BM = WM * 64, VB = BM * SB_ROW, STAGE = VB + SB_UB;  // B3: V 24 | 12 KiB + U 24 KiB per stage; H2: 16 | 8 + 16
    constexpr int VP = VB / 1024, UPW = SB_UB / 8192;                 // V DMA pieces per stage (B3 24 | 12, H2 16 | 8); U pieces per wave (3 | 2)
    constexpr int NSTORE = 2 * NT2 * 4;                               // 16-byte stores of a wave's epilogue (32 | 16)
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [stage][V BM x 96 | U 256 x 96]
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hb = lane >> 5;                       // 32x32x16 MFMA: lane = (row / column, k group of 8)
    const int wm = wave / WN, wn = wave % WN;
    // fragment of plane p: row l32 of the wave's tile (rows of a tile start at multiples of 32: the swizzle bits are l32's)
    unsigned vfrag[NP], ufrag[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        vfrag[p] = (unsigned)((wm * 64 + l32) * SB_ROW) + F::chunk(l32, p, hb);
        ufrag[p] = (unsigned)(VB + (wn * (NT2 * 32) + l32) * SB_ROW) + F::chunk(l32, p, hb);
    }
    const unsigned dma_lane = (unsigned)(wave * 1024 + lane * 16);
    const bool vextra = wave < VP % 8;                                // this wave carries one V piece more than VP / 8 (WM = 2: waves 0..3)

    struct Item { const char* vplane; const char* upanel; float* mplane; long long m0; int nb; };
    const int rounds_total = (a.item_end - a.item_begin) * a.parts;
    const int perm = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);   // XCD-contiguous slot in a round
    auto decode = [&](int r, Item& it) -> bool {
        const int id = r * (int)gridDim.x + perm;                     // gridDim.x is a multiple of 8
        if (id >= rounds_total) return false;
        const int L = a.item_begin + id / a.parts, h = id % a.parts;
        const int nb = L % a.nblocks;
        const int mbx = L / a.nblocks;
        const int mb = a.mb_begin + mbx % a.mblocks, xi = mbx / a.mblocks;
        it.nb = nb;
        it.m0 = (long long)mb * a.mrows + h * BM;
        it.vplane = a.V + (size_t)xi * a.ksteps * a.v_step_bytes;
        it.upanel = a.U + ((size_t)xi * a.nblocks + nb) * ((size_t)a.ksteps * SB_UB);
        it.mplane = a.M + (size_t)xi * a.T * a.Cout;
        return true;
    };
    // one K step of one item -> LDS stage `buf`: VP / 8 (+1) + 3 wave instructions of 1 KiB per wave.  Rows >= T of a V
    // sub-plane lie beyond its buffer window: zeros (their outputs are never stored).
    // piece j of a stage's DMAs of this wave: j < NV the V pieces (the last one only on waves that carry it), then the UPW U pieces
    constexpr int NV = (VP + 7) / 8, NPIECE = NV + UPW;
    auto issue_piece = [&](const Item& it, int s, int buf, int j) {
        char* sb = smem + buf * STAGE;
        if (j < NV) {
            if (VP % 8 != 0 && j == NV - 1 && !vextra) return;
            const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(it.vplane + (size_t)s * a.v_step_bytes), 0, a.v_step_bytes, 0x00020000);
            const unsigned vo = (unsigned)(it.m0 * SB_ROW) + dma_lane;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (lds_void*)(sb + (wave + 8 * j) * 1024), 16, vo + j * 8192, 0, 0, RN_BF3_VLOAD_AUX);
        } else {
            const int i = j - NV;
            const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(it.upanel + (size_t)s * SB_UB), 0, SB_UB, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lds_void*)(sb + VB + (wave + 8 * i) * 1024), 16, dma_lane + i * 8192, 0, 0, RN_BF3_ULOAD_AUX);
        }
    };
    auto issue = [&](const Item& it, int s, int buf) {
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) issue_piece(it, s, buf, j);
    };
    // wait until all but the newest stage's DMAs of this wave (and, behind an item's end, its NSTORE stores) have landed, then the barrier
    auto wait_stage = [&](bool issued, bool after_store) {
        if (!issued) { BF3_WAIT_BARRIER(0); return; }
        if constexpr (F::ID == 1) {                                   // H2: 2 + 2 | 1 + 2 DMAs per stage and wave, no uneven V share
            static_assert(VP % 8 == 0, "format H2: every wave carries the same number of V pieces");
            if constexpr (WM == 4) { if (after_store) BF3_WAIT_BARRIER(36); else BF3_WAIT_BARRIER(4); }
            else                   { if (after_store) BF3_WAIT_BARRIER(19); else BF3_WAIT_BARRIER(3); }
        } else if constexpr (WM == 4) {
            if (after_store) BF3_WAIT_BARRIER(38); else BF3_WAIT_BARRIER(6);
        } else {
            if (vextra) { if (after_store) BF3_WAIT_BARRIER(21); else BF3_WAIT_BARRIER(5); }
            else        { if (after_store) BF3_WAIT_BARRIER(20); else BF3_WAIT_BARRIER(4); }
        }
    };

    if constexpr (P16) {
        // ---- format B3 on v_mfma_f32_16x16x32_bf16, the K = 32 of one instruction = the 16 channels of a K step x TWO pieces: with
        // A = [u_p | u_q] and B = [v_r | v_s] (lanes 0..31 read the first piece's plane, lanes 32..63 the second's) one MFMA is
        // u_p.v_r + u_q.v_s, so the six products are three instructions per 16 x 16 tile:
        //     [u0|u2].[v2|v0]   (the two smallest terms first)      [u0|u1].[v1|v1]      [u0|u1].[v0|v0]
        // Same LDS stages, DMA schedule and item walk as the 32 x 32 x 16 form below; per K step a wave reads 4 x 3 V fragments (held for
        // the step) + 2 per 16-channel group of U (double-buffered) = 28 | 20 ds_read_b128 for 96 | 48 MFMAs of 16 cycles.
        // Why: on random data the chip sustains 1978 TFLOP/s of 16x16x32 against 1754 of 32x32x16 (scripts/mfma_power_probe.hip,
        // profiles/r06p_mfma_power_probe.txt: half the accumulator traffic per MAC), and this stage runs at that power ceiling.
        static_assert(F::ID == 0, "paired 16x16x32 products: format B3 only");
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        constexpr int TT = 4, CT = NT2 * 2;                           // 16-row tile groups of the wave's 64 rows; 16-channel groups of its 128 | 64 channels
        // lane (i, g4) of an operand holds row i, k group g4 (8 values).  A ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15,
        // 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- i.e. rows {0-3, 12-15} of one chunk with rows {4-11} of the other; with the stage's
        // row layout (96-byte rows, chunks swapped in rows with bit 3 set) that is a 2-way bank conflict on every read (SQ_LDS_BANK_CONFLICT =
        // half of SQ_LDS_IDX_ACTIVE) unless operand row i is LDS row pi(i), pi = swap of the quads 8-11 and 12-15: then both chunk sets are
        // closed under row + 8 and the 16 lanes of a group hit 16 different bank quads.  The outputs move with it: D row 4 g4 + e is channel
        // 4 sigma(g4) + e, sigma = (0, 1, 3, 2), D column i is tile row pi(i).
        const int i16 = lane & 15, g4 = lane >> 4;
        const int r16 = i16 < 8 ? i16 : i16 ^ 4, gs = g4 < 2 ? g4 : g4 ^ 1;
        const unsigned sw = (unsigned)(((g4 & 1) ^ ((r16 >> 3) & 1)) << 4);   // 16-byte chunk of the plane (swapped in rows with bit 3 set)
        const bool second = g4 >= 2;                                   // this lane's 8 values belong to the second piece of the pair
        const unsigned vrow = (unsigned)((wm * 64 + r16) * SB_ROW) + sw, urow = (unsigned)(VB + (wn * (NT2 * 32) + r16) * SB_ROW) + sw;
        const unsigned vo_a = vrow, vo_b = vrow + 32u, vo_c = vrow + (second ? 0u : 64u);     // [v0|v0], [v1|v1], [v2|v0]
        const unsigned uo_a = urow + (second ? 32u : 0u), uo_b = urow + (second ? 64u : 0u);  // [u0|u1], [u0|u2]
        f32x4_ acc[TT][CT];
        frag va[TT], vb[TT], vc[TT], ua[2], ub[2];
        auto ldu = [&](const char* sb, int ct, int slot) {
            ua[slot] = *reinterpret_cast<const frag*>(sb + uo_a + ct * (16 * SB_ROW));
            ub[slot] = *reinterpret_cast<const frag*>(sb + uo_b + ct * (16 * SB_ROW));
        };
        Item cur, nxt;
        if (!decode(0, cur)) return;
        bool have_next = decode(1, nxt);
        issue(cur, 0, 0);
        issue(cur, 1, 1);
        wait_stage(true, false);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            va[tt] = *reinterpret_cast<const frag*>(smem + vo_a + tt * (16 * SB_ROW));
            vb[tt] = *reinterpret_cast<const frag*>(smem + vo_b + tt * (16 * SB_ROW));
            vc[tt] = *reinterpret_cast<const frag*>(smem + vo_c + tt * (16 * SB_ROW));
        }
        ldu(smem, 0, 0);
        int buf = 0;
        bool after_store = false;
        auto step = [&](int s) {
            const char* sb = smem + buf * STAGE;
            const int bn = buf == SB_NSTAGE - 1 ? 0 : buf + 1;
            const int b2 = bn == SB_NSTAGE - 1 ? 0 : bn + 1;
            const int s2 = s + 2;
            const bool in_item = s2 < a.ksteps;
            const bool issued = !(a.probe & 1) && (in_item || have_next);
            const Item& src = in_item ? cur : nxt;
            const int ss = in_item ? s2 : s2 - a.ksteps;
            // the two buffer descriptors and the row offset of the stage's source once per step (issue_piece derives them per piece: ~20 scalar
            // instructions each, between MFMA groups)
            const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src.vplane + (size_t)ss * a.v_step_bytes), 0, a.v_step_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src.upanel + (size_t)ss * SB_UB), 0, SB_UB, 0x00020000);
            const unsigned vo = (unsigned)(src.m0 * SB_ROW) + dma_lane;
            char* const sd = smem + b2 * STAGE;
            auto dma = [&](int j) {
                if (!issued || j >= NPIECE) return;
                if (j < NV) {
                    if (VP % 8 != 0 && j == NV - 1 && !vextra) return;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (lds_void*)(sd + (wave + 8 * j) * 1024), 16, vo + j * 8192, 0, 0, RN_BF3_VLOAD_AUX);
                } else {
                    const int i = j - NV;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lds_void*)(sd + VB + (wave + 8 * i) * 1024), 16, dma_lane + i * 8192, 0, 0, RN_BF3_ULOAD_AUX);
                }
            };
            const bool more = s + 1 < a.ksteps || have_next;
            const char* sn = smem + bn * STAGE;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int slot = ct & 1;
                const bool last = ct == CT - 1;
#if RN_P16_SCHED == 2
                // the group's U reads + DMA piece go out BEHIND its first four MFMAs (a wave that starts a group with ~10 scalar / memory instructions
                // leaves the matrix pipe to the other wave of the SIMD for that long: +2.3 % on the stage, profiles/r06s_*; the DMA behind the second
                // four, or the last group's wait + barrier behind its first four, measured slower)
                if (last) {
                    // every DMA piece of the stage after next is out: wait for the NEXT stage (counted), barrier, then this step's last
                    // channel group runs while the next step's first operands replace the fragments it has finished with
                    wait_stage(issued, after_store);
                    if (more) ldu(sn, 0, slot ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ub[slot], vc[tt], acc[tt][ct], 0, 0, 0);
                if (!last) {
                    ldu(sb, ct + 1, slot ^ 1);
                    if (CT == 8) dma(ct); else { dma(ct); dma(ct + 3); }
                    __builtin_amdgcn_sched_barrier(0);
                }
#else
                if (!last) ldu(sb, ct + 1, slot ^ 1);
                if (CT == 8) { if (ct < CT - 1) dma(ct); }
                else { if (ct < CT - 1) { dma(ct); dma(ct + 3); } }
                if (last) {
                    // every DMA piece of the stage after next is out: wait for the NEXT stage (counted), barrier, then this step's last
                    // channel group runs while the next step's first operands replace the fragments it has finished with
                    wait_stage(issued, after_store);
                    if (more) ldu(sn, 0, slot ^ 1);
                }
#if RN_P16_SCHED == 0
                __builtin_amdgcn_sched_barrier(0);                    // (the reads of the next group's U fragments stay AHEAD of this group's MFMAs)
#endif
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ub[slot], vc[tt], acc[tt][ct], 0, 0, 0);
#endif
                if (last && more) {
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt) vc[tt] = *reinterpret_cast<const frag*>(sn + vo_c + tt * (16 * SB_ROW));
                }
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua[slot], vb[tt], acc[tt][ct], 0, 0, 0);
                if (last && more) {
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt) vb[tt] = *reinterpret_cast<const frag*>(sn + vo_b + tt * (16 * SB_ROW));
                }
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua[slot], va[tt], acc[tt][ct], 0, 0, 0);
                if (last && more) {
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt) {
                        va[tt] = *reinterpret_cast<const frag*>(sn + vo_a + tt * (16 * SB_ROW));
                    }
                }
#if RN_P16_SCHED != 1
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            buf = bn;
            after_store = false;
        };
        for (int r = 0;; ++r) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[tt][ct] = f32x4_{0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < a.ksteps; ++s) step(s);
            // D (16 x 16) = U group (rows: channels) x V group (cols: tile rows): register e of lane (i, g4) is channel 4 sigma(g4) + e, tile row pi(i)
            if (!(a.probe & 2)) {
                const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(cur.mplane, 0, a.m_bytes, 0x00020000);
                const unsigned mo = (unsigned)(((cur.m0 + wm * 64 + r16) * a.Cout + cur.nb * SB_BN + wn * (NT2 * 32) + gs * 4) * 4);
#pragma unroll
                for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[tt][ct]), mrsrc,
                                                               mo + (unsigned)(tt * 16 * a.Cout * 4) + ct * 64, 0, RN_BF3_M_AUX);
                static_assert(NSTORE == TT * CT, "the counted waits assume this many stores per wave");
                after_store = true;
            }
            if (!have_next) break;
            cur = nxt;
            have_next = decode(r + 2, nxt);
        }
        return;
    }
    f32x16 acc[2][NT2];
    auto ldv = [&](const char* sb, int mt, frag (&v)[NP]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) v[p] = *reinterpret_cast<const frag*>(sb + vfrag[p] + mt * (32 * SB_ROW));
    };
    auto ldu = [&](const char* sb, int nt, frag (&u)[NP]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) u[p] = *reinterpret_cast<const frag*>(sb + ufrag[p] + nt * (32 * SB_ROW));
    };
    // one 32 x 32 tile and K step: the piece products of the format (B3: the six with i + j <= 2; H2: three), smallest terms first
    auto grp = [&](const frag (&v)[NP], const frag (&u)[NP], f32x16& c) {
#pragma unroll
        for (int k = 0; k < F::NPROD; ++k) c = F::mfma(u[F::PU[k]], v[F::PV[k]], c);
    };

    Item cur, nxt;
    if (!decode(0, cur)) return;
    bool have_next = decode(1, nxt);
    frag x0[NP], x1[NP], ua[NP], ub[NP];
    issue(cur, 0, 0);
    issue(cur, 1, 1);
    wait_stage(true, false);
    ldv(smem, 0, x0);
    ldu(smem, 0, ua);
    int buf = 0;                                                      // stage of the current K step
    bool after_store = false;                                         // the previous item's stores sit in the queue behind DMA(g+1)

    // One K step = 2 x NT2 (row tile, channel tile) groups of 6 MFMAs in an order in which consecutive groups share an operand,
    // so that 4 fragment sets (2 V, 2 U: 48 registers) suffice: a group's new operand is read from LDS about two groups ahead
    // of its use.  At entry v0 holds the fragments of row tile 0 and ua those of channel tile 0 (read under the previous step's
    // last group); v1 is free.  Before the last group the next stage is waited for (counted: the stage after it stays in
    // flight), the barrier says every wave has finished reading this stage, and the next step's first operands are read into
    // the two sets that have just become free -- the next step runs with the roles of v0 / v1 swapped.
    auto step = [&](int s, frag (&v0)[NP], frag (&v1)[NP]) {
        const char* sb = smem + buf * STAGE;
        const int bn = buf == SB_NSTAGE - 1 ? 0 : buf + 1;
        const int b2 = bn == SB_NSTAGE - 1 ? 0 : bn + 1;
        const int s2 = s + 2;
        // the DMAs of the stage two steps ahead go out one piece at a time behind the MFMA groups (all eight waves issuing
        // their six pieces together at the head of the step stalled the matrix pipe for the length of the issue)
        const bool in_item = s2 < a.ksteps;
        const bool issued = !(a.probe & 1) && (in_item || have_next);
        const Item& src = in_item ? cur : nxt;
        const int ss = in_item ? s2 : s2 - a.ksteps;
        const bool spread = !(a.probe & 4);
        if (issued && !spread) issue(src, ss, b2);
        auto dma = [&](int j) { if (issued && spread && j < NPIECE) issue_piece(src, ss, b2, j); };
        ldv(sb, 1, v1);
        ldu(sb, 1, ub);
        grp(v0, ua, acc[0][0]);
        dma(0);
        grp(v1, ua, acc[1][0]);
        dma(1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NT2 == 4) {
            ldu(sb, 2, ua);
            grp(v1, ub, acc[1][1]);
            dma(2);
            grp(v0, ub, acc[0][1]);
            dma(3);
            __builtin_amdgcn_sched_barrier(0);
            ldu(sb, 3, ub);
            grp(v0, ua, acc[0][2]);
            dma(4);
            grp(v1, ua, acc[1][2]);
            dma(5);
        } else {
            dma(2); dma(3); dma(4);
        }
        grp(v1, ub, acc[1][NT2 - 1]);
        wait_stage(issued, after_store);
        if (s + 1 < a.ksteps || have_next) {
            const char* sn = smem + bn * STAGE;
            ldv(sn, 0, v1);
            ldu(sn, 0, ua);
        }
        grp(v0, ub, acc[0][NT2 - 1]);
        __builtin_amdgcn_sched_barrier(0);
        buf = bn;
        after_store = false;
    };

    for (int r = 0;; ++r) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        for (int s = 0; s < a.ksteps; s += 2) {
            step(s, x0, x1);
            step(s + 1, x1, x0);
        }
        // D (32 x 32) = U-tile (rows: channels) x V-tile (cols: tile rows): register r of lane (l32, hb) is channel
        // (r & 3) + 8*(r >> 2) + 4*hb of the 32-channel tile, tile row l32 -> four 16-byte stores per MFMA tile
        if (!(a.probe & 2)) {
            if constexpr (F::ID == 1) {                               // H2: back to the scale of the fp32 operands (powers of two: exact)
                const float sc = h2_scale(__builtin_bit_cast(float, *a.amax_v), a.bound_v) * h2_scale(__builtin_bit_cast(float, *a.amax_u), a.bound_u);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT2; ++nt) acc[mt][nt] *= sc;
            }
            const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(cur.mplane, 0, a.m_bytes, 0x00020000);
            const unsigned mo = (unsigned)(((cur.m0 + wm * 64 + l32) * a.Cout + cur.nb * SB_BN + wn * (NT2 * 32) + hb * 4) * 4);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 o = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), mrsrc,
                                                               mo + (unsigned)(mt * 32 * a.Cout * 4) + nt * 128 + g * 32, 0, RN_BF3_M_AUX);
                    }
            static_assert(NSTORE == 2 * NT2 * 4, "the counted waits assume this many stores per wave");
            after_store = true;
        }
        if (!have_next) break;
        cur = nxt;
        have_next = decode(r + 2, nxt);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// 2b.  The same GEMM items with FOUR waves per workgroup (one per SIMD): wave tile 128 x 128 = 4 x 4 MFMA tiles, 256 accumulators
// (the whole accumulator half of a 512-register wave), 2 x 2 waves over the 256 x 256 block.  Why: per K step the
// 8-wave kernel reads 18 fragment sets for 48 MFMAs per wave, this one 24 for 96 -- a third fewer LDS bytes per MFMA, which is the one lever
// on the POWER the stage runs into (MFMA busy 0.74 at 1.75 GHz) rather than on a stall count -- and a step's barrier is met by four waves
// instead of eight.  With one wave per SIMD nothing hides a wave's own latencies but its own instruction stream: the V fragments of a
// step are all read one step ahead (double-buffered: 2 x 48 registers), the U fragments one channel tile ahead, the stage-after-next's DMA
// pieces go out between the MFMA groups.  Whole 256-row items only (ragged blocks and partial rounds stay on the 8-wave kernel's half items).
// Selected per launch by RN_WINO_BF3_W4 (default: see gemm_split_planes).
template <class F, int TAG>
__global__ __launch_bounds__(256, 1)
void wino_gemm_bf3_w4_kernel(const Bf3GemmArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename F::frag frag;
    constexpr int NP = F::NP, SB_ROW = F::ROW, SB_UB = SB_BN * SB_ROW;
    constexpr int BM = 256, VB = BM * SB_ROW, STAGE = VB + SB_UB;     // B3: 24 + 24 KiB, H2: 16 + 16
    constexpr int NPIECE = STAGE / 4096;                               // DMA pieces (1 KiB) per wave and stage: 12 | 8
    constexpr int NVP = VB / 4096;                                     // ... of which V: 6 | 4
    constexpr int NSTORE = 4 * 4 * 4;                                  // 16-byte stores of a wave's epilogue
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hb = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    unsigned vfrag[NP], ufrag[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        vfrag[p] = (unsigned)((wm * 128 + l32) * SB_ROW) + F::chunk(l32, p, hb);
        ufrag[p] = (unsigned)(VB + (wn * 128 + l32) * SB_ROW) + F::chunk(l32, p, hb);
    }
    const unsigned dma_lane = (unsigned)(wave * 1024 + lane * 16);

    struct Item { const char* vplane; const char* upanel; float* mplane; long long m0; int nb; };
    const int rounds_total = (a.item_end - a.item_begin) * a.parts;
    const int perm = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    auto decode = [&](int r, Item& it) -> bool {
        const int id = r * (int)gridDim.x + perm;
        if (id >= rounds_total) return false;
        const int L = a.item_begin + id / a.parts, h = id % a.parts;
        const int nb = L % a.nblocks;
        const int mbx = L / a.nblocks;
        const int mb = a.mb_begin + mbx % a.mblocks, xi = mbx / a.mblocks;
        it.nb = nb;
        it.m0 = (long long)mb * a.mrows + h * BM;
        it.vplane = a.V + (size_t)xi * a.ksteps * a.v_step_bytes;
        it.upanel = a.U + ((size_t)xi * a.nblocks + nb) * ((size_t)a.ksteps * SB_UB);
        it.mplane = a.M + (size_t)xi * a.T * a.Cout;
        return true;
    };
    // piece j of a stage's DMAs of this wave: j < NVP the V pieces (rows wave + 4 j of the 1-KiB grid), then the U pieces.  `oob` = 2^31
    // turns the piece into a zero fill of its LDS slot (offset beyond the buffer window): the step issues its pieces UNCONDITIONALLY -- a
    // branch around each of them cuts the step into basic blocks, and hipcc then drains the LDS counter at every join instead of where a
    // fragment is first used; with one wave per SIMD that wait is on the critical path.  (The zero-filled stage is never read: there is no
    // next item.)
    auto issue_piece = [&](const Item& it, int s, int buf, int j, unsigned oob) {
        char* sb = smem + buf * STAGE;
        if (j < NVP) {
            const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(it.vplane + (size_t)s * a.v_step_bytes), 0, a.v_step_bytes, 0x00020000);
            const unsigned vo = (unsigned)(it.m0 * SB_ROW) + dma_lane;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (lds_void*)(sb + (wave + 4 * j) * 1024), 16, (vo + j * 4096) | oob, 0, 0, RN_BF3_VLOAD_AUX);
        } else {
            const int i = j - NVP;
            const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(it.upanel + (size_t)s * SB_UB), 0, SB_UB, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lds_void*)(sb + VB + (wave + 4 * i) * 1024), 16, (dma_lane + i * 4096) | oob, 0, 0, RN_BF3_ULOAD_AUX);
        }
    };
    auto issue = [&](const Item& it, int s, int buf) {
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) issue_piece(it, s, buf, j, 0u);
    };
    // all but the newest stage's DMAs of this wave (always NPIECE: see above) have landed (behind an item's end its 64 stores sit in front of
    // them: the counter holds 63 at most, so the first step of an item also waits for the oldest of those stores), then the barrier
    auto wait_stage = [&](bool after_store) {
        if (after_store) { BF3_WAIT_BARRIER(63); return; }
        if constexpr (NPIECE == 12) BF3_WAIT_BARRIER(12); else BF3_WAIT_BARRIER(8);
    };
    static_assert(NPIECE == 12 || NPIECE == 8, "counted waits");

    f32x16 acc[4][4];
    auto ldv = [&](const char* sb, int mt, frag (&v)[NP]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) v[p] = *reinterpret_cast<const frag*>(sb + vfrag[p] + mt * (32 * SB_ROW));
    };
    auto ldu = [&](const char* sb, int nt, frag (&u)[NP]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) u[p] = *reinterpret_cast<const frag*>(sb + ufrag[p] + nt * (32 * SB_ROW));
    };
    auto grp = [&](const frag (&v)[NP], const frag (&u)[NP], f32x16& c) {
#pragma unroll
        for (int k = 0; k < F::NPROD; ++k) c = F::mfma(u[F::PU[k]], v[F::PV[k]], c);
    };

    Item cur, nxt;
    if (!decode(0, cur)) return;
    bool have_next = decode(1, nxt);
    frag va[4][NP], vb[4][NP], ua[NP], ub[NP];
    issue(cur, 0, 0);
    issue(cur, 1, 1);
    wait_stage(false);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) ldv(smem, mt, va[mt]);
    ldu(smem, 0, ua);
    int buf = 0;
    bool after_store = false;

    // One K step: 16 (row tile, channel tile) groups, channel tile by channel tile; the V fragments v0 of all four row tiles were read during
    // the previous step, the next channel tile's U fragments are read one tile ahead; before the last channel tile the next stage is waited
    // for and the barrier taken (nobody reads this stage any more), and the next step's V fragments (v1) and first U fragments are read
    // under the last 24 MFMAs.
    auto step = [&](int s, frag (&v0)[4][NP], frag (&v1)[4][NP]) {
        const char* sb = smem + buf * STAGE;
        const int bn = buf == SB_NSTAGE - 1 ? 0 : buf + 1;
        const int b2 = bn == SB_NSTAGE - 1 ? 0 : bn + 1;
        const int s2 = s + 2;
        const bool in_item = s2 < a.ksteps;
        const unsigned oob = (!(a.probe & 1) && (in_item || have_next)) ? 0u : 0x80000000u;
        const Item& src = in_item ? cur : nxt;
        const int ss = in_item ? s2 : s2 - a.ksteps;
        auto dma = [&](int j) { if (j < NPIECE) issue_piece(src, ss, b2, j, oob); };
        ldu(sb, 1, ub);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { grp(v0[mt], ua, acc[mt][0]); dma(mt); }
        __builtin_amdgcn_sched_barrier(0);
        ldu(sb, 2, ua);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { grp(v0[mt], ub, acc[mt][1]); dma(4 + mt); }
        __builtin_amdgcn_sched_barrier(0);
        ldu(sb, 3, ub);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { grp(v0[mt], ua, acc[mt][2]); dma(8 + mt); }
        __builtin_amdgcn_sched_barrier(0);
        wait_stage(after_store);
        {   // (unconditional: behind the last step of the last item the fragments read here are never used)
            const char* sn = smem + bn * STAGE;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) ldv(sn, mt, v1[mt]);
            ldu(sn, 0, ua);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) grp(v0[mt], ub, acc[mt][3]);
        __builtin_amdgcn_sched_barrier(0);
        buf = bn;
        after_store = false;
    };

    for (int r = 0;; ++r) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        for (int s = 0; s < a.ksteps; s += 2) {
            step(s, va, vb);
            step(s + 1, vb, va);
        }
        if (!(a.probe & 2)) {
            // H2: back to the scale of the fp32 operands (powers of two: exact) -- applied to the four values of a store, not to the 256
            // accumulators at once (that would pull the whole accumulator file through the vector half)
            float sc = 1.f;
            if constexpr (F::ID == 1)
                sc = h2_scale(__builtin_bit_cast(float, *a.amax_v), a.bound_v) * h2_scale(__builtin_bit_cast(float, *a.amax_u), a.bound_u);
            const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(cur.mplane, 0, a.m_bytes, 0x00020000);
            const unsigned mo = (unsigned)(((cur.m0 + wm * 128 + l32) * a.Cout + cur.nb * SB_BN + wn * 128 + hb * 4) * 4);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 o = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                        if constexpr (F::ID == 1) o *= sc;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), mrsrc,
                                                               mo + (unsigned)(mt * 32 * a.Cout * 4) + nt * 128 + g * 32, 0, RN_BF3_M_AUX);
                    }
            static_assert(NSTORE == 64, "the counted waits assume this many stores per wave");
            after_store = true;
        }
        if (!have_next) break;
        cur = nxt;
        have_next = decode(r + 2, nxt);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the zero-fill pieces of the last steps must not outlive the workgroup's LDS
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// `scheme` of the entry points below = Winograd scheme | operand format << 8 (RN_SPLIT_FMT_H2 = 0x100: two fp16 pieces of the scaled
// value, three products; 0: three bf16 pieces, six products).
namespace {
inline int fmt_of(int scheme) { return (scheme >> 8) & 0xff; }
inline int sch_of(int scheme) { return scheme & 0xff; }
inline size_t h2_u_data(int sch, int Cin, int Cout) { return (size_t)rn_split_scheme_nxi(sch) * Cin * Cout * 4; }
inline size_t h2_v_data(int sch, long long T, int Cin) { return ((size_t)rn_split_scheme_nxi(sch) * T * Cin * 4 + 255) / 256 * 256; }
inline float h2_bound_v(int sch) { return sch == RN_WINO_F11 ? 1.f : sch == RN_WINO_F43 ? H2Bound<WinoF43>::bt() : sch == RN_WINO_F44 ? H2Bound<WinoF44>::bt() : H2Bound<WinoF63>::bt(); }
inline float h2_bound_u(int sch) { return sch == RN_WINO_F11 ? 1.f : sch == RN_WINO_F43 ? H2Bound<WinoF43>::g() : sch == RN_WINO_F44 ? H2Bound<WinoF44>::g() : H2Bound<WinoF63>::g(); }

// *out = bit pattern of max |x| over n floats (n % 4 == 0)
int launch_absmax(const float* x, size_t n, unsigned* out, hipStream_t st) { return rn_launch_absmax(x, n, out, st); }
}  // namespace

int rn_launch_word(unsigned* dst, const unsigned* src, hipStream_t st)
{
    hipLaunchKernelGGL(word_kernel, dim3(1), dim3(1), 0, st, dst, src);
    return rn_check_launch("word");
}

int rn_launch_absmax(const float* x, size_t n, unsigned* out, hipStream_t st)
{
    if (n % 4 != 0) return rn_set_error(RN_E_INVALID, "absmax: %zu floats", n);
    { const int rc = rn_launch_word(out, nullptr, st); if (rc != RN_OK) return rc; }
    const size_t n4 = n / 4;
    const unsigned blocks = (unsigned)(n4 / 1024 + 1 < 1024 ? n4 / 1024 + 1 : 1024);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, st, x, n4, out);
    return rn_check_launch("absmax");
}


int rn_split_scheme_nxi(int scheme) { return scheme == RN_WINO_F11 ? 1 : rn_wino_scheme_nxi(scheme); }
int rn_split_scheme_m(int scheme) { return scheme == RN_WINO_F11 ? 1 : rn_wino_scheme_m(scheme); }

bool rn_wino_bf3_supported(int scheme, int Cin, int Cout)
{
    static const bool off = getenv("RN_NO_WINOGRAD_BF3") != nullptr;
    static const bool off11 = getenv("RN_NO_SPLIT_1X1") != nullptr;
    if (off || fmt_of(scheme) > 1) return false;
    if (sch_of(scheme) == RN_WINO_F11) return !off11 && Cin >= 32 && Cin % 32 == 0 && Cout >= 256 && Cout % 256 == 0;
    return rn_wino43_supported(sch_of(scheme), Cin, Cout);
}

// B3: 6 bytes per element.  H2: 4 bytes per element + a 256-byte tail whose first word is max|w| (bit pattern) of the filter.
size_t rn_wino_bf3_packed_bytes(int scheme, int Cin, int Cout)
{
    if (fmt_of(scheme) == 1) return h2_u_data(sch_of(scheme), Cin, Cout) + 256;
    return (size_t)rn_split_scheme_nxi(scheme) * Cin * Cout * 6;
}

// workspace: Vs (B3: nxi * T * Cin * 6 bytes, rounded up to 256; H2: ... * 4 + a 256-byte tail holding max|x|) followed by M (nxi * T * Cout floats)
size_t rn_wino_bf3_v_bytes(int scheme, long long T, int Cin)
{
    if (fmt_of(scheme) == 1) return h2_v_data(sch_of(scheme), T, Cin) + 256;
    return ((size_t)rn_split_scheme_nxi(scheme) * T * Cin * 6 + 255) / 256 * 256;
}

size_t rn_wino_bf3_workspace_bytes(int scheme, int B, int H, int W, int Cin, int Cout)
{
    const int m = rn_split_scheme_m(sch_of(scheme));
    if (m == 0) return 0;
    const long long T = (long long)B * ((H + m - 1) / m) * ((W + m - 1) / m);
    return rn_wino_bf3_v_bytes(scheme, T, Cin) + (size_t)rn_split_scheme_nxi(sch_of(scheme)) * T * Cout * 4;
}

int rn_launch_wino_pack_bf3(int scheme, const float* w_tf, void* us, int Cin, int Cout, int transposed, hipStream_t st)
{
    if (!rn_wino_bf3_supported(scheme, Cin, Cout)) return rn_set_error(RN_E_UNSUPPORTED, "wino_pack_bf3: scheme=%d Cin=%d Cout=%d", scheme, Cin, Cout);
    const int fmt = fmt_of(scheme);
    scheme = sch_of(scheme);
    const unsigned nbw = (unsigned)((Cout / PK_CO) * (Cin / 16));
    char* u = static_cast<char*>(us);
    if (fmt == 1) {
        unsigned* amax = reinterpret_cast<unsigned*>(u + h2_u_data(scheme, Cin, Cout));
        const int R = scheme == RN_WINO_F11 ? 1 : scheme == RN_WINO_F44 ? 4 : 3;
        const int rc = launch_absmax(w_tf, (size_t)R * R * Cin * Cout, amax, st);
        if (rc != RN_OK) return rc;
        if (scheme == RN_WINO_F11) hipLaunchKernelGGL(wino_pack_h2_kernel<WinoF11>, dim3(nbw), dim3(256), 0, st, w_tf, u, amax, Cin, Cout, transposed);
        else if (scheme == RN_WINO_F43) hipLaunchKernelGGL(wino_pack_h2_kernel<WinoF43>, dim3(nbw), dim3(256), 0, st, w_tf, u, amax, Cin, Cout, transposed);
        else if (scheme == RN_WINO_F44) hipLaunchKernelGGL(wino_pack_h2_kernel<WinoF44>, dim3(nbw), dim3(256), 0, st, w_tf, u, amax, Cin, Cout, transposed);
        else hipLaunchKernelGGL(wino_pack_h2_kernel<WinoF63>, dim3(nbw), dim3(256), 0, st, w_tf, u, amax, Cin, Cout, transposed);
        return rn_check_launch("wino_pack_h2");
    }
    if (scheme == RN_WINO_F11) hipLaunchKernelGGL(wino_pack_bf3_kernel<WinoF11>, dim3(nbw), dim3(256), 0, st, w_tf, u, Cin, Cout, transposed);
    else if (scheme == RN_WINO_F43) hipLaunchKernelGGL(wino_pack_bf3_kernel<WinoF43>, dim3(nbw), dim3(256), 0, st, w_tf, u, Cin, Cout, transposed);
    else if (scheme == RN_WINO_F44) hipLaunchKernelGGL(wino_pack_bf3_kernel<WinoF44>, dim3(nbw), dim3(256), 0, st, w_tf, u, Cin, Cout, transposed);
    else hipLaunchKernelGGL(wino_pack_bf3_kernel<WinoF63>, dim3(nbw), dim3(256), 0, st, w_tf, u, Cin, Cout, transposed);
    return rn_check_launch("wino_pack_bf3");
}

int rn_launch_wino_input_bf3(int scheme, const float* x, void* Vs, int B, int H, int W, int C, int pad_lo, hipStream_t st)
{
    return rn_launch_wino_input_bf3_ex(scheme, x, Vs, B, H, W, C, pad_lo, nullptr, st);
}

// amax_x (format H2 only, may be null): a device word that already holds the bit pattern of max|x| (or of an upper bound of it), e.g.
// from the launch that produced x -- copied into the tail of Vs instead of a pass over x
int rn_launch_wino_input_bf3_ex(int scheme, const float* x, void* Vs, int B, int H, int W, int C, int pad_lo, const unsigned* amax_x, hipStream_t st)
{
    const int fmt = fmt_of(scheme);
    scheme = sch_of(scheme);
    const int m = rn_split_scheme_m(scheme);
    if (m == 0 || C % 16 != 0) return rn_set_error(RN_E_INVALID, "wino_input_bf3: scheme %d, C %d", scheme, C);
    const int th = (H + m - 1) / m, tw = (W + m - 1) / m;
    const long long T = (long long)B * th * tw;
    const unsigned ncb = (unsigned)((C + 63) / 64);
    const unsigned long long n = (unsigned long long)((T + IB_TILES - 1) / IB_TILES) * ncb;
    if (n > 0x7fffff00ULL) return rn_set_error(RN_E_UNSUPPORTED, "wino_input_bf3: grid too large");
    const unsigned nwg = (unsigned)n, nblk8 = (unsigned)((n + 7) / 8 * 8);
    char* v = static_cast<char*>(Vs);
    if (fmt == 1) {
        unsigned* amax = reinterpret_cast<unsigned*>(v + h2_v_data(scheme, T, C));
        if (amax_x) {
            const int rc = rn_launch_word(amax, amax_x, st);
            if (rc != RN_OK) return rc;
        } else {
            const int rc = launch_absmax(x, (size_t)B * H * W * C, amax, st);
            if (rc != RN_OK) return rc;
        }
        if (scheme == RN_WINO_F11)
            hipLaunchKernelGGL((wino_input_h2_kernel<WinoF11>), dim3(nblk8), dim3(256), 0, st, x, v, amax, H, W, C, th, tw, T, ncb, nwg, nblk8, pad_lo);
        else if (scheme == RN_WINO_F43)
            hipLaunchKernelGGL((wino_input_h2_kernel<WinoF43>), dim3(nblk8), dim3(256), 0, st, x, v, amax, H, W, C, th, tw, T, ncb, nwg, nblk8, pad_lo);
        else if (scheme == RN_WINO_F44)
            hipLaunchKernelGGL((wino_input_h2_kernel<WinoF44>), dim3(nblk8), dim3(256), 0, st, x, v, amax, H, W, C, th, tw, T, ncb, nwg, nblk8, pad_lo);
        else
            hipLaunchKernelGGL((wino_input_h2_kernel<WinoF63>), dim3(nblk8), dim3(256), 0, st, x, v, amax, H, W, C, th, tw, T, ncb, nwg, nblk8, pad_lo);
        return rn_check_launch("wino_input_h2");
    }
    if (scheme == RN_WINO_F11)
        hipLaunchKernelGGL((wino_input_bf3_kernel<WinoF11>), dim3(nblk8), dim3(256), 0, st, x, v, H, W, C, th, tw, T, ncb, nwg, nblk8, pad_lo);
    else if (scheme == RN_WINO_F43)
        hipLaunchKernelGGL((wino_input_bf3_kernel<WinoF43>), dim3(nblk8), dim3(256), 0, st, x, v, H, W, C, th, tw, T, ncb, nwg, nblk8, pad_lo);
    else if (scheme == RN_WINO_F44)
        hipLaunchKernelGGL((wino_input_bf3_kernel<WinoF44>), dim3(nblk8), dim3(256), 0, st, x, v, H, W, C, th, tw, T, ncb, nwg, nblk8, pad_lo);
    else
        hipLaunchKernelGGL((wino_input_bf3_kernel<WinoF63>), dim3(nblk8), dim3(256), 0, st, x, v, H, W, C, th, tw, T, ncb, nwg, nblk8, pad_lo);
    return rn_check_launch("wino_input_bf3");
}

// items [begin, end) of the block range the args name, `parts` BM-row parts of each (0: all 4 / WM of them)
// RN_WINO_BF3_P16: 1 (default) = format B3 on v_mfma_f32_16x16x32_bf16 with paired pieces (see the kernel), 0 = on v_mfma_f32_32x32x16_bf16
// (rounds 4-6; kept for the A/B of profiles/r06p_* and as the reference of the four-wave variant's bit-identity test)
static bool p16_mode() { static const bool m = getenv("RN_WINO_BF3_P16") ? atoi(getenv("RN_WINO_BF3_P16")) != 0 : true; return m; }

template <class F, int WM, int TAG>
static int wino_gemm_bf3_launch_t(Bf3GemmArgs a, int begin, int end, int parts, hipStream_t st)
{
    a.item_begin = begin; a.item_end = end; a.parts = parts > 0 ? parts : 4 / WM;
    const size_t lds = (size_t)SB_NSTAGE * (WM * 64 * F::ROW + SB_BN * F::ROW);
    auto kern = wino_gemm_bf3_kernel<F, WM, TAG>;
    if constexpr (F::ID == 0) { if (p16_mode()) kern = wino_gemm_bf3_kernel<F, WM, TAG, true>; }
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc_ != RN_OK) return rc_; }
    const int n = (end - begin) * a.parts;
    // RN_WINO_BF3_GRID (a multiple of 8, <= 256; measurement): workgroups = CUs the persistent kernel occupies (one workgroup per CU)
    static const unsigned gmax = getenv("RN_WINO_BF3_GRID") ? (unsigned)atoi(getenv("RN_WINO_BF3_GRID")) / 8 * 8 : 256u;
    const unsigned cap = gmax >= 8 && gmax <= 256 ? gmax : 256u;
    hipLaunchKernelGGL(kern, dim3((unsigned)n < cap ? (unsigned)((n + 7) / 8 * 8) : cap), dim3(512), lds, st, a);
    return rn_check_launch("wino_gemm_bf3");
}

template <class F, int WM>
static int wino_gemm_bf3_launch_w(int tag, const Bf3GemmArgs& a, int begin, int end, int parts, hipStream_t st)
{
    switch (tag) {
    case 0: return wino_gemm_bf3_launch_t<F, WM, 0>(a, begin, end, parts, st);
    case 1: return wino_gemm_bf3_launch_t<F, WM, 1>(a, begin, end, parts, st);
    case 2: return wino_gemm_bf3_launch_t<F, WM, 2>(a, begin, end, parts, st);
    case 4: return wino_gemm_bf3_launch_t<F, WM, 4>(a, begin, end, parts, st);
    case 5: return wino_gemm_bf3_launch_t<F, WM, 5>(a, begin, end, parts, st);
    default: return wino_gemm_bf3_launch_t<F, WM, 3>(a, begin, end, parts, st);
    }
}

template <class F, int TAG>
static int wino_gemm_bf3_w4_launch_t(Bf3GemmArgs a, int begin, int end, hipStream_t st)
{
    a.item_begin = begin; a.item_end = end; a.parts = 1;
    const size_t lds = (size_t)SB_NSTAGE * (256 * F::ROW + SB_BN * F::ROW);
    auto kern = wino_gemm_bf3_w4_kernel<F, TAG>;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc_ != RN_OK) return rc_; }
    const int n = end - begin;
    hipLaunchKernelGGL(kern, dim3(n < 256 ? (unsigned)((n + 7) / 8 * 8) : 256u), dim3(256), lds, st, a);
    return rn_check_launch("wino_gemm_bf3_w4");
}

template <class F>
static int wino_gemm_bf3_w4_launch(int tag, const Bf3GemmArgs& a, int begin, int end, hipStream_t st)
{
    switch (tag) {
    case 0: return wino_gemm_bf3_w4_launch_t<F, 0>(a, begin, end, st);
    case 1: return wino_gemm_bf3_w4_launch_t<F, 1>(a, begin, end, st);
    case 2: return wino_gemm_bf3_w4_launch_t<F, 2>(a, begin, end, st);
    case 4: return wino_gemm_bf3_w4_launch_t<F, 4>(a, begin, end, st);
    case 5: return wino_gemm_bf3_w4_launch_t<F, 5>(a, begin, end, st);
    default: return wino_gemm_bf3_w4_launch_t<F, 3>(a, begin, end, st);
    }
}

// RN_WINO_BF3_W4: 1 = whole 256-row items on the four-wave kernel (128 x 128 wave tiles), 0 = on the eight-wave kernel
static int w4_mode() { static const int m = getenv("RN_WINO_BF3_W4") ? atoi(getenv("RN_WINO_BF3_W4")) : 0; return m; }

static int wino_gemm_bf3_launch(int fmt, int wm, int tag, const Bf3GemmArgs& a, int begin, int end, int parts, hipStream_t st)
{
    // (format B3 only: the H2 instance of the four-wave kernel spills -- 256 + 11 vector registers -- and stays on the eight-wave kernel)
    if (fmt == 0 && wm == 4 && parts == 0 && w4_mode() && a.ksteps % 2 == 0) return wino_gemm_bf3_w4_launch<FmtB3>(tag, a, begin, end, st);
    if (fmt == 1) return wm == 4 ? wino_gemm_bf3_launch_w<FmtH2, 4>(tag, a, begin, end, parts, st) : wino_gemm_bf3_launch_w<FmtH2, 2>(tag, a, begin, end, parts, st);
    return wm == 4 ? wino_gemm_bf3_launch_w<FmtB3, 4>(tag, a, begin, end, parts, st) : wino_gemm_bf3_launch_w<FmtB3, 2>(tag, a, begin, end, parts, st);
}

// nxi independent GEMMs  M[p] (T x Cout) = V[p] (T x Cin) . U[p] (Cin x Cout) on split operands:
//   V [nxi][Cin/16][T][3][16], U [nxi][Cout/256][Cin/16][256][3][16], M [nxi][T][Cout] fp32; Cin % 16 == 0, Cout % 256 == 0.
// The forward path calls it with (tiles, input channels, output channels); the filter gradient (conv_wino_bf3_wgrad.hip) with
// (input channels, tiles of one K split, output channels) and nxi = planes x K splits.
// fmt 0: B3 (rows of 96 bytes); fmt 1: H2 (rows of 64 bytes; amax_v / amax_u = device words holding the bit patterns of max|x| of the
// two untransformed tensors, bound_v / bound_u the factors by which the transforms can grow them: the kernel derives the two scales).
static int gemm_split_planes(int fmt, int nxi, int tag, const void* Vs, const void* us, float* M, long long T, int Cin, int Cout,
                             const unsigned* amax_v, const unsigned* amax_u, float bound_v, float bound_u, hipStream_t st)
{
    const int SB_ROW = fmt == 1 ? FmtH2::ROW : FmtB3::ROW;
    if (nxi < 1 || Cin < 16 || Cin % 16 != 0 || Cout < SB_BN || Cout % SB_BN != 0)
        return rn_set_error(RN_E_UNSUPPORTED, "gemm_bf3: planes=%d K=%d N=%d", nxi, Cin, Cout);
    if (T < 1 || (T + SB_BM) * (long long)Cout * 4 >= 0xffffff00LL || T * (long long)Cout * 4 >= 0x7fffff00LL || (T + SB_BM) * (long long)SB_ROW >= 0x7fffff00LL)
        return rn_set_error(RN_E_UNSUPPORTED, "wino_gemm_bf3: a transform plane must stay below the 2 GiB buffer window");
    Bf3GemmArgs a;
    a.V = static_cast<const char*>(Vs); a.U = static_cast<const char*>(us); a.M = M; a.T = T; a.Cin = Cin; a.Cout = Cout;
    a.nblocks = Cout / SB_BN; a.ksteps = Cin / 16;
    a.v_step_bytes = (unsigned)(T * SB_ROW); a.m_bytes = (unsigned)(T * Cout * 4);
    { static const int probe = getenv("RN_WINO_BF3_PROBE") ? atoi(getenv("RN_WINO_BF3_PROBE")) : 0; a.probe = probe; }
    a.amax_v = amax_v; a.amax_u = amax_u; a.bound_v = bound_v; a.bound_u = bound_u;
    static const bool notail = getenv("RN_WINO_BF3_NOTAIL") != nullptr;
    const int full = (int)(T / SB_BM), ragged = (int)(T % SB_BM);    // whole 256-row blocks; rows of the last, partial one
    if ((long long)nxi * (full + 1) * a.nblocks * 2 > 0x3fffffff) return rn_set_error(RN_E_UNSUPPORTED, "wino_gemm_bf3: too many items");
    a.mrows = SB_BM;
    // One workgroup per CU takes items id, id + 256, ...  Costs below are in sixteenths of a round of whole (256-row) items.  A round of 128-row
    // half items costs 9, not 8 -- a half item loads the same 256-channel U panel for half the rows -- and every further launch about 4 (drain,
    // launch, pipeline fill), fitted on the res2 shape at batch 2 and 3 (profiles/r06l_halfcost.txt: batch 2 as one round of whole items 7.98 ms
    // per step against 8.29 as two rounds of half items; batch 3 as three rounds of half items in one launch 12.17 against a round of whole items
    // + a launch of half items 12.47).  RN_WINO_BF3_HALF_COST overrides the 9 (measurement).
    static const int HC = getenv("RN_WINO_BF3_HALF_COST") ? atoi(getenv("RN_WINO_BF3_HALF_COST")) : 9;
    constexpr int WC = 16, LC = 4;
    auto tail_cost = [](int rem) { if (rem == 0) return 0; const int half = (2 * rem + 255) / 256; return half < 2 ? half * HC : WC; };
    // the ragged block's rows as whole items or as 1 .. 2 half items per (xi, n-block)
    auto ragged_plan = [&](int& wm, int& parts) {
        const int nit = nxi * a.nblocks;
        wm = 4; parts = 1;
        int best = (nit + 255) / 256 * WC;
        const int hp = (ragged + 127) / 128, hc = (nit * hp + 255) / 256 * HC;
        if (!notail && hc < best) { best = hc; wm = 2; parts = hp; }
        return best;
    };
    // Few tiles (one GPU's share of a strongly scaled batch): below two blocks walk ALL of T in 128-row items, one launch,
    // several items per CU back to back, whenever that costs no more than whole blocks + a ragged tail.
    if (!notail && full <= 1) {
        const int nfull = nxi * full * a.nblocks;
        const int tc = tail_cost(nfull % 256);                               // (a tail at whole-item cost stays in the main launch)
        int cost_split = nfull / 256 * WC + tc, launches = (nfull >= 256 || tc == WC) + (tc != 0 && tc < WC);
        if (ragged > 0) { int wm, parts; cost_split += ragged_plan(wm, parts); ++launches; }
        cost_split += LC * (launches - 1);
        if (ragged > 0 && full == 1) {                                       // ... or the ragged block as one more block of whole items (the plan below)
            const int nm = nxi * 2 * a.nblocks, tcm = tail_cost(nm % 256);
            const int merged = nm / 256 * WC + tcm + (tcm != 0 && tcm < WC && nm >= 256 ? LC : 0);
            if (merged < cost_split) cost_split = merged;                    // (batch 4 on the res2 shape: two rounds of whole items 13.2 ms per step against four rounds of half items 14.1)
        }
        const long long items = (long long)nxi * a.nblocks * ((T + 127) / 128);
        const int ucost = (int)((items + 255) / 256) * HC;
        if (ucost <= cost_split) {
            a.mb_begin = 0; a.mrows = 128; a.mblocks = (int)((T + 127) / 128);
            return wino_gemm_bf3_launch(fmt, 2, tag, a, 0, nxi * a.mblocks * a.nblocks, 1, st);
        }
    }
    // the whole blocks in one launch of whole items (+ a launch of half items for a last round that is at most half full); the ragged block
    // either as a launch of its own (ragged_plan) or -- when that is cheaper by the costs above -- as one more block of the main launch
    // (RN_WINO_BF3_NOMERGE: never): the 512-channel layers at batch 24 become 6 full rounds instead of 5 + two half-empty launches.
    auto main_cost = [&](int nitems) {
        const int rem = nitems % 256, tc = tail_cost(rem);
        return nitems / 256 * WC + tc + (tc != 0 && tc < WC && nitems >= 256 ? LC : 0);
    };
    static const bool nomerge = getenv("RN_WINO_BF3_NOMERGE") != nullptr;
    int blocks = full, left = ragged;
    if (ragged > 0 && !notail && !nomerge) {
        int wm, parts;
        const int separate = main_cost(nxi * full * a.nblocks) + ragged_plan(wm, parts) + (full > 0 ? LC : 0);
        if (main_cost(nxi * (full + 1) * a.nblocks) < separate) { blocks = full + 1; left = 0; }
    }
    if (blocks > 0) {
        a.mb_begin = 0; a.mblocks = blocks;
        const int nitems = nxi * blocks * a.nblocks;
        const int rem = nitems % 256;
        const bool half_tail = !notail && rem != 0 && tail_cost(rem) < WC;       // the last, partial round as half items: half a round
        const int tail = half_tail ? rem : 0;
        if (nitems - tail > 0) {
            const int rc = wino_gemm_bf3_launch(fmt, 4, tag, a, 0, nitems - tail, 0, st);
            if (rc != RN_OK) return rc;
        }
        if (tail > 0) {
            const int rc = wino_gemm_bf3_launch(fmt, 2, tag, a, nitems - tail, nitems, 0, st);
            if (rc != RN_OK) return rc;
        }
    }
    if (left > 0) {
        a.mb_begin = full; a.mblocks = 1;
        int wm, parts;
        ragged_plan(wm, parts);
        return wino_gemm_bf3_launch(fmt, wm, tag, a, 0, nxi * a.nblocks, parts, st);
    }
    return RN_OK;
}

int rn_launch_gemm_bf3_planes(int nxi, int tag, const void* Vs, const void* us, float* M, long long T, int Cin, int Cout, hipStream_t st)
{
    return gemm_split_planes(0, nxi, tag, Vs, us, M, T, Cin, Cout, nullptr, nullptr, 1.f, 1.f, st);
}

int rn_launch_wino_gemm_bf3(int scheme, const void* Vs, const void* us, float* M, long long T, int Cin, int Cout, hipStream_t st)
{
    if (!rn_wino_bf3_supported(scheme, Cin, Cout)) return rn_set_error(RN_E_UNSUPPORTED, "wino_gemm_bf3: scheme=%d Cin=%d Cout=%d", scheme, Cin, Cout);
    const int fmt = fmt_of(scheme);
    scheme = sch_of(scheme);
    const int tag = scheme == RN_WINO_F11 ? 5 : scheme == RN_WINO_F43 ? 0 : scheme == RN_WINO_F44 ? 1 : Cin >= 1024 ? 2 : 3;
    if (fmt == 1) {
        const unsigned* av = reinterpret_cast<const unsigned*>(static_cast<const char*>(Vs) + h2_v_data(scheme, T, Cin));
        const unsigned* au = reinterpret_cast<const unsigned*>(static_cast<const char*>(us) + h2_u_data(scheme, Cin, Cout));
        return gemm_split_planes(1, rn_split_scheme_nxi(scheme), tag, Vs, us, M, T, Cin, Cout, av, au, h2_bound_v(scheme), h2_bound_u(scheme), st);
    }
    return rn_launch_gemm_bf3_planes(rn_split_scheme_nxi(scheme), tag, Vs, us, M, T, Cin, Cout, st);
}

// x [B,H,W,Cin] -> y [B,H,W,Cout]; us from rn_launch_wino_pack_bf3; ws >= rn_wino_bf3_workspace_bytes(...) bytes
int rn_launch_conv_wino_bf3(int scheme, const float* x, const void* us, const float* bias, const float* alpha, const float* residual,
                            float* y, float* preact, void* ws, int B, int H, int W, int Cin, int Cout, int pad_lo, int act, hipStream_t st)
{
    return rn_launch_conv_wino_bf3_ex(scheme, x, us, bias, alpha, residual, y, preact, ws, B, H, W, Cin, Cout, pad_lo, act, nullptr, nullptr, st);
}

// amax_x: see rn_launch_wino_input_bf3_ex.  amax_y (may be null): receives the bit pattern of max|y|, for the next layer's amax_x.
static int conv_wino_bf3_rec(int scheme, const float* x, const void* us, const float* bias, const float* alpha, const float* residual,
                             float* y, float* preact, void* ws, int B, int H, int W, int Cin, int Cout, int pad_lo, int act,
                             const unsigned* amax_x, unsigned* amax_y, hipStream_t st)
{
    const int m = rn_split_scheme_m(sch_of(scheme));
    const int th = (H + m - 1) / m, tw = (W + m - 1) / m;
    const long long T = (long long)B * th * tw;
    const int cmax = Cin > Cout ? Cin : Cout;
    const long long lim = rn_wino43_plane_limit();
    if ((long long)th * tw * cmax * 4 >= lim)
        return rn_set_error(RN_E_UNSUPPORTED, "conv_wino_bf3: one image's transform plane exceeds the 2 GiB buffer window");
    if (T * cmax * 4 >= lim) {                                  // batch chunks
        const int chunk = (int)((lim - 1) / ((long long)th * tw * cmax * 4));
        for (int b0 = 0; b0 < B; b0 += chunk) {
            const int nb = B - b0 < chunk ? B - b0 : chunk;
            const size_t xo = (size_t)b0 * H * W * Cin, yo = (size_t)b0 * H * W * Cout;
            const int rc = conv_wino_bf3_rec(scheme, x + xo, us, bias, alpha, residual ? residual + yo : nullptr, y + yo,
                                             preact ? preact + yo : nullptr, ws, nb, H, W, Cin, Cout, pad_lo, act, amax_x, amax_y, st);
            if (rc != RN_OK) return rc;
        }
        return RN_OK;
    }
    char* Vs = static_cast<char*>(ws);
    float* M = reinterpret_cast<float*>(Vs + rn_wino_bf3_v_bytes(scheme, T, Cin));
    int rc = rn_launch_wino_input_bf3_ex(scheme, x, Vs, B, H, W, Cin, pad_lo, amax_x, st);
    if (rc != RN_OK) return rc;
    rc = rn_launch_wino_gemm_bf3(scheme, Vs, us, M, T, Cin, Cout, st);
    if (rc != RN_OK) return rc;
    return rn_launch_wino_output_amax(sch_of(scheme), M, bias, alpha, residual, y, preact, B, H, W, Cout, act, amax_y, st);
}

int rn_launch_conv_wino_bf3_ex(int scheme, const float* x, const void* us, const float* bias, const float* alpha, const float* residual,
                               float* y, float* preact, void* ws, int B, int H, int W, int Cin, int Cout, int pad_lo, int act,
                               const unsigned* amax_x, unsigned* amax_y, hipStream_t st)
{
    if (!rn_wino_bf3_supported(scheme, Cin, Cout)) return rn_set_error(RN_E_UNSUPPORTED, "conv_wino_bf3: scheme=%d Cin=%d Cout=%d", scheme, Cin, Cout);
    if (amax_y) { const int rc = rn_launch_word(amax_y, nullptr, st); if (rc != RN_OK) return rc; }
    return conv_wino_bf3_rec(scheme, x, us, bias, alpha, residual, y, preact, ws, B, H, W, Cin, Cout, pad_lo, act, amax_x, amax_y, st);
}
