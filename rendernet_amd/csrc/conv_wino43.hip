// Stride-1 2-D convs through Winograd minimal filtering with 4x4 / 6x6 output tiles, in three launches, exact-fp32 MFMA for the
// multiply stage:
//   scheme F43 -- F(4x4,3x3), 6x6 input tiles, 36 products per 16 outputs instead of 144 (the fused F(2x2,3x3) kernel: 64):
//                 the wide res_block_2d / *_skip convs (slim.conv2d [3,3]: tools/layer_util.py:91-105,
//                 RenderNet_Shader.py:71-84, :91-99);
//   scheme F44 -- F(4x4,4x4), 7x7 input tiles, 49 products per 16 outputs instead of 256 (the F(2x2,2x2)x4 kernel: 144):
//                 e_conv5 / e_conv6 (slim.conv2d [4,4], RenderNet_Shader.py:86-88, :101-103);
//   scheme F63 -- F(6x6,3x3), 8x8 input tiles, 64 products per 36 outputs instead of 324 (1.78 per output against F43's 2.25):
//                 the same 3x3 layers where the map is large enough for the 6-pixel tile grid to pay (ops.py picks per shape;
//                 on the 64x64 res2 maps 11x11 tiles of 64 products replace 16x16 tiles of 36: 0.84 of the multiplies and
//                 0.84 of the V / M traffic; fp32 error 2.7e-5 of max|y| against F43's 9e-6 -- wino_mats.h);
// forward and, with the transposed pack (taps flipped, channel roles swapped, pad_lo = R - 2), the input gradient.
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A          matrices: wino_mats.h (generated, scripts/gen_wino_mats.py)
//
//   1. wino_input_kernel     x [B,H,W,Cin]            -> V [nxi][T tiles][Cin]     (B^T d B per tile and channel; HBM bound)
//   2. wino43_gemm_kernel    V, U [nxi][Cin][Cout]    -> M [nxi][T][Cout]          (nxi GEMMs T x Cin x Cout; MFMA bound)
//   3. wino_output_kernel    M, bias, alpha, residual -> y [B,H,W,Cout]            (A^T m A + the conv epilogue; HBM bound)
//
// Unlike the fused F(2x2,3x3) kernel (conv_wino.hip), which transforms its patch again for every 32 output channels and
// holds all xi of a tile in one wave, the 36 / 49 xi planes do not fit the accumulator file next to a useful channel block,
// so the transforms run once, in their own launches, and the multiply stage is a plain batched GEMM with 256 x 256 blocks.
// V and M live in a caller-provided workspace (nxi*T*(Cin+Cout) floats: 1.8 GB on the headline res2 shape, against 288 GB).
//
// GEMM: 512 threads = 8 waves (4 along tiles x 2 along channels), block 256 tiles x 256 channels, K step 32; both operands
// go global -> LDS by DMA (raw_ptr_buffer_load_lds, 1 KiB per wave instruction), two stages of 64 KiB; v_mfma_f32_32x32x2_f32
// (a wave = 2 x 4 tiles of 32 x 32, 128 accumulators; the 16x16x4 form of the same kernel measured 1-3 % slower and needs 256
// VGPRs against 209).  U is the MFMA A operand so that accumulator registers come in groups of four consecutive output
// channels (16-byte stores).  Persistent grid (one workgroup per CU), items enumerated so that the 32
// workgroups of an XCD share one xi and neighbouring blocks (its L2 then holds their U panel and V panels once).
#include "rn_common.h"
#include "wino_mats.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int GBM = 256, GBN = 256, GBK = 32;
constexpr int G_UB = GBK * GBN * 4;                                                  // U part of a stage: 32 KiB

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// XCD k (= workgroup id % 8) gets a contiguous range of the logical ids: neighbouring tiles share patch pixels / lines
__device__ __forceinline__ unsigned xcd_contiguous(unsigned blk, unsigned nblk8) { return (blk & 7u) * (nblk8 >> 3) + (blk >> 3); }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// 1. input transform V = B^T d B.  thread = (tile, VW channels); S = WinoF43 | WinoF44 | WinoF63 (wino_mats.h).  The matrix entries
// are compile-time constants of fully unrolled loops: zero entries cost nothing, the rest become FMAs.
template <class S, int VW>
__global__ __launch_bounds__(256)
void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int H, int W, int C, int th, int tw,
                       long long T, unsigned nblk8, int pad_lo)
{
    typedef float vec __attribute__((ext_vector_type(VW)));
    constexpr int A = S::TA;
    const unsigned blk = xcd_contiguous(blockIdx.x, nblk8);
    const long long idx = (long long)blk * 256 + threadIdx.x;
    const int CV = C / VW;
    const int cv = (int)(idx % CV);
    const long long t = idx / CV;
    if (t >= T) return;
    const int tx = (int)(t % tw), ty = (int)((t / tw) % th);
    const long long b = t / ((long long)tw * th);
    const int y0 = S::M * ty - pad_lo, x0 = S::M * tx - pad_lo;
    const float* xb = x + ((size_t)b * H * W) * C + cv * VW;
    vec tt[A][A];                                              // (B^T d)[i][col]
#pragma unroll
    for (int col = 0; col < A; ++col) {
        vec d[A];
        const int ix = x0 + col;
#pragma unroll
        for (int r = 0; r < A; ++r) {
            const int iy = y0 + r;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            d[r] = ok ? *reinterpret_cast<const vec*>(xb + ((size_t)iy * W + ix) * C) : vec(0.f);
        }
#pragma unroll
        for (int i = 0; i < A; ++i) {
            vec acc = vec(0.f);
#pragma unroll
            for (int k = 0; k < A; ++k) {
                const float c = S::BT(i, k);
                if (c != 0.f) acc += c * d[k];
            }
            tt[i][col] = acc;
        }
    }
    float* vb = V + (size_t)t * C + cv * VW;
    const size_t plane = (size_t)T * C;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j < A; ++j) {
            vec acc = vec(0.f);
#pragma unroll
            for (int k = 0; k < A; ++k) {
                const float c = S::BT(j, k);
                if (c != 0.f) acc += c * tt[i][k];
            }
            *reinterpret_cast<vec*>(vb + (size_t)(i * A + j) * plane) = acc;       // (non-temporal stores measured 2-4 % slower here)
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3. output transform Y = A^T m A + the conv epilogue.  thread = (tile, VW channels).
// RES (compile time): the launch has a residual -- its values are fetched a whole output row ahead (12 more registers, which the
// residual-free instance must not pay for: 84 registers = six waves per SIMD).
template <class S, int VW, bool RES = false>
__global__ __launch_bounds__(256)
void wino_output_kernel(const float* __restrict__ M, const float* __restrict__ bias, const float* __restrict__ alpha,
                        const float* __restrict__ res, float* __restrict__ y, float* __restrict__ z,
                        int H, int W, int C, int th, int tw, long long T, int act, unsigned nblk8, unsigned* __restrict__ amax)
{
    typedef float vec __attribute__((ext_vector_type(VW)));
    constexpr int A = S::TA;
    const unsigned blk = xcd_contiguous(blockIdx.x, nblk8);
    const long long idx = (long long)blk * 256 + threadIdx.x;
    const int CV = C / VW;
    const int cv = (int)(idx % CV);
    const long long t_raw = idx / CV;
    // amax != nullptr (the consumer is a split-format H2 layer): max |y| of the tensor is gathered on the way -- a wave reduction at
    // the end, so the threads behind the last tile stay (they redo tile T - 1 and store nothing); otherwise they leave here
    if (t_raw >= T && !amax) return;
    const bool live = t_raw < T;
    const long long t = live ? t_raw : T - 1;
    float ymax = 0.f;
    const int tx = (int)(t % tw), ty = (int)((t / tw) % th);
    const long long b = t / ((long long)tw * th);
    const float* mb = M + (size_t)t * C + cv * VW;
    const size_t plane = (size_t)T * C;
    constexpr int MO = S::M;                                   // output pixels per tile side
    vec s[MO][A];                                              // (A^T m)[p][j]
#pragma unroll
    for (int j = 0; j < A; ++j) {
        vec m[A];
#pragma unroll
        for (int i = 0; i < A; ++i) m[i] = __builtin_nontemporal_load(reinterpret_cast<const vec*>(mb + (size_t)(i * A + j) * plane));   // read once: non-temporal (output transform 0.327 -> 0.317 ms on the res2 shape)
#pragma unroll
        for (int p_ = 0; p_ < MO; ++p_) {
            vec acc = vec(0.f);
#pragma unroll
            for (int i = 0; i < A; ++i) {
                const float c = S::AT(p_, i);
                if (c != 0.f) acc += c * m[i];
            }
            s[p_][j] = acc;
        }
    }
    const vec bv = bias ? *reinterpret_cast<const vec*>(bias + cv * VW) : vec(0.f);
    const vec av = (act & RN_ACT_PRELU) ? *reinterpret_cast<const vec*>(alpha + cv * VW) : vec(0.f);
#pragma unroll
    for (int p_ = 0; p_ < MO; ++p_) {
        const int oy = MO * ty + p_;
        if (oy >= H) continue;
        // RES: the residual values of the whole output row are fetched BEFORE the row's arithmetic, as one batch of independent loads --
        // issued one by one at their use they sat behind the bounds branch of every pixel (res2 layer, B = 24: 0.326 -> 0.294 ms)
        vec rv[RES ? MO : 1];
        if constexpr (RES) {
#pragma unroll
            for (int q = 0; q < MO; ++q) {
                const int ox = MO * tx + q;
                rv[q] = ox < W ? __builtin_nontemporal_load(reinterpret_cast<const vec*>(res + (((size_t)b * H + oy) * W + ox) * C + cv * VW)) : vec(0.f);
            }
        }
#pragma unroll
        for (int q = 0; q < MO; ++q) {
            const int ox = MO * tx + q;
            if (ox >= W) continue;
            vec v = bv;
#pragma unroll
            for (int j = 0; j < A; ++j) {
                const float c = S::AT(q, j);
                if (c != 0.f) v += c * s[p_][j];
            }
            const size_t off = (((size_t)b * H + oy) * W + ox) * C + cv * VW;
            if (z && live) *reinterpret_cast<vec*>(z + off) = v;
            if (act & RN_ACT_PRELU) {
#pragma unroll
                for (int e = 0; e < VW; ++e) v[e] = fmaxf(v[e], 0.f) + av[e] * fminf(v[e], 0.f);
            }
            if (act & RN_ACT_ELU) {
#pragma unroll
                for (int e = 0; e < VW; ++e) v[e] = v[e] > 0.f ? v[e] : expf(v[e]) - 1.f;
            }
            if constexpr (RES) v += rv[q];
            if (act & RN_ACT_SIGMOID) {
#pragma unroll
                for (int e = 0; e < VW; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
            }
            if (live) *reinterpret_cast<vec*>(y + off) = v;
            if (amax) {
#pragma unroll
                for (int e = 0; e < VW; ++e) ymax = fmaxf(ymax, fabsf(v[e]));
            }
        }
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o));
        // (a plain read first: once a few waves have reported, almost none exceeds the running maximum -- no atomic storm on one address)
        const unsigned bits = __builtin_bit_cast(unsigned, ymax);
        if ((threadIdx.x & 63) == 0 && bits > *reinterpret_cast<volatile unsigned*>(amax)) atomicMax(amax, bits);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3+1. output transform of one conv FUSED with the input transform of the NEXT conv of a res-block stack (inference):
//     M_a [nxi][T][C]  ->  y = epilogue(A^T m A)  ->  V_b = B^T y B [nxi][T][C]        (res_block_2d, tools/layer_util.py:91-105:
//     x + conv(prelu(conv(x))): conv1 -> conv2 inside a block, conv2 -> conv1 of the next block, ... -> the *_skip conv).
// In the three-launch path the activation y is written by one launch and read straight back (with 1.45x halo re-reads) by
// the next; here it stays in LDS: a workgroup owns (image, CG channels) and walks the tile rows of the image top to bottom;
// per step it output-transforms tile row `it` into a ring of 3*M pixel rows (zero where the map ends: SAME padding), then
// input-transforms tile row `it - 1`, whose (M+2)-row patches are now complete (last row of tile row it-2, tile row it-1,
// first row of tile row it).  thread = (tile of the row, 2 channels), exactly the arithmetic of wino_output_kernel and
// wino_input_kernel per element, so V_b and y are bit-identical to the unfused launches.  y goes to HBM only when the
// caller needs it (the block output: the next block's residual); conv1 -> conv2 writes nothing but V_b.
// Traffic per res-block (two convs, F63 on the 64x64x1024 maps at B=24): 3.85 GB instead of 5.42 GB.
template <class S, int CG>
__global__ __launch_bounds__(256)
void wino_outin_kernel(const float* __restrict__ M, const float* __restrict__ bias, const float* __restrict__ alpha,
                       const float* __restrict__ res, float* __restrict__ y, float* __restrict__ V,
                       int H, int W, int C, int th, int tw, long long T, int act, unsigned nwg, unsigned nblk8)
{
    typedef float vec __attribute__((ext_vector_type(2)));
    constexpr int A = S::TA, MO = S::M, VW = 2, TPT = CG / VW, NR = 3 * MO;
    extern __shared__ __attribute__((aligned(16))) float ysh[];           // [NR rows][cols][CG]
    const int cols = tw * MO + 2;                                          // pixel columns -1 .. tw*MO
    const unsigned L = xcd_contiguous(blockIdx.x, nblk8);                  // consecutive channel groups of an image share M lines: same XCD
    const int ng = C / CG;
    if (L >= nwg) return;
    const int g = (int)(L % (unsigned)ng);
    const long long b = L / (unsigned)ng;
    const int tx = threadIdx.x / TPT, cv = threadIdx.x % TPT;
    const bool active = tx < tw;
    const int c0 = g * CG + cv * VW;
    for (int i = threadIdx.x; i < NR * cols * CG / 4; i += blockDim.x) reinterpret_cast<f32x4*>(ysh)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const size_t plane = (size_t)T * C;
    const vec bv = bias ? *reinterpret_cast<const vec*>(bias + c0) : vec(0.f);
    const vec av = (act & RN_ACT_PRELU) ? *reinterpret_cast<const vec*>(alpha + c0) : vec(0.f);
    __syncthreads();
    for (int it = 0; it <= th; ++it) {
        if (it < th && active) {
            // ---- output transform of tile (b, it, tx) -> ring rows (MO*it .. MO*it+MO-1) % NR
            const long long t = (b * th + it) * tw + tx;
            const float* mb = M + (size_t)t * C + c0;
            vec s[MO][A];                                                   // (A^T m)[p][j]
#pragma unroll
            for (int j = 0; j < A; ++j) {
                vec m[A];
#pragma unroll
                for (int i = 0; i < A; ++i) m[i] = *reinterpret_cast<const vec*>(mb + (size_t)(i * A + j) * plane);
#pragma unroll
                for (int p_ = 0; p_ < MO; ++p_) {
                    vec acc = vec(0.f);
#pragma unroll
                    for (int i = 0; i < A; ++i) {
                        const float c = S::AT(p_, i);
                        if (c != 0.f) acc += c * m[i];
                    }
                    s[p_][j] = acc;
                }
            }
#pragma unroll
            for (int p_ = 0; p_ < MO; ++p_) {
                const int oy = MO * it + p_;
                float* yrow = ysh + ((size_t)(oy % NR) * cols + MO * tx + 1) * CG + cv * VW;
#pragma unroll
                for (int q = 0; q < MO; ++q) {
                    const int ox = MO * tx + q;
                    vec v = bv;
#pragma unroll
                    for (int j = 0; j < A; ++j) {
                        const float c = S::AT(q, j);
                        if (c != 0.f) v += c * s[p_][j];
                    }
                    const bool inimg = oy < H && ox < W;
                    if (act & RN_ACT_PRELU) {
#pragma unroll
                        for (int e = 0; e < VW; ++e) v[e] = fmaxf(v[e], 0.f) + av[e] * fminf(v[e], 0.f);
                    }
                    if (inimg) {
                        const size_t off = (((size_t)b * H + oy) * W + ox) * C + c0;
                        if (res) v += *reinterpret_cast<const vec*>(res + off);
                        if (y) *reinterpret_cast<vec*>(y + off) = v;
                    } else {
                        v = vec(0.f);                                       // beyond the map: the next conv's SAME padding
                    }
                    *reinterpret_cast<vec*>(yrow + q * CG) = v;
                }
            }
        }
        __syncthreads();
        if (it >= 1 && active) {
            // ---- input transform of tile (b, it-1, tx): pixel rows MO*(it-1)-1 .. +A-1, columns MO*tx-1 .. (ring column MO*tx ..)
            const int r = it - 1;
            const long long t = (b * th + r) * tw + tx;
            vec tt[A][A];                                                   // (B^T d)[i][col]
#pragma unroll
            for (int col = 0; col < A; ++col) {
                vec d[A];
#pragma unroll
                for (int k = 0; k < A; ++k) {
                    const int oy = MO * r - 1 + k;
                    d[k] = (oy >= 0 && oy < MO * th)
                         ? *reinterpret_cast<const vec*>(ysh + ((size_t)(oy % NR) * cols + MO * tx + col) * CG + cv * VW) : vec(0.f);
                }
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    vec acc = vec(0.f);
#pragma unroll
                    for (int k = 0; k < A; ++k) {
                        const float c = S::BT(i, k);
                        if (c != 0.f) acc += c * d[k];
                    }
                    tt[i][col] = acc;
                }
            }
            float* vb = V + (size_t)t * C + c0;
#pragma unroll
            for (int i = 0; i < A; ++i)
#pragma unroll
                for (int j = 0; j < A; ++j) {
                    vec acc = vec(0.f);
#pragma unroll
                    for (int k = 0; k < A; ++k) {
                        const float c = S::BT(j, k);
                        if (c != 0.f) acc += c * tt[i][k];
                    }
                    *reinterpret_cast<vec*>(vb + (size_t)(i * A + j) * plane) = acc;
                }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2. the GEMMs  M[xi] = V[xi] (T x Cin) . U[xi] (Cin x Cout), xi = 0 .. nxi-1
struct W43GemmArgs {
    const float* V; const float* U; float* M;
    long long T;
    int Cin, Cout;
    int mblocks, nblocks, ksteps;   // 256-row blocks of T this launch walks (from mb_begin), 256-channel blocks, K steps of 32
    int mb_begin, parts;            // parts: BM-row parts of a block that are enumerated (4 / WM; fewer for the ragged last block)
    int mrows;                      // rows between consecutive m-blocks: GBM, or BM itself when the whole of T is walked in BM-row items
    int item_begin, item_end;       // this launch's range of the items L = (xi*mblocks + mb - mb_begin)*nblocks + nb
    unsigned v_bytes, u_bytes, m_bytes;  // one xi plane of V; one (xi, n-block) panel of U; one xi plane of M
    int probe;                      // RN_WINO43_PROBE (timing experiments; results are wrong when set): 1 no DMA in the loop, 2 no stores, 4 no barrier
};

template <int VP> struct W43Item { const float* vplane; const float* upanel; float* mplane; long long m0; int nb; unsigned voff[VP]; };

// WM = waves along the tile rows: 4 -> block 256 rows x 256 channels (waves 4 x 2, wave tile 64 x 128 = 2 x 4 MFMA tiles,
// 128 accumulators); 2 -> block 128 x 256 (waves 2 x 4, wave tile 64 x 64); 1 -> block 64 x 256 (waves 1 x 8, 64 x 32): the launcher
// runs the last, partial round of a launch as half or quarter items so that it costs half / three quarters of a round.
// TAG only names the kernel per layer class in profiler tables (0: F43, Cin >= 1024 -- the res2 trunk; 1: narrower F43 layers;
// 2: F(4x4,4x4); 3: F63, Cin >= 1024; 4: narrower F63 layers)
template <int WM, int TAG>
__global__ __launch_bounds__(512, 1)
void wino43_gemm_kernel(const W43GemmArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)     // the host pass only needs the stub: the amdgcn builtins below do not instantiate there
    constexpr int WN = 8 / WM, NT = 16 / WN;                          // 16-channel groups per wave along channels (8 | 4 | 2)
    constexpr int BM = WM * 64, VB = BM * GBK * 4, STAGE = VB + G_UB; // V 32 | 16 KiB + U 32 KiB per stage
    constexpr int VP = BM / 8 / 8;                                    // V DMA pieces per wave and stage (4 | 2 | 1)
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [stage][V BM x 32 | U 8 x 256 x 4]
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hb = lane >> 5;                       // 32x32x2 MFMA: lane = (row / column, k)
    const int wm = wave / WN, wn = wave % WN;

    // fragment-read offsets inside a stage.  A K step of 32 is consumed as four sub-groups q of 8 k: lane (l32, hb) supplies
    // k = 8q + 4hb + s to MFMA s = 0..3 of the sub-group -- one 16-byte LDS read per operand tile.
    const unsigned vrow = (unsigned)((wm * 64 + l32) * 128);
    const unsigned vsw = (unsigned)((l32 >> 1) & 7);
    unsigned vaddr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) vaddr[q] = vrow + (((unsigned)(2 * q + hb) ^ vsw) << 4);
    const unsigned uaddr = (unsigned)(VB + ((hb * 256) + wn * (NT * 16) + l32) * 16);

    // one item: decoded into scalars + the per-lane V offsets of this wave's DMA pieces (piece p = wave + 8i covers rows
    // 8p .. 8p+7 (lane >> 3); chunk slot lane & 7 holds chunk slot ^ swizzle(row))
    typedef W43Item<VP> Item;
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
        it.vplane = a.V + (size_t)xi * a.T * a.Cin;
        it.upanel = a.U + ((size_t)xi * a.nblocks + nb) * ((size_t)a.Cin * GBN);
        it.mplane = a.M + (size_t)xi * a.T * a.Cout;
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const int row = (wave + 8 * i) * 8 + (lane >> 3);
            const unsigned chunk = (unsigned)(lane & 7) ^ (unsigned)((row >> 1) & 7);
            it.voff[i] = (unsigned)(((it.m0 + row) * a.Cin) * 4) + chunk * 16u;     // rows >= T fall outside v_bytes: zeros
        }
        return true;
    };
    auto issue = [&](const Item& it, int s, int stage) {
        const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(it.vplane), 0, a.v_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(it.upanel), 0, a.u_bytes, 0x00020000);
        char* sb = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < VP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (lds_void*)(sb + (wave + 8 * i) * 1024), 16, it.voff[i], s * (GBK * 4), 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lds_void*)(sb + VB + (wave + 8 * i) * 1024), 16,
                                                     (unsigned)((wave + 8 * i) * 1024 + lane * 16), s * G_UB, 0, 0);
    };

    constexpr int NT2 = NT / 2;                                       // 32-channel MFMA tiles per wave along channels (4 | 2 | 1)
    f32x16 acc[2][NT2];
    auto load_frags = [&](const char* sb, int q, f32x4 (&v)[2], f32x4 (&u)[NT2]) {
        const char* vp = sb + vaddr[q];
        const char* up = sb + uaddr + q * (2 * 256 * 16);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) v[mt] = *reinterpret_cast<const f32x4*>(vp + mt * (32 * 128));
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) u[nt] = *reinterpret_cast<const f32x4*>(up + nt * (32 * 16));
    };
    auto mfmas = [&](const f32x4 (&v)[2], const f32x4 (&u)[NT2]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[nt][j], v[mt][j], acc[mt][nt], 0, 0, 0);
    };

    Item cur, nxt;
    if (!decode(0, cur)) return;
    // Software pipeline: the fragments of sub-group q+1 are read from LDS while the 32 (16, 8) MFMAs of sub-group q run, across
    // K steps and across items -- the one barrier of a step sits before its last sub-group, where the next stage has landed
    // (DMAs issued at the head of the step; in an item's last step they fetch the NEXT item's first stage) and every wave
    // has finished reading the current one.  The epilogue's stores drain under the next item's first step.
    f32x4 v0[2], u0[NT2], v1[2], u1[NT2];
    issue(cur, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_frags(smem, 0, v0, u0);
    int stage = 0;
    for (int r = 0;; ++r) {
        const bool have_next = decode(r + 1, nxt);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        for (int s = 0; s < a.ksteps; ++s) {
            const char* sb = smem + stage * STAGE;
            const char* sn = smem + (stage ^ 1) * STAGE;
            const bool last = s + 1 == a.ksteps;
            if (!(a.probe & 1)) {
                if (!last) issue(cur, s + 1, stage ^ 1);
                else if (have_next) issue(nxt, 0, stage ^ 1);
            }
            load_frags(sb, 1, v1, u1);
            mfmas(v0, u0);
            load_frags(sb, 2, v0, u0);
            mfmas(v1, u1);
            load_frags(sb, 3, v1, u1);
            mfmas(v0, u0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(a.probe & 4)) __syncthreads();
            if (!last || have_next) load_frags(sn, 0, v0, u0);
            mfmas(v1, u1);
            stage ^= 1;
        }
        // D (32 x 32) = U-tile (rows: channels) x V-tile (cols: tile rows): register r of lane (l32, hb) is channel
        // (r & 3) + 8*(r >> 2) + 4*hb of the tile, tile row l32 -> four 16-byte stores per MFMA tile
        if (!(a.probe & 2)) {
            const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(cur.mplane, 0, a.m_bytes, 0x00020000);
            const unsigned mo = (unsigned)(((cur.m0 + wm * 64 + l32) * a.Cout + cur.nb * GBN + wn * (NT * 16) + hb * 4) * 4);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 o = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), mrsrc,
                                                               mo + (unsigned)(mt * 32 * a.Cout * 4) + nt * 128 + g * 32, 0, 0);
                    }
        }
        if (!have_next) break;
        cur = nxt;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// filter transform U = G g G^T, packed [nxi][Cout/256][Cin/4][256 channels][4 k]: the panel of one (xi, 256-channel block)
// is contiguous, K step s of the GEMM is its s-th 32 KiB.  transposed = 0 reads a conv filter w_tf[R,R,Cin,Cout];
// transposed = 1 a conv_transpose filter w_tf[R,R,Cout,Cin] with the taps flipped (a stride-1 transposed conv = the input
// gradient of the conv with that filter).  Off the hot path (once per weight update): the R*R-term sums run in double.
template <class S>
__global__ __launch_bounds__(256)
void wino_pack_kernel(const float* __restrict__ w_tf, float* __restrict__ u, int Cin, int Cout, int transposed)
{
    // thread = (4 input channels kg, output channel co): the R*R taps of its four filters are read once (coalesced along
    // co for a conv filter; along c for a transposed one), every xi plane gets one 16-byte store (contiguous along co)
    constexpr int A = S::TA, R = S::R;
    const int nkg = Cin / 4, nblocks = Cout / 256;
    const size_t total = (size_t)nkg * Cout;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int slot = (int)(idx & 255);
        const size_t rest = idx >> 8;
        const int kg = (int)(rest % nkg), nb = (int)(rest / nkg);
        const int co = nb * 256 + slot;
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
        float* ub = u + (((size_t)nb * nkg + kg) * 256 + slot) * 4;
        const size_t plane = (size_t)nblocks * nkg * 1024;      // floats per xi
#pragma unroll
        for (int i = 0; i < A; ++i)
#pragma unroll
            for (int j = 0; j < A; ++j) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < R; ++q) acc = __builtin_fma(gg[i][q][r], S::G(j, q), acc);
                    o[r] = (float)acc;
                }
                st4(ub + (size_t)(i * A + j) * plane, o);
            }
    }
}

int rn_wino_scheme_nxi(int scheme) { return scheme == RN_WINO_F43 ? WinoF43::NXI : scheme == RN_WINO_F44 ? WinoF44::NXI : scheme == RN_WINO_F63 ? WinoF63::NXI : 0; }
int rn_wino_scheme_r(int scheme) { return scheme == RN_WINO_F43 ? WinoF43::R : scheme == RN_WINO_F44 ? WinoF44::R : scheme == RN_WINO_F63 ? WinoF63::R : 0; }
int rn_wino_scheme_m(int scheme) { return scheme == RN_WINO_F43 ? WinoF43::M : scheme == RN_WINO_F44 ? WinoF44::M : scheme == RN_WINO_F63 ? WinoF63::M : 0; }

bool rn_wino43_supported(int scheme, int Cin, int Cout)
{
    static const bool off = getenv("RN_NO_WINOGRAD43") != nullptr || getenv("RN_NO_WINOGRAD") != nullptr;
    static const bool off44 = getenv("RN_NO_WINOGRAD44") != nullptr;
    static const bool off63 = getenv("RN_NO_WINOGRAD63") != nullptr;
    if (off || (scheme == RN_WINO_F44 && off44) || (scheme == RN_WINO_F63 && off63) || rn_wino_scheme_nxi(scheme) == 0) return false;
    return Cin >= 32 && Cin % 32 == 0 && Cout >= 256 && Cout % 256 == 0;
}

int rn_launch_wino_pack(int scheme, const float* w_tf, float* u, int Cin, int Cout, int transposed, hipStream_t st)
{
    const size_t tot = (size_t)(Cin / 4) * Cout;
    const unsigned nbw = (unsigned)((tot + 255) / 256 > 65536 ? 65536 : (tot + 255) / 256);
    if (scheme == RN_WINO_F43) hipLaunchKernelGGL(wino_pack_kernel<WinoF43>, dim3(nbw), dim3(256), 0, st, w_tf, u, Cin, Cout, transposed);
    else if (scheme == RN_WINO_F44) hipLaunchKernelGGL(wino_pack_kernel<WinoF44>, dim3(nbw), dim3(256), 0, st, w_tf, u, Cin, Cout, transposed);
    else if (scheme == RN_WINO_F63) hipLaunchKernelGGL(wino_pack_kernel<WinoF63>, dim3(nbw), dim3(256), 0, st, w_tf, u, Cin, Cout, transposed);
    else return rn_set_error(RN_E_INVALID, "wino_pack: unknown scheme %d", scheme);
    return rn_check_launch("wino_pack");
}

// the three stages on their own (the C ABI exposes them: a caller can keep V / M, or time the stages separately)
int rn_launch_wino_input(int scheme, const float* x, float* V, int B, int H, int W, int C, int pad_lo, hipStream_t st)
{
    const int m = rn_wino_scheme_m(scheme);
    if (m == 0) return rn_set_error(RN_E_INVALID, "wino_input: unknown scheme %d", scheme);
    const int th = (H + m - 1) / m, tw = (W + m - 1) / m;
    const long long T = (long long)B * th * tw;
    const int vw = scheme == RN_WINO_F43 ? 4 : 2;                       // channels per thread: 36 x 4 | 49 x 2 | 64 x 2 registers of patch
    const unsigned long long n = ((unsigned long long)T * (C / vw) + 255) / 256;
    const unsigned nblk8 = (unsigned)((n + 7) / 8 * 8);
    if (scheme == RN_WINO_F43)
        hipLaunchKernelGGL((wino_input_kernel<WinoF43, 4>), dim3(nblk8), dim3(256), 0, st, x, V, H, W, C, th, tw, T, nblk8, pad_lo);
    else if (scheme == RN_WINO_F44)
        hipLaunchKernelGGL((wino_input_kernel<WinoF44, 2>), dim3(nblk8), dim3(256), 0, st, x, V, H, W, C, th, tw, T, nblk8, pad_lo);
    else
        hipLaunchKernelGGL((wino_input_kernel<WinoF63, 2>), dim3(nblk8), dim3(256), 0, st, x, V, H, W, C, th, tw, T, nblk8, pad_lo);
    return rn_check_launch("wino_input");
}

// items [begin, end) of the block range the args name, `parts` BM-row parts of each (0: all 4 / WM of them)
template <int WM, int TAG>
static int wino43_gemm_launch_t(W43GemmArgs a, int begin, int end, int parts, hipStream_t st)
{
    a.item_begin = begin; a.item_end = end; a.parts = parts > 0 ? parts : 4 / WM;
    const size_t lds = (size_t)2 * (WM * 64 * GBK * 4 + G_UB);
    auto kern = wino43_gemm_kernel<WM, TAG>;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc_ != RN_OK) return rc_; }
    const int n = (end - begin) * a.parts;
    hipLaunchKernelGGL(kern, dim3(n < 256 ? (unsigned)((n + 7) / 8 * 8) : 256u), dim3(512), lds, st, a);
    return rn_check_launch("wino43_gemm");
}

template <int WM>
static int wino43_gemm_launch_w(int tag, const W43GemmArgs& a, int begin, int end, int parts, hipStream_t st)
{
    switch (tag) {
    case 0: return wino43_gemm_launch_t<WM, 0>(a, begin, end, parts, st);
    case 1: return wino43_gemm_launch_t<WM, 1>(a, begin, end, parts, st);
    case 2: return wino43_gemm_launch_t<WM, 2>(a, begin, end, parts, st);
    case 3: return wino43_gemm_launch_t<WM, 3>(a, begin, end, parts, st);
    default: return wino43_gemm_launch_t<WM, 4>(a, begin, end, parts, st);
    }
}

static int wino43_gemm_launch(int wm, int tag, const W43GemmArgs& a, int begin, int end, int parts, hipStream_t st)
{
    return wm == 4 ? wino43_gemm_launch_w<4>(tag, a, begin, end, parts, st)
         : wm == 2 ? wino43_gemm_launch_w<2>(tag, a, begin, end, parts, st) : wino43_gemm_launch_w<1>(tag, a, begin, end, parts, st);
}

int rn_launch_wino_gemm(int scheme, const float* V, const float* u, float* M, long long T, int Cin, int Cout, hipStream_t st)
{
    if (!rn_wino43_supported(scheme, Cin, Cout)) return rn_set_error(RN_E_UNSUPPORTED, "wino_gemm: scheme=%d Cin=%d Cout=%d", scheme, Cin, Cout);
    if (T < 1 || (T + GBM) * (Cin > Cout ? Cin : Cout) * 4 >= 0xffffff00LL || T * (Cin > Cout ? Cin : Cout) * 4 >= 0x7fffff00LL)
        return rn_set_error(RN_E_UNSUPPORTED, "wino_gemm: a transform plane must stay below the 2 GiB buffer window");
    W43GemmArgs a;
    a.V = V; a.U = u; a.M = M; a.T = T; a.Cin = Cin; a.Cout = Cout;
    a.nblocks = Cout / GBN; a.ksteps = Cin / GBK;
    const int nxi = rn_wino_scheme_nxi(scheme);
    const int full = (int)(T / GBM), ragged = (int)(T % GBM);          // whole 256-row blocks; rows of the last, partial one
    a.v_bytes = (unsigned)(T * Cin * 4); a.u_bytes = (unsigned)((size_t)Cin * GBN * 4); a.m_bytes = (unsigned)(T * Cout * 4);
    { static const int probe = getenv("RN_WINO43_PROBE") ? atoi(getenv("RN_WINO43_PROBE")) : 0; a.probe = probe; }
    // one workgroup per CU takes items id, id + 256, ...: the items of a last, partial round run as half items (128 rows:
    // rounds of half the length) or quarter items (64 rows) when that fills the machine better -- rem = 128: 256 half items =
    // half a round; rem = 176: 704 quarter items = 3 quarter rounds instead of a whole one
    static const bool notail = getenv("RN_WINO43_NOTAIL") != nullptr;
    static const bool nouniform = getenv("RN_WINO43_NOUNIFORM") != nullptr;
    const int tag = scheme == RN_WINO_F44 ? 2 : scheme == RN_WINO_F63 ? (Cin >= 1024 ? 3 : 4) : Cin >= 1024 ? 0 : 1;
    a.mrows = GBM;
    // Few tiles (one GPU's share of a strongly scaled batch: 3 frames = 363 tiles of F63): whole blocks + a ragged tail are two
    // launches of one item per CU each, all pipeline fill and drain.  When T is below two blocks, walk ALL of T in 128- or 64-row
    // items instead -- one launch, several items per CU back to back (an item's last step prefetches the next item's first) --
    // whenever that costs no more quarter rounds than the split plan.
    static const int umax = getenv("RN_WINO43_UNIFORM_MAXFULL") ? atoi(getenv("RN_WINO43_UNIFORM_MAXFULL")) : 1;      // measurement
    if (!notail && !nouniform && full <= umax) {
        auto tail_cost = [](int rem) {                      // quarter rounds of a last, partial round of `rem` whole items
            if (rem == 0) return 0;
            const int half = (2 * rem + 255) / 256 * 2, quarter = (4 * rem + 255) / 256;
            int c = 4;
            if (half < c) c = half;
            if (quarter < c) c = quarter;
            return c;
        };
        const int nfull = nxi * full * a.nblocks;
        int cost_split = nfull / 256 * 4 + tail_cost(nfull % 256);
        if (ragged > 0) {
            const int nit = nxi * a.nblocks;
            int best = (nit + 255) / 256 * 4;
            for (int wm = 2; wm >= 1; --wm) {
                const int parts = (ragged + wm * 64 - 1) / (wm * 64), cost = (nit * parts + 255) / 256 * wm;
                if (cost < best) best = cost;
            }
            cost_split += best;
        }
        int uwm = 0, ucost = cost_split + 1;
        for (int wm = 2; wm >= 1; --wm) {
            const long long items = (long long)nxi * a.nblocks * ((T + wm * 64 - 1) / (wm * 64));
            const int cost = (int)((items + 255) / 256) * wm;
            if (cost < ucost) { ucost = cost; uwm = wm; }
        }
        if (uwm != 0 && ucost <= cost_split) {
            a.mb_begin = 0; a.mrows = uwm * 64; a.mblocks = (int)((T + a.mrows - 1) / a.mrows);
            return wino43_gemm_launch(uwm, tag, a, 0, nxi * a.mblocks * a.nblocks, 1, st);
        }
    }
    if (full > 0) {
        a.mb_begin = 0; a.mblocks = full;
        const int nitems = nxi * full * a.nblocks;
        const int rem = nitems % 256;
        int tail_wm = 4;
        if (!notail && rem != 0) {
            const int half = (2 * rem + 255) / 256 * 2, quarter = (4 * rem + 255) / 256;       // cost in quarter rounds (whole: 4)
            if (half < 4) tail_wm = 2;
            if (quarter < (tail_wm == 2 ? half : 4)) tail_wm = 1;
        }
        const int tail = tail_wm == 4 ? 0 : rem;
        if (nitems - tail > 0) {
            const int rc = wino43_gemm_launch(4, tag, a, 0, nitems - tail, 0, st);
            if (rc != RN_OK) return rc;
        }
        if (tail > 0) {
            const int rc = wino43_gemm_launch(tail_wm, tag, a, nitems - tail, nitems, 0, st);
            if (rc != RN_OK) return rc;
        }
    }
    if (ragged > 0) {
        // the last block's rows (T = 2904 tiles of the F63 res2 maps: 88) as the cheapest of whole / half / quarter items that
        // cover them: 256 (xi, n-block) pairs x one 128-row part = half a round instead of a whole one
        a.mb_begin = full; a.mblocks = 1;
        const int nitems = nxi * a.nblocks;
        int best_wm = 4, best_parts = 1, best_cost = (nitems + 255) / 256 * 4;
        for (int wm = 2; wm >= 1 && !notail; --wm) {
            const int parts = (ragged + wm * 64 - 1) / (wm * 64);
            const int cost = (nitems * parts + 255) / 256 * wm;
            if (cost < best_cost) { best_wm = wm; best_parts = parts; best_cost = cost; }
        }
        return wino43_gemm_launch(best_wm, tag, a, 0, nitems, best_parts, st);
    }
    return RN_OK;
}

int rn_launch_wino_output(int scheme, const float* M, const float* bias, const float* alpha, const float* residual, float* y,
                          float* preact, int B, int H, int W, int C, int act, hipStream_t st)
{
    return rn_launch_wino_output_amax(scheme, M, bias, alpha, residual, y, preact, B, H, W, C, act, nullptr, st);
}

// ... with max |y| gathered into *amax (bit pattern of a non-negative float, for a consumer in split format H2): an atomic maximum
// onto what the word holds -- the caller zeroes it (once, however many launches write one tensor)
int rn_launch_wino_output_amax(int scheme, const float* M, const float* bias, const float* alpha, const float* residual, float* y,
                               float* preact, int B, int H, int W, int C, int act, unsigned* amax, hipStream_t st)
{
    const int m = scheme == RN_WINO_F11 ? 1 : rn_wino_scheme_m(scheme);
    if (m == 0) return rn_set_error(RN_E_INVALID, "wino_output: unknown scheme %d", scheme);
    const int th = (H + m - 1) / m, tw = (W + m - 1) / m;
    const long long T = (long long)B * th * tw;
    const int vw = (scheme == RN_WINO_F43 || scheme == RN_WINO_F11) ? 4 : 2;
    const unsigned long long n = ((unsigned long long)T * (C / vw) + 255) / 256;
    const unsigned nblk8 = (unsigned)((n + 7) / 8 * 8);
#define RN_OUT_LAUNCH(S_, VW_)                                                                                                              \
    do {                                                                                                                                    \
        if (residual) hipLaunchKernelGGL((wino_output_kernel<S_, VW_, true>), dim3(nblk8), dim3(256), 0, st, M, bias, alpha, residual, y,  \
                                         preact, H, W, C, th, tw, T, act, nblk8, amax);                                                     \
        else hipLaunchKernelGGL((wino_output_kernel<S_, VW_, false>), dim3(nblk8), dim3(256), 0, st, M, bias, alpha, residual, y, preact,  \
                                H, W, C, th, tw, T, act, nblk8, amax);                                                                      \
    } while (0)
    if (scheme == RN_WINO_F11) RN_OUT_LAUNCH(WinoF11, 4);          // one plane, identity transform: the conv epilogue over M (split path of a 1x1 filter, conv_wino_bf3.hip)
    else if (scheme == RN_WINO_F43) RN_OUT_LAUNCH(WinoF43, 4);
    else if (scheme == RN_WINO_F44) RN_OUT_LAUNCH(WinoF44, 2);
    else RN_OUT_LAUNCH(WinoF63, 2);
#undef RN_OUT_LAUNCH
    return rn_check_launch("wino_output");
}

// The fused transform (wino_outin_kernel).  Returns RN_E_UNSUPPORTED without setting an error when the shape does not fit its
// tiling (the caller then runs the two separate launches): channels not a multiple of 16, more than 32 tiles per row, a ring
// beyond the CU's LDS, an epilogue other than bias / PReLU / residual.
int rn_launch_wino_outin(int scheme, const float* M, const float* bias, const float* alpha, const float* residual, float* y,
                         float* V, int B, int H, int W, int C, int act, hipStream_t st)
{
    static const bool off = getenv("RN_NO_WINO_OUTIN") != nullptr;
    const int m = rn_wino_scheme_m(scheme);
    if (off || m == 0 || scheme == RN_WINO_F44) return RN_E_UNSUPPORTED;
    constexpr int CG = 16;
    const int th = (H + m - 1) / m, tw = (W + m - 1) / m;
    const long long T = (long long)B * th * tw;
    const size_t lds = (size_t)3 * m * (tw * m + 2) * CG * sizeof(float);
    if (C % CG != 0 || tw * (CG / 2) > 256 || lds > (size_t)160 * 1024 || (act & ~RN_ACT_PRELU) != 0) return RN_E_UNSUPPORTED;
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "wino_outin: PReLU needs alpha");
    const unsigned nwg = (unsigned)((long long)B * (C / CG));
    const unsigned nblk8 = (nwg + 7) / 8 * 8;
    const unsigned threads = tw * (CG / 2) <= 128 ? 128u : 256u;
    if (scheme == RN_WINO_F43) {
        auto kern = wino_outin_kernel<WinoF43, CG>;
        { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc_ != RN_OK) return rc_; }
        hipLaunchKernelGGL(kern, dim3(nblk8), dim3(threads), lds, st, M, bias, alpha, residual, y, V, H, W, C, th, tw, T, act, nwg, nblk8);
    } else {
        auto kern = wino_outin_kernel<WinoF63, CG>;
        { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc_ != RN_OK) return rc_; }
        hipLaunchKernelGGL(kern, dim3(nblk8), dim3(threads), lds, st, M, bias, alpha, residual, y, V, H, W, C, th, tw, T, act, nwg, nblk8);
    }
    return rn_check_launch("wino_outin");
}

long long rn_wino43_plane_limit()
{
    const char* e = getenv("RN_WINO43_MAX_PLANE");
    const long long v = e ? atoll(e) : 0;
    return (v > 0 && v < 0x7fffff00LL) ? v : 0x7fffff00LL;
}

size_t rn_wino43_workspace_floats(int scheme, int B, int H, int W, int Cin, int Cout)
{
    const int m = rn_wino_scheme_m(scheme);
    if (m == 0) return 0;
    const size_t T = (size_t)B * ((H + m - 1) / m) * ((W + m - 1) / m);
    return (size_t)rn_wino_scheme_nxi(scheme) * T * ((size_t)Cin + Cout);
}

// x [B,H,W,Cin] -> y [B,H,W,Cout]; u from wino_pack_kernel; ws >= rn_wino43_workspace_floats(...) floats; pad_lo = rows /
// columns of zero padding before the first pixel (SAME conv: (R-1)/2 = 1; stride-1 transposed conv: R-1-1)
int rn_launch_conv_wino43(int scheme, const float* x, const float* u, const float* bias, const float* alpha, const float* residual,
                          float* y, float* preact, float* ws, int B, int H, int W, int Cin, int Cout, int pad_lo, int act, hipStream_t st)
{
    if (!rn_wino43_supported(scheme, Cin, Cout)) return rn_set_error(RN_E_UNSUPPORTED, "conv_wino43: scheme=%d Cin=%d Cout=%d", scheme, Cin, Cout);
    const int m = rn_wino_scheme_m(scheme);
    const int th = (H + m - 1) / m, tw = (W + m - 1) / m;
    const long long T = (long long)B * th * tw;
    const int cmax = Cin > Cout ? Cin : Cout;
    // every xi plane must fit a buffer resource (2 GiB); RN_WINO43_MAX_PLANE lowers the limit so that tests can reach the
    // batch-chunk branch without 150 GB of workspace
    const long long lim = rn_wino43_plane_limit();
    if ((long long)th * tw * cmax * 4 >= lim)
        return rn_set_error(RN_E_UNSUPPORTED, "conv_wino43: one image's transform plane exceeds the 2 GiB buffer window");
    if (T * cmax * 4 >= lim) {                                  // batch chunks
        const int chunk = (int)((lim - 1) / ((long long)th * tw * cmax * 4));      // >= 1: one image fits (checked above)
        for (int b0 = 0; b0 < B; b0 += chunk) {
            const int nb = B - b0 < chunk ? B - b0 : chunk;
            const size_t xo = (size_t)b0 * H * W * Cin, yo = (size_t)b0 * H * W * Cout;
            const int rc = rn_launch_conv_wino43(scheme, x + xo, u, bias, alpha, residual ? residual + yo : nullptr, y + yo,
                                                 preact ? preact + yo : nullptr, ws, nb, H, W, Cin, Cout, pad_lo, act, st);
            if (rc != RN_OK) return rc;
        }
        return RN_OK;
    }
    float* V = ws;
    float* M = ws + (size_t)rn_wino_scheme_nxi(scheme) * T * Cin;
    int rc = rn_launch_wino_input(scheme, x, V, B, H, W, Cin, pad_lo, st);
    if (rc != RN_OK) return rc;
    rc = rn_launch_wino_gemm(scheme, V, u, M, T, Cin, Cout, st);
    if (rc != RN_OK) return rc;
    return rn_launch_wino_output(scheme, M, bias, alpha, residual, y, preact, B, H, W, Cout, act, st);
}
