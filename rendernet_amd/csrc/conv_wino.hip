// Winograd F(2x2,3x3) convolution on the gfx950 matrix cores, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// Serves the stride-1 3x3 SAME 2-D convs of the RenderNet trunk -- res_block_2d / the *_skip convs,
// tools/layer_util.py:101-104, RenderNet_Shader.py:71-84,91-99 (86.9 % of the path's FLOPs) -- and
// their input gradients.  The direct implicit-GEMM kernel (conv_igemm.hip) already runs at 0.90 of
// the fp32 MFMA peak on these layers, so the only lever left is to issue fewer MFMAs: F(2x2,3x3)
// computes a 2x2 output tile from a 4x4 input tile with 16 multiplies per (cin, cout) pair instead
// of 36 (2.25x fewer), still in fp32.
//
//     Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A          per 2x2 output tile and output channel
//
// One fused kernel, no transformed tensors in HBM:
//   * filters are pre-transformed once by rn_pack_weights (RN_PACK_CONV_WINO): U[xi][c][n], 16 "xi"
//     planes, stored [Cout/32][Cin/16][16 xi][4][32][4] so that every (n-block, 16-channel step) is one
//     contiguous 32 KiB piece that goes global -> LDS with buffer_load ... lds;
//   * a workgroup (512 threads = 8 waves, two per SIMD, 128 accumulator registers each) owns a block of
//     16x8 tiles (32x16 outputs) x 32 output channels x all 16 xi.  Its RAW 34x18-pixel input patch goes
//     global -> LDS, 16 channels per stage (SAME padding from the buffer bounds check);
//   * the input transform B^T d B happens at fragment-read time: wave w owns tile row w; a lane reads the
//     16 pixels of its tile (ds_read_b128 = 4 channels each) and 32 vector adds give the 16 xi fragments;
//     no two waves transform the same tile.  fp32 MFMA runs at the fp32 VECTOR rate and the adds measurably
//     take issue time from it, so they are v_pk_add_f32 (one packed add per MFMA pair);
//   * per xi the wave multiplies its 16 tiles x 32 channels (two 16x16 MFMA tiles), so all 16 xi of a
//     (tile, channel) sit in one lane and the output transform A^T M A is a per-lane sum; the epilogue
//     (bias, PReLU, residual, pre-activation) applies to the 2x2 outputs directly.
#include "rn_common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WinoArgs {
    const float* x; const float* u; const float* bias; const float* alpha; const float* res; float* y; float* z;
    unsigned x_bytes, u_bytes;
    int B, H, W, Cin, Cout;
    int bh, bw;                 // 16x8-tile blocks per image along H (16 rows each) and W (32 columns each)
    int mblocks, nblocks;       // B*bh*bw, Cout/32
    int nstep;                  // Cin/16
    int act;
};

namespace {
constexpr int WPW = 34, WPH = 18;         // patch: 32+2 columns, 16+2 rows
constexpr int WNPIX = WPW * WPH;          // 612
constexpr int WRAW_PIECES = 40;           // 1 KiB DMA pieces of 16 pixels x 64 B (612 -> 640 pixel slots: 5 per wave)
constexpr int WRAW_B = WRAW_PIECES * 1024;   // bytes per raw stage (40 960)
constexpr int WU_B = 16 * 4 * 32 * 16;    // bytes per U stage (32 768)
constexpr unsigned WOOB = 0x80000000u;
}

size_t rn_wino_lds_bytes() { return (size_t)2 * WRAW_B + 2 * WU_B; }

__device__ __forceinline__ f32x4 pk_add(f32x4 x, f32x4 y)
{
    f32x4 r;
    asm("v_pk_add_f32 %0, %2, %3\n\tv_pk_add_f32 %1, %4, %5"
        : "=&v"(*reinterpret_cast<double*>(&r)), "=&v"(*(reinterpret_cast<double*>(&r) + 1))
        : "v"(*reinterpret_cast<double*>(&x)), "v"(*reinterpret_cast<double*>(&y)),
          "v"(*(reinterpret_cast<double*>(&x) + 1)), "v"(*(reinterpret_cast<double*>(&y) + 1)));
    return r;
}
__device__ __forceinline__ f32x4 pk_sub(f32x4 x, f32x4 y)
{
    f32x4 r;
    asm("v_pk_add_f32 %0, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %4, %5 neg_lo:[0,1] neg_hi:[0,1]"
        : "=&v"(*reinterpret_cast<double*>(&r)), "=&v"(*(reinterpret_cast<double*>(&r) + 1))
        : "v"(*reinterpret_cast<double*>(&x)), "v"(*reinterpret_cast<double*>(&y)),
          "v"(*(reinterpret_cast<double*>(&x) + 1)), "v"(*(reinterpret_cast<double*>(&y) + 1)));
    return r;
}

// PROBE (measurement switches, RN_WINO_PROBE; 0 = the product kernel): 1 = skip the input transform (wrong results),
// 2 = no DMA inside the loop (wrong results), 4 = input transform with plain v_add/v_sub instead of v_pk_add_f32,
// 8 = all DMAs of a step at its top instead of interleaved with the MFMAs.
template <int PROBE>
__global__ __launch_bounds__(512, 1)
void conv_wino_kernel(const WinoArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [raw 0][raw 1][U 0][U 1]
    typedef __attribute__((address_space(3))) void lds_void;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;

    // block id -> (m-block, n-block).  Enumeration e: groups of 8 m-blocks, n-major inside a group, so that the 32
    // workgroups resident on one XCD (block id % 8; one workgroup per CU) stream 4 filter slabs and 8 patches between
    // them, and the 8 XCDs of a round of 256 read the same 8 patches.
    int mb, nb;
    {
        const int T = a.mblocks * a.nblocks, id = blockIdx.x;
        int e = id;
        if (id < (T & ~255)) { const int s = id >> 3; e = (s >> 5) * 256 + (id & 7) * 32 + (s & 31); }
        const int per = 8 * a.nblocks;
        const int g = e / per, full = a.mblocks >> 3;
        int rem = e - g * per, gs = 8, g0 = g;
        if (g >= full) { rem = e - full * per; gs = a.mblocks - full * 8; g0 = full; }
        nb = rem / gs;
        mb = g0 * 8 + rem % gs;
    }
    const int bx = mb % a.bw, by = (mb / a.bw) % a.bh, b = mb / (a.bw * a.bh);
    const int y0 = by * 16 - 1, x0 = bx * 32 - 1;

    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ursrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u_bytes, 0x00020000);

    // raw-patch DMA: piece p = wave + 8 i (i < 5) holds pixels q = 16 p + lane/4 (q = py*34 + px); the lane
    // fetches LOGICAL chunk (lane%4) ^ swz(px) into physical slot lane%4, swz(px) = (px>>1)&3 (two lanes of a
    // ds_read_b128 group at most share a 16-B slot)
    unsigned roff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int p = wave + 8 * i;
        const int q = p * 16 + (lane >> 2);
        const int py = q / WPW, px = q - py * WPW;
        const int iy = y0 + py, ix = x0 + px;
        const bool ok = q < WNPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const unsigned e = (unsigned)((b * a.H + iy) * a.W + ix) * (unsigned)a.Cin * 4u;
        roff[i] = ok ? e + (unsigned)(((lane & 3) ^ ((px >> 1) & 3)) * 16) : WOOB;
    }
    // filter DMA: the 32 KiB piece of (nb, step) is lane-linear; wave w moves KiB 4w .. 4w+3
    const unsigned uoff = ((unsigned)nb * (unsigned)a.nstep) * 32768u + (unsigned)wave * 4096u + (unsigned)lane * 16u;

    // fragment-read addresses (bytes).  Tile (ty, tx) = (wave, l16); pixel (2ty+ai, 2tx+bi) sits at q*64 with
    // q = (2ty+ai)*34 + 2tx+bi, physical chunk kq ^ ((tx + (bi>>1)) & 3): two address registers, the rest immediates.
    unsigned raddr[2];
#pragma unroll
    for (int hj = 0; hj < 2; ++hj)
        raddr[hj] = (unsigned)((2 * wave * WPW + 2 * l16) * 64 + ((kq ^ ((l16 + hj) & 3)) << 4));
    unsigned uaddr = (unsigned)(2 * WRAW_B + kq * 512 + l16 * 16);
    // opaque to the optimiser: keeps ONE base register + 16-bit immediates (xi*2048 + nt*256 + stage*32768 < 65536) for the
    // 64 filter-fragment reads instead of one hoisted address register each
    asm volatile("" : "+v"(uaddr));

    f32x4 acc[16][2];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][nt][r] = 0.f;

    // DMA number idx_ (0..8) of this wave into `stage`: 0..4 = raw-patch pieces (per-lane offsets ro_[]), 5..8 = filter
    // pieces (per-lane offset uo_, step offset us_ in an SGPR).  A step that has no successor still issues them, with
    // out-of-range offsets: the hardware then writes zeros into the stage nobody reads -- no branches in the loop.
#define WINO_DMA_ONE(stage, idx_)                                                                         \
    {                                                                                                     \
        if ((idx_) < 5) {                                                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)(smem + (stage) * WRAW_B + (wave + 8 * (idx_)) * 1024), \
                                                     16, ro_[(idx_) < 5 ? (idx_) : 0], 0, 0, 0);          \
        } else {                                                                                          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lds_void*)(smem + 2 * WRAW_B + (stage) * WU_B + (wave * 4 + (idx_) - 5) * 1024), \
                                                     16, uo_, us_ + (unsigned)((idx_) - 5) * 1024u, 0, 0);    \
        }                                                                                                 \
    }
#define WINO_DMA_SETUP(s, DO)                                                                             \
        unsigned ro_[5];                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 5; ++i_) ro_[i_] = (DO) ? roff[i_] + (unsigned)(s) * 64u : WOOB; \
        const unsigned uo_ = (DO) ? uoff : WOOB;                                                          \
        const unsigned us_ = (DO) ? (unsigned)(s) * 32768u : 0u;
#define WINO_DMA(stage)                                                                                   \
    { _Pragma("unroll") for (int i_ = 0; i_ < 9; ++i_) WINO_DMA_ONE(stage, i_); }

    // 4x4 input transform of one xi row: t = (B^T d)[i][*], v = t B
#define WINO_ROW(i)                                                                                       \
            f32x4 t_[4], v_[4];                                                                           \
            if (PROBE & 1) {                                                                              \
                _Pragma("unroll") for (int bi = 0; bi < 4; ++bi) v_[bi] = d_[i][bi];                      \
            } else if (!(PROBE & 4)) {                                                                    \
                _Pragma("unroll") for (int bi = 0; bi < 4; ++bi)                                          \
                    t_[bi] = i == 0 ? pk_sub(d_[0][bi], d_[2][bi]) : i == 1 ? pk_add(d_[1][bi], d_[2][bi]) \
                           : i == 2 ? pk_sub(d_[2][bi], d_[1][bi]) : pk_sub(d_[1][bi], d_[3][bi]);        \
                v_[0] = pk_sub(t_[0], t_[2]); v_[1] = pk_add(t_[1], t_[2]); v_[2] = pk_sub(t_[2], t_[1]); \
                v_[3] = pk_sub(t_[1], t_[3]);                                                             \
                asm volatile("s_nop 1" : "+v"(v_[0]), "+v"(v_[1]), "+v"(v_[2]), "+v"(v_[3]));             \
            } else {                                                                                      \
                _Pragma("unroll") for (int bi = 0; bi < 4; ++bi)                                          \
                    t_[bi] = i == 0 ? d_[0][bi] - d_[2][bi] : i == 1 ? d_[1][bi] + d_[2][bi]              \
                           : i == 2 ? d_[2][bi] - d_[1][bi] : d_[1][bi] - d_[3][bi];                      \
                v_[0] = t_[0] - t_[2]; v_[1] = t_[1] + t_[2]; v_[2] = t_[2] - t_[1]; v_[3] = t_[1] - t_[3]; \
            }

    // one 16-channel step on stage STG; the next step's nine DMAs (into stage STG^1) are issued one at a time behind the
    // MFMA groups of xi 0..8 (all at the top of the step: 7.64 ms instead of 7.10 on res2 -- they stall the step's head)
#define WINO_COMPUTE(STG)                                                                                 \
    {                                                                                                     \
        const char* rb_ = smem + (STG) * WRAW_B;                                                          \
        const char* ub_ = smem + uaddr + (STG) * WU_B;                                                    \
        f32x4 d_[4][4];                                                                                   \
        _Pragma("unroll") for (int ai = 0; ai < 4; ++ai)                                                  \
            _Pragma("unroll") for (int bi = 0; bi < 4; ++bi)                                              \
                d_[ai][bi] = *reinterpret_cast<const f32x4*>(rb_ + raddr[bi >> 1] + (ai * WPW + bi) * 64); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                   \
            WINO_ROW(i)                                                                                   \
            _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                            \
                const f32x4 b0_ = *reinterpret_cast<const f32x4*>(ub_ + (i * 4 + jj) * 2048);             \
                const f32x4 b1_ = *reinterpret_cast<const f32x4*>(ub_ + (i * 4 + jj) * 2048 + 256);       \
                _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                        \
                    acc[i * 4 + jj][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v_[jj][s_], b0_[s_], acc[i * 4 + jj][0], 0, 0, 0); \
                    acc[i * 4 + jj][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v_[jj][s_], b1_[s_], acc[i * 4 + jj][1], 0, 0, 0); \
                }                                                                                         \
                if (!(PROBE & 10) && i * 4 + jj < 9) WINO_DMA_ONE((STG) ^ 1, i * 4 + jj);                 \
            }                                                                                             \
        }                                                                                                 \
    }
#define WINO_SYNC()                                                                                       \
    {                                                                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
        __syncthreads();                                                                                  \
    }

    {
        WINO_DMA_SETUP(0, true);
        WINO_DMA(0);
    }
    WINO_SYNC();
    for (int s = 0; s < a.nstep; s += 2) {
        const bool m1_ = s + 1 < a.nstep, m2_ = s + 2 < a.nstep;
        {
            WINO_DMA_SETUP(s + 1, m1_);
            if ((PROBE & 10) == 8) WINO_DMA(1);
            WINO_COMPUTE(0);
        }
        WINO_SYNC();
        if (m1_) {
            WINO_DMA_SETUP(s + 2, m2_);
            if ((PROBE & 10) == 8) WINO_DMA(0);
            WINO_COMPUTE(1);
            WINO_SYNC();
        }
    }
#undef WINO_DMA_SETUP
#undef WINO_ROW
#undef WINO_DMA_ONE
#undef WINO_SYNC
#undef WINO_COMPUTE
#undef WINO_DMA

    // epilogue.  C/D layout of the 16x16 MFMA: col = lane&15 (channel), row = 4*(lane>>4) + r = the tile's tx.
    // Y = A^T M A with A^T = [[1,1,1,0],[0,1,-1,-1]], M[i][j] = acc[4i+j].
    const int ty = wave;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = nb * 32 + nt * 16 + l16;
        const float bv = a.bias ? a.bias[n] : 0.f;
        const float av = a.alpha ? a.alpha[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s_[4][2];                  // column transform of every xi row: M[i][*] A
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s_[i][0] = (acc[i * 4 + 0][nt][r] + acc[i * 4 + 1][nt][r]) + acc[i * 4 + 2][nt][r];
                s_[i][1] = (acc[i * 4 + 1][nt][r] - acc[i * 4 + 2][nt][r]) - acc[i * 4 + 3][nt][r];
            }
            const int tx = 4 * kq + r;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int oy = by * 16 + 2 * ty + dy, ox = bx * 32 + 2 * tx + dx;
                    if (oy < a.H && ox < a.W) {
                        float v = dy == 0 ? (s_[0][dx] + s_[1][dx]) + s_[2][dx] : (s_[1][dx] - s_[2][dx]) - s_[3][dx];
                        v += bv;
                        const size_t oo = ((size_t)(b * a.H + oy) * a.W + ox) * a.Cout + n;
                        if (a.z) a.z[oo] = v;
                        if (a.act & RN_ACT_PRELU) v = fmaxf(v, 0.f) + av * fminf(v, 0.f);
                        if (a.act & RN_ACT_ELU) v = v > 0.f ? v : expf(v) - 1.f;
                        if (a.res) v += a.res[oo];
                        if (a.act & RN_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                        a.y[oo] = v;
                    }
                }
        }
    }
}

bool rn_wino_supported(int Cin, int Cout)
{
    static const bool off = getenv("RN_NO_WINOGRAD") != nullptr;
    return !off && Cin % 16 == 0 && Cout % 32 == 0;
}

// x [B,H,W,Cin] -> y [B,H,W,Cout], 3x3 stride 1 SAME; u from rn_pack_weights(RN_PACK_CONV_WINO | RN_PACK_CONVT_S1_WINO)
int rn_launch_conv_wino(const float* x, const float* u, const float* bias, const float* alpha, const float* residual,
                        float* y, float* preact, int B, int H, int W, int Cin, int Cout, int act, hipStream_t st)
{
    if (Cin % 16 != 0 || Cout % 32 != 0)
        return rn_set_error(RN_E_UNSUPPORTED, "conv_wino: Cin=%d (need %%16) Cout=%d (need %%32)", Cin, Cout);
    const long long per_item = (long long)H * W * Cin * 4;
    if (per_item >= 0x80000000LL)
        return rn_set_error(RN_E_UNSUPPORTED, "conv_wino: one batch item of %lld bytes exceeds the 2 GiB buffer window", per_item);
    if (per_item * B >= 0x80000000LL) {
        // 32-bit byte offsets with the upper half reserved for the hardware zero fill: batch chunks that fit the window
        const int chunk = (int)(0x7fffffffLL / per_item);
        const size_t ostep = (size_t)H * W * Cout;
        for (int b0 = 0; b0 < B; b0 += chunk) {
            const int nbi = B - b0 < chunk ? B - b0 : chunk;
            const int rc = rn_launch_conv_wino(x + (size_t)b0 * (per_item / 4), u, bias, alpha,
                                               residual ? residual + b0 * ostep : nullptr, y + b0 * ostep,
                                               preact ? preact + b0 * ostep : nullptr, nbi, H, W, Cin, Cout, act, st);
            if (rc != RN_OK) return rc;
        }
        return RN_OK;
    }
    WinoArgs a;
    a.x = x; a.u = u; a.bias = bias; a.alpha = alpha; a.res = residual; a.y = y; a.z = preact;
    a.x_bytes = (unsigned)(per_item * B);
    const long long ub = 16LL * Cin * Cout * 4;
    if (ub >= 0x80000000LL) return rn_set_error(RN_E_UNSUPPORTED, "conv_wino: transformed filter of %lld bytes exceeds 2 GiB", ub);
    a.u_bytes = (unsigned)ub;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.bh = (H + 15) / 16; a.bw = (W + 31) / 32;
    const long long mbl = (long long)B * a.bh * a.bw;
    a.nblocks = Cout / 32;
    if (mbl * a.nblocks > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv_wino: grid too large");
    a.mblocks = (int)mbl;
    a.nstep = Cin / 16;
    a.act = act;
    static const int probe = getenv("RN_WINO_PROBE") ? atoi(getenv("RN_WINO_PROBE")) : 0;
    const size_t lds = rn_wino_lds_bytes();
    auto kern = probe == 1 ? conv_wino_kernel<1> : probe == 2 ? conv_wino_kernel<2> : probe == 3 ? conv_wino_kernel<3>
              : probe == 4 ? conv_wino_kernel<4> : probe == 8 ? conv_wino_kernel<8> : probe == 12 ? conv_wino_kernel<12>
              : conv_wino_kernel<0>;
    // per launch: the attribute is per device, and a process may drive several
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.mblocks * a.nblocks)), dim3(512), lds, st, a);
    return rn_check_launch("conv_wino");
}
