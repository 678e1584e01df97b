// 3x3x3 stride-1 SAME convolution of the RenderNet 3-D encoder (21 x 32->32, e_conv3 16->32;
// RenderNet_Shader.py:44-64, tools/layer_util.py:60-73,228-265) on the fp32 matrix cores, with the
// depth halo kept in LDS.
//
// Why a second conv kernel: with N = Cout = 32 the generic implicit GEMM (conv_igemm.hip) re-stages
// every A row once per tap for only 32 output channels -- 16 MFMAs per wave per barrier.  Here the
// GEMM rows of a workgroup are 4 "columns" (b,h,w) x 32 consecutive depth positions.  In
// channels-last [B,H,W,D,C] a column is ONE contiguous run of D*C floats, so for a filter offset
// (t0,t1) the 34-row slab (depth halo of 1 on both sides) of each column is loaded once -- 1 KiB
// contiguous per wave instruction -- and serves the three depth taps t2 = 0,1,2 by a row shift of the
// LDS read address.  Per barrier a wave issues 48 MFMAs instead of 16 and the global->LDS traffic
// per MFMA drops 3x.  Halo rows beyond [0,D) and SAME-padding columns are zero-filled by the buffer
// bounds check (offset >= 2^31 -> 0).  Epilogue: bias -> PReLU -> residual, as in conv_igemm.
#include "rn_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct DrunArgs {
    const float* x; const float* w; const float* bias; const float* alpha; const float* res; float* y; float* z;
    unsigned x_bytes, w_bytes;
    int ncols;            // B*H*W
    int H, W, D, Cout;
    long long os_b, os0, os1, os2, out_off;
    int act;
};

template <int CIN>
__global__ __launch_bounds__(256, 2)
void conv3d_k3_drun_kernel(const DrunArgs a)
{
    constexpr int COLS = 4, TD = 32, ROWS = TD + 2;
    constexpr int LDA = CIN + 4;                         // padded row (floats)
    constexpr int TPR = CIN / 4;                         // float4 per row
    constexpr int AF4 = COLS * ROWS * TPR;               // float4 per A stage
    constexpr int APASS = (AF4 + 255) / 256;
    constexpr int BF4 = 3 * (CIN / 4) * 32;              // float4 per B stage (3 depth taps)
    constexpr int BPT = (BF4 + 255) / 256;
    constexpr int ASZ = COLS * ROWS * LDA;               // floats
    constexpr int BSZ = BF4 * 4;
    constexpr unsigned OOB = 0x80000000u;

    __shared__ __attribute__((aligned(16))) float smem[2 * (ASZ + BSZ)];
    __shared__ unsigned colbase[2][9][COLS];             // byte offset of the source column per (t0,t1), or OOB
    float* As = smem;
    float* Bs = smem + 2 * ASZ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    // Persistent workgroups: work item = (group of 4 columns, 32-deep depth chunk).  XCD-aware and
    // contiguous: block b runs on XCD b%8 and walks a contiguous eighth of the items, so the (h+-1, w+-1)
    // halo columns of its neighbours in time are found in that XCD's L2.
    const int ndch = (a.D + TD - 1) / TD;
    const int nitems = ((a.ncols + COLS - 1) / COLS) * ndch;
    const int nslots = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_xcd = (nitems + 7) >> 3;
    const int nit = (slot < per_xcd) ? (per_xcd - slot + nslots - 1) / nslots : 0;
#define DRUN_ITEM(i) (xcd * per_xcd + (i) * nslots + slot)
    if (nit == 0) return;

    // per-thread A element assignment (fixed): (column, row, channel group)
    int acol[APASS], arow[APASS], arc[APASS], alds[APASS];
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
        const int idx = tid + p * 256;
        const int c = idx / (ROWS * TPR), rem = idx % (ROWS * TPR);
        const int row = rem / TPR, cg = rem % TPR;
        acol[p] = (idx < AF4) ? c : 0;
        arow[p] = (idx < AF4) ? row : -0x40000000;       // invalid slot -> always out of range
        arc[p] = (row * CIN + cg * 4) * 4;
        alds[p] = (c * ROWS + row) * LDA + cg * 4;
    }
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);

#define DRUN_SETUP_COLS(cbuf, item)                                                                   \
    if (tid < 9 * COLS) {                                                                             \
        const int step_ = tid / COLS, c_ = tid % COLS;                                                \
        const int t0_ = step_ / 3, t1_ = step_ % 3;                                                   \
        const int col_ = ((item) / ndch) * COLS + c_;                                                 \
        unsigned off_ = OOB;                                                                          \
        if ((item) < nitems && col_ < a.ncols) {                                                      \
            const int w_ = col_ % a.W, h_ = (col_ / a.W) % a.H, b_ = col_ / (a.W * a.H);             \
            const int hh_ = h_ + t0_ - 1, ww_ = w_ + t1_ - 1;                                         \
            if ((unsigned)hh_ < (unsigned)a.H && (unsigned)ww_ < (unsigned)a.W)                       \
                off_ = (unsigned)((b_ * a.H + hh_) * a.W + ww_) * (unsigned)(a.D * CIN * 4);          \
        }                                                                                             \
        colbase[cbuf][step_][c_] = off_;                                                              \
    }

    u32x4 ra[APASS];
    f32x4 rb[BPT];
    // global -> registers: slab rows d0-1 .. d0+32 of the 4 source columns of filter offset `step`
#define DRUN_GLOAD(cbuf, step, d0_)                                                                   \
    {                                                                                                 \
        _Pragma("unroll") for (int p = 0; p < APASS; ++p) {                                           \
            const unsigned cb = colbase[cbuf][step][acol[p]];                                         \
            const int dd = (d0_) - 1 + arow[p];                                                       \
            const unsigned ro = (unsigned)(((d0_) - 1) * (CIN * 4) + arc[p]);                         \
            const unsigned off = ((cb & OOB) || (unsigned)dd >= (unsigned)a.D) ? OOB : cb + ro;       \
            ra[p] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0);                          \
        }                                                                                             \
        const float* wk = a.w + (size_t)(step) * BSZ;                                                 \
        _Pragma("unroll") for (int i = 0; i < BPT; ++i)                                               \
            if (BF4 % 256 == 0 || tid + i * 256 < BF4)                                                \
                rb[i] = *reinterpret_cast<const f32x4*>(wk + (tid + i * 256) * 4);                    \
    }
#define DRUN_LSTORE(buf)                                                                              \
    {                                                                                                 \
        float* Ab_ = As + (buf) * ASZ;                                                                \
        float* Bb_ = Bs + (buf) * BSZ;                                                                \
        _Pragma("unroll") for (int p = 0; p < APASS; ++p)                                             \
            if (p < APASS - 1 || tid + p * 256 < AF4) *reinterpret_cast<u32x4*>(Ab_ + alds[p]) = ra[p]; \
        _Pragma("unroll") for (int i = 0; i < BPT; ++i)                                               \
            if (BF4 % 256 == 0 || tid + i * 256 < BF4)                                                \
                *reinterpret_cast<f32x4*>(Bb_ + (tid + i * 256) * 4) = rb[i];                         \
    }

    const int n = li;
    const bool nok = n < a.Cout;
    const float bv = (a.bias && nok) ? a.bias[n] : 0.f;
    const float av = (a.alpha && nok) ? a.alpha[n] : 0.f;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    {
        const int item0 = DRUN_ITEM(0);
        DRUN_SETUP_COLS(0, item0);
        __syncthreads();
        DRUN_GLOAD(0, 0, (item0 % ndch) * TD);
        DRUN_LSTORE(0);
        __syncthreads();
    }

    int cur = 0;
    for (int i = 0; i < nit; ++i) {
        const int item = DRUN_ITEM(i);
        const int d0 = (item % ndch) * TD;
        const bool has_next = i + 1 < nit;
        const int item_n = DRUN_ITEM(i + 1);
        if (has_next) DRUN_SETUP_COLS((i + 1) & 1, item_n);     // first read at step 8, >= 8 barriers later
#pragma nounroll
        for (int step = 0; step < 9; ++step) {
            if (step < 8) DRUN_GLOAD(i & 1, step + 1, d0)
            else if (has_next) DRUN_GLOAD((i + 1) & 1, 0, (item_n % ndch) * TD)
            // keep the prefetch at the top of the step: hipcc otherwise sinks the loads down to their
            // first use (the LDS store), exposing the full L2 latency in front of the barrier
            __builtin_amdgcn_sched_barrier(0);
            const bool more = step < 8 || has_next;
            const float* Ab = As + cur * ASZ + (wave * ROWS + li) * LDA + lh * 4;
            const float* Bb = Bs + cur * BSZ + (lh * 32 + li) * 4;
            // fragments of depth tap t2+1 are read while the MFMAs of tap t2 run
            f32x4 fa[2][CIN / 8], fb[2][CIN / 8];
#pragma unroll
            for (int kb = 0; kb < CIN / 8; ++kb) {
                fa[0][kb] = *reinterpret_cast<const f32x4*>(Ab + kb * 8);
                fb[0][kb] = *reinterpret_cast<const f32x4*>(Bb + (kb * 2) * 32 * 4);
            }
#pragma unroll
            for (int t2 = 0; t2 < 3; ++t2) {
                if (t2 < 2) {
#pragma unroll
                    for (int kb = 0; kb < CIN / 8; ++kb) {
                        fa[(t2 + 1) & 1][kb] = *reinterpret_cast<const f32x4*>(Ab + (t2 + 1) * LDA + kb * 8);
                        fb[(t2 + 1) & 1][kb] = *reinterpret_cast<const f32x4*>(Bb + ((t2 + 1) * (CIN / 4) + kb * 2) * 32 * 4);
                    }
                }
                if (t2 == 2 && more) DRUN_LSTORE(cur ^ 1);
#pragma unroll
                for (int kb = 0; kb < CIN / 8; ++kb)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t2 & 1][kb][s], fb[t2 & 1][kb][s], acc, 0, 0, 0);
            }
            __syncthreads();
            cur ^= 1;
        }
        // epilogue of this item: wave <-> column, MFMA row <-> depth, MFMA col <-> output channel
        const int col = (item / ndch) * COLS + wave;
        if (item < nitems && col < a.ncols && nok) {
            const int w = col % a.W, h = (col / a.W) % a.H, b = col / (a.W * a.H);
            const long long cbase = a.out_off + b * a.os_b + h * a.os0 + w * a.os1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = d0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (d < a.D) {
                    const long long oo = cbase + d * a.os2 + n;
                    float v = acc[r] + bv;
                    if (a.z) a.z[oo] = v;
                    if (a.act & RN_ACT_PRELU) v = fmaxf(v, 0.f) + av * fminf(v, 0.f);
                    if (a.act & RN_ACT_ELU) v = v > 0.f ? v : expf(v) - 1.f;
                    if (a.res) v += a.res[oo];
                    if (a.act & RN_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                    a.y[oo] = v;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
#undef DRUN_GLOAD
#undef DRUN_LSTORE
#undef DRUN_SETUP_COLS
#undef DRUN_ITEM
}


// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (Cin = 32): slabs and filter taps go global -> LDS with buffer_load_dwordx4 ... lds
// (no staging VGPRs, no ds_write pass -- the same change took the 2-D kernel from 133 to 144 TFLOP/s).
// The A stage is the 136 slab rows (4 columns x 34 depth rows) stored unpadded, 128 B each; the DMA
// writes lane-linear, so LDS row rr holds logical 16-B chunk c at physical chunk c ^ ((rr>>1)&7)
// (swizzle applied to the per-lane SOURCE address and to the fragment read) -- any 16 consecutive-ish
// rows of a ds_read_b128 lane group then hit 16 distinct 16-B slots.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2)
void conv3d_k3_drun_dma_kernel(const DrunArgs a)
{
    constexpr int CIN = 32, COLS = 4, TD = 32, ROWS = TD + 2, NROWS = COLS * ROWS;   // 136 rows
    constexpr int ASZ = NROWS * CIN;                     // 4352 floats = 17 KiB = 17 DMA instructions
    constexpr int BSZ = 3 * (CIN / 4) * 32 * 4;          // 3072 floats = 12 KiB = 12 DMA instructions
    constexpr int NAI = NROWS / 8;                       // 17
    constexpr unsigned OOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void lds_void;

    __shared__ __attribute__((aligned(16))) float smem[2 * (ASZ + BSZ)];
    __shared__ unsigned colbase[2][9][COLS];
    float* As = smem;
    float* Bs = smem + 2 * ASZ;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    const int ndch = (a.D + TD - 1) / TD;
    const int nitems = ((a.ncols + COLS - 1) / COLS) * ndch;
    const int nslots = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_xcd = (nitems + 7) >> 3;
    const int nit = (slot < per_xcd) ? (per_xcd - slot + nslots - 1) / nslots : 0;
#define DRUN_ITEM(i) (xcd * per_xcd + (i) * nslots + slot)
    if (nit == 0) return;

    // DMA assignment: wave w issues A instructions q = w, w+4, w+8, w+12 (and 16 for wave 0); lane -> LDS row
    // rr = 8q + lane/8, physical chunk lane%8.
    int acol[5], arow[5]; unsigned arc[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const int q = wave + 4 * p;
        const int rr = q * 8 + (lane >> 3);
        const int c = rr / ROWS, r = rr - c * ROWS;
        acol[p] = (q < NAI) ? c : 0;
        arow[p] = (q < NAI) ? r : -0x40000000;
        arc[p] = (unsigned)(r * CIN * 4 + (((lane & 7) ^ ((rr >> 1) & 7)) * 16));
    }
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);

#define DRUN_SETUP_COLS(cbuf, item)                                                                   \
    if (tid < 9 * COLS) {                                                                             \
        const int step_ = tid / COLS, c_ = tid % COLS;                                                \
        const int t0_ = step_ / 3, t1_ = step_ % 3;                                                   \
        const int col_ = ((item) / ndch) * COLS + c_;                                                 \
        unsigned off_ = OOB;                                                                          \
        if ((item) < nitems && col_ < a.ncols) {                                                      \
            const int w_ = col_ % a.W, h_ = (col_ / a.W) % a.H, b_ = col_ / (a.W * a.H);             \
            const int hh_ = h_ + t0_ - 1, ww_ = w_ + t1_ - 1;                                         \
            if ((unsigned)hh_ < (unsigned)a.H && (unsigned)ww_ < (unsigned)a.W)                       \
                off_ = (unsigned)((b_ * a.H + hh_) * a.W + ww_) * (unsigned)(a.D * CIN * 4);          \
        }                                                                                             \
        colbase[cbuf][step_][c_] = off_;                                                              \
    }

#define DRUN_DMA(cbuf, step, d0_, stage)                                                              \
    {                                                                                                 \
        _Pragma("unroll") for (int p = 0; p < 5; ++p) {                                               \
            if (p < 4 || wave == 0) {                                                                 \
                const unsigned cb = colbase[cbuf][step][acol[p]];                                     \
                const int dd = (d0_) - 1 + arow[p];                                                   \
                const unsigned off = ((cb & OOB) || (unsigned)dd >= (unsigned)a.D)                    \
                                         ? OOB : cb + (unsigned)(((d0_) - 1) * (CIN * 4)) + arc[p];   \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                             \
                    xrsrc, (lds_void*)(As + (stage) * ASZ + (wave + 4 * p) * 256), 16, off, 0, 0, 0); \
            }                                                                                         \
        }                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                 \
                wrsrc, (lds_void*)(Bs + (stage) * BSZ + (wave * 3 + j) * 256), 16,                    \
                (unsigned)((step) * (BSZ * 4) + ((wave * 3 + j) * 64 + lane) * 16), 0, 0, 0);         \
    }

    const int n = li;
    const bool nok = n < a.Cout;
    const float bv = (a.bias && nok) ? a.bias[n] : 0.f;
    const float av = (a.alpha && nok) ? a.alpha[n] : 0.f;
    int swz[3];
#pragma unroll
    for (int t2 = 0; t2 < 3; ++t2) swz[t2] = ((wave * ROWS + li + t2) >> 1) & 7;

    // two accumulators (even / odd 8-k groups) break the MFMA dependency chain; summed in the epilogue
    f32x16 acc, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }

    {
        const int item0 = DRUN_ITEM(0);
        DRUN_SETUP_COLS(0, item0);
        __syncthreads();
        DRUN_DMA(0, 0, (item0 % ndch) * TD, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    int cur = 0;
    for (int i = 0; i < nit; ++i) {
        const int item = DRUN_ITEM(i);
        const int d0 = (item % ndch) * TD;
        const bool has_next = i + 1 < nit;
        const int item_n = DRUN_ITEM(i + 1);
        if (has_next) DRUN_SETUP_COLS((i + 1) & 1, item_n);
#pragma nounroll
        for (int step = 0; step < 9; ++step) {
            if (step < 8) DRUN_DMA(i & 1, step + 1, d0, cur ^ 1)
            else if (has_next) DRUN_DMA((i + 1) & 1, 0, (item_n % ndch) * TD, cur ^ 1)
            const float* Ab = As + cur * ASZ + (wave * ROWS + li) * CIN;
            const float* Bb = Bs + cur * BSZ + (lh * 32 + li) * 4;
            f32x4 fa[2][CIN / 8], fb[2][CIN / 8];
#pragma unroll
            for (int kb = 0; kb < CIN / 8; ++kb) {
                fa[0][kb] = *reinterpret_cast<const f32x4*>(Ab + ((kb * 2 + lh) ^ swz[0]) * 4);
                fb[0][kb] = *reinterpret_cast<const f32x4*>(Bb + (kb * 2) * 32 * 4);
            }
#pragma unroll
            for (int t2 = 0; t2 < 3; ++t2) {
                if (t2 < 2) {
#pragma unroll
                    for (int kb = 0; kb < CIN / 8; ++kb) {
                        fa[(t2 + 1) & 1][kb] = *reinterpret_cast<const f32x4*>(Ab + (t2 + 1) * CIN + ((kb * 2 + lh) ^ swz[t2 + 1]) * 4);
                        fb[(t2 + 1) & 1][kb] = *reinterpret_cast<const f32x4*>(Bb + ((t2 + 1) * (CIN / 4) + kb * 2) * 32 * 4);
                    }
                }
#pragma unroll
                for (int kb = 0; kb < CIN / 8; ++kb)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t2 & 1][kb][s], fb[t2 & 1][kb][s], acc1, 0, 0, 0);
                        else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t2 & 1][kb][s], fb[t2 & 1][kb][s], acc, 0, 0, 0);
                    }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
        const int col = (item / ndch) * COLS + wave;
        if (item < nitems && col < a.ncols && nok) {
            const int w = col % a.W, h = (col / a.W) % a.H, b = col / (a.W * a.H);
            const long long cbase = a.out_off + b * a.os_b + h * a.os0 + w * a.os1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = d0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (d < a.D) {
                    const long long oo = cbase + d * a.os2 + n;
                    float v = (acc[r] + acc1[r]) + bv;
                    if (a.z) a.z[oo] = v;
                    if (a.act & RN_ACT_PRELU) v = fmaxf(v, 0.f) + av * fminf(v, 0.f);
                    if (a.act & RN_ACT_ELU) v = v > 0.f ? v : expf(v) - 1.f;
                    if (a.res) v += a.res[oo];
                    if (a.act & RN_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                    a.y[oo] = v;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }
    }
#undef DRUN_DMA
#undef DRUN_SETUP_COLS
#undef DRUN_ITEM
}

bool rn_drun_supported(const RnConvProblem& p)
{
    if (p.K[0] != 3 || p.K[1] != 3 || p.K[2] != 3) return false;
    if (p.S[0] != 1 || p.S[1] != 1 || p.S[2] != 1) return false;
    if (p.P[0] != 1 || p.P[1] != 1 || p.P[2] != 1) return false;
    if (p.Cin != 16 && p.Cin != 32) return false;
    if (p.Npad != 32 || p.Cout < 16) return false;
    if (p.O[0] != p.I[0] || p.O[1] != p.I[1] || p.O[2] != p.I[2]) return false;
    const long long xb = (long long)p.B * p.I[0] * p.I[1] * p.I[2] * p.Cin * 4;
    return xb < 0x80000000LL;
}

int rn_launch_conv3d_drun(const RnConvProblem& p, hipStream_t st)
{
    if (!rn_drun_supported(p)) return rn_set_error(RN_E_UNSUPPORTED, "conv3d_drun: unsupported problem");
    DrunArgs a;
    a.x = p.x; a.w = p.w; a.bias = p.bias; a.alpha = p.alpha; a.res = p.residual; a.y = p.y; a.z = p.preact;
    a.x_bytes = (unsigned)((long long)p.B * p.I[0] * p.I[1] * p.I[2] * p.Cin * 4);
    a.w_bytes = (unsigned)((27 * p.Cin / 4) * 32 * 16);
    const long long ncols = (long long)p.B * p.I[0] * p.I[1];
    if (ncols > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv3d_drun: too many columns");
    a.ncols = (int)ncols;
    a.H = p.I[0]; a.W = p.I[1]; a.D = p.I[2]; a.Cout = p.Cout;
    a.os_b = p.os_b; a.os0 = p.os[0]; a.os1 = p.os[1]; a.os2 = p.os[2]; a.out_off = p.out_off;
    a.act = p.act;
    // persistent grid: 2 workgroups per CU (LDS-bound), a multiple of 8 (one slice per XCD)
    const long long nitems = ((ncols + 3) / 4) * ((p.I[2] + 31) / 32);
    unsigned nblk = (unsigned)((nitems + 7) / 8 * 8 < 512 ? (nitems + 7) / 8 * 8 : 512);
    dim3 grid(nblk);
    static const bool no_dma = getenv("RN_DRUN_NO_DMA") != nullptr;
    if (p.Cin == 32 && !no_dma) hipLaunchKernelGGL(conv3d_k3_drun_dma_kernel, grid, dim3(256), 0, st, a);
    else if (p.Cin == 32) hipLaunchKernelGGL(conv3d_k3_drun_kernel<32>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv3d_k3_drun_kernel<16>, grid, dim3(256), 0, st, a);
    return rn_check_launch("conv3d_drun");
}
