// Implicit-GEMM convolution on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// One kernel serves every conv of the RenderNet path with Cin % 16 == 0 (3-D convs of the
// encoder, the projection unit's 1x1, all 2-D convs, transposed convs rewritten as forward
// convs / sub-pixel phases).  GEMM view: M = B*O0*O1*O2 output positions, N = Cout,
// K = taps*Cin with k = tap*Cin + c.  Replaces tf.nn.conv3d / tf.nn.conv2d / slim.conv2d /
// conv2d_transpose call sites tools/layer_util.py:171,212,253,295 and RenderNet_Shader.py:83-129.
//
// Structure (per workgroup, BM x BN output tile, K walked in BK-channel slices of one tap):
//   global --(16 B/lane, zero-filled where SAME padding applies)--> registers --> LDS (2 stages)
//   LDS --ds_read_b128--> MFMA fragments: a lane reads 4 consecutive k of its row/column and
//   feeds them to 4 successive 32x32x2 MFMAs (the k permutation is the same for A and B).
//   A rows are padded by 4 floats (row stride 36 or 20 dwords) so that the 16-lane groups of a
//   ds_read_b128 hit 16 distinct 16-B slots; B is stored [k/4][n][4] so lanes read consecutive
//   16-B slots.  Epilogue (bias, PReLU, residual, sigmoid) is applied on the accumulators.
//   blockIdx -> tile mapping is XCD-aware: each XCD (private 4 MiB L2) walks a contiguous run of
//   tiles with the N tiles of one M tile adjacent, so A slices are fetched once per XCD.
#include "rn_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct IgemmArgs {
    const float* x; const float* w; const float* bias; const float* alpha; const float* res; float* y; float* z;
    int M;
    unsigned x_bytes, w_bytes;
    int I0, I1, I2, Cin;
    int O0, O1, O2, Cout, Npad;
    int K0, K1, K2, S0, S1, S2, P0, P1, P2;
    long long os_b, os0, os1, os2, out_off;
    int act, ctiles, nk, mtiles, ntiles;
};

template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 2)
void conv_igemm_kernel(const IgemmArgs a)
{
    constexpr int NT = 64 * WM * WN;
    constexpr int LDA = BK + 4;                 // padded A row (floats)
    constexpr int WTM = BM / WM, WTN = BN / WN; // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32; // 32x32 MFMA tiles per wave
    constexpr int TPR = BK / 4;                 // threads per A row (one float4 each)
    constexpr int RPP = NT / TPR;               // A rows per pass
    constexpr int APASS = BM / RPP;
    constexpr int BF4 = (BK / 4) * BN;          // float4s in a B tile
    constexpr int BPT = (BF4 + NT - 1) / NT;
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && BM % RPP == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                   // [2][BM][LDA]
    float* Bs = As + 2 * BM * LDA;                                // [2][BK/4][BN][4]
    int4* rowinfo = reinterpret_cast<int4*>(Bs + 2 * BK * BN);    // [BM] {b, in0, in1, in2}
    long long* outoff = reinterpret_cast<long long*>(rowinfo + BM); // [BM]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware, bijective block -> tile map (block b runs on XCD b % 8)
    int tile;
    {
        const int nb = a.mtiles * a.ntiles, id = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = id & 7, within = id >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int tn_idx = tile % a.ntiles, tm_idx = tile / a.ntiles;
    const int m0 = tm_idx * BM, n0 = tn_idx * BN;

    // per-row metadata (output position -> input origin, output offset)
    for (int r = tid; r < BM; r += NT) {
        const int m = m0 + r;
        int4 ri; long long oo = -1;
        if (m < a.M) {
            int t = m;
            const int o2 = t % a.O2; t /= a.O2;
            const int o1 = t % a.O1; t /= a.O1;
            const int o0 = t % a.O0; const int b = t / a.O0;
            ri = make_int4(b, o0 * a.S0 - a.P0, o1 * a.S1 - a.P1, o2 * a.S2 - a.P2);
            oo = a.out_off + b * a.os_b + o0 * a.os0 + o1 * a.os1 + o2 * a.os2;
        } else {
            ri = make_int4(-1, 0, 0, 0);
        }
        rowinfo[r] = ri; outoff[r] = oo;
    }
    __syncthreads();

    const int arow = tid / TPR, acg = tid % TPR;
    int rb[APASS], r0[APASS], r1[APASS], r2[APASS];
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
        const int4 ri = rowinfo[arow + p * RPP];
        rb[p] = ri.x; r0[p] = ri.y; r1[p] = ri.z; r2[p] = ri.w;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A is fetched through a buffer resource: SAME-padding taps and rows past M get a byte offset
    // >= 2^31 (beyond num_records), for which the hardware returns zeros -- no branches, no selects.
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // K-walk state: tap (t0,t1,t2) and channel slice
    int t0 = 0, t1 = 0, t2 = 0, ct = 0;
    unsigned aoff[APASS];
#define RN_TAP_SETUP()                                                                                   \
    _Pragma("unroll") for (int p = 0; p < APASS; ++p) {                                                  \
        const int i0 = r0[p] + t0, i1 = r1[p] + t1, i2 = r2[p] + t2;                                     \
        const bool ok = rb[p] >= 0 && (unsigned)i0 < (unsigned)a.I0 && (unsigned)i1 < (unsigned)a.I1 &&  \
                        (unsigned)i2 < (unsigned)a.I2;                                                   \
        const unsigned e = (unsigned)(((rb[p] * a.I0 + i0) * a.I1 + i1) * a.I2 + i2) * (unsigned)a.Cin;  \
        aoff[p] = ok ? (e + acg * 4) * 4u : OOB;                                                         \
    }
    RN_TAP_SETUP();

    // register staging set: the loads of K-tile kt+1 are in flight during the MFMAs of tile kt
    u32x4 ra0[APASS];
    f32x4 rb0[BPT];
    // global -> registers for K-tile kt (then advance the K-walk)
#define RN_GLOAD(kt, RA, RB)                                                                             \
    {                                                                                                    \
        const unsigned c0b = (unsigned)ct * (BK * 4);                                                    \
        _Pragma("unroll") for (int p = 0; p < APASS; ++p)                                                \
            RA[p] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, aoff[p] + c0b, 0, 0);                   \
        const float* wk = a.w + ((size_t)(kt) * (BK / 4) * a.Npad + n0) * 4;                             \
        _Pragma("unroll") for (int i = 0; i < BPT; ++i) {                                                \
            const int idx = tid + i * NT;                                                                \
            if (BF4 % NT == 0 || idx < BF4)                                                              \
                RB[i] = *reinterpret_cast<const f32x4*>(wk + ((size_t)(idx / BN) * a.Npad + idx % BN) * 4); \
        }                                                                                                \
        if (++ct == a.ctiles) {                                                                          \
            ct = 0;                                                                                      \
            if (++t2 == a.K2) { t2 = 0; if (++t1 == a.K1) { t1 = 0; ++t0; } }                            \
            RN_TAP_SETUP();                                                                              \
        }                                                                                                \
    }
    // registers -> LDS stage `buf`
#define RN_LSTORE(buf, RA, RB)                                                                           \
    {                                                                                                    \
        float* Ab_ = As + (buf) * BM * LDA;                                                              \
        float* Bb_ = Bs + (buf) * BK * BN;                                                               \
        _Pragma("unroll") for (int p = 0; p < APASS; ++p)                                                \
            *reinterpret_cast<u32x4*>(Ab_ + (arow + p * RPP) * LDA + acg * 4) = RA[p];                   \
        _Pragma("unroll") for (int i = 0; i < BPT; ++i) {                                                \
            const int idx = tid + i * NT;                                                                \
            if (BF4 % NT == 0 || idx < BF4) *reinterpret_cast<f32x4*>(Bb_ + idx * 4) = RB[i];           \
        }                                                                                                \
    }
    // MFMAs of the K-tile in LDS stage `cur`; STORE_STMT (the next tile's ds_writes) runs before the last 8-k group,
    // so that the writes issue under the MFMAs
#define RN_COMPUTE(STORE_STMT)                                                                           \
    {                                                                                                    \
        const float* Ab = As + cur * BM * LDA + (wm * WTM + li) * LDA + lh * 4;                          \
        const float* Bb = Bs + cur * BK * BN + (lh * BN + wn * WTN + li) * 4;                            \
        _Pragma("unroll") for (int kb = 0; kb < BK / 8; ++kb) {                                          \
            if (kb == BK / 8 - 1) { STORE_STMT; }                                                        \
            f32x4 af[TM], bf[TN];                                                                        \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                               \
                af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDA + kb * 8);                     \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                               \
                bf[j] = *reinterpret_cast<const f32x4*>(Bb + (kb * 2 * BN + j * 32) * 4);                \
            _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                           \
                    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                       \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0); \
        }                                                                                                \
    }

    int cur = 0;
    RN_GLOAD(0, ra0, rb0);
    RN_LSTORE(0, ra0, rb0);
    __syncthreads();
    for (int kt = 0; kt < a.nk; ++kt) {
        const bool more = kt + 1 < a.nk;
        if (more) RN_GLOAD(kt + 1, ra0, rb0);
        RN_COMPUTE(if (more) RN_LSTORE(cur ^ 1, ra0, rb0));
        __syncthreads();
        cur ^= 1;
    }
#undef RN_COMPUTE
#undef RN_TAP_SETUP
#undef RN_GLOAD
#undef RN_LSTORE

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + li;
        const bool nok = n < a.Cout;
        const float bv = (a.bias && nok) ? a.bias[n] : 0.f;
        const float av = (a.alpha && nok) ? a.alpha[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const long long oo = outoff[row];
                if (oo >= 0 && nok) {
                    float v = acc[i][j][r] + bv;
                    if (a.z) a.z[oo + n] = v;
                    if (a.act & RN_ACT_PRELU) v = fmaxf(v, 0.f) + av * fminf(v, 0.f);
                    if (a.act & RN_ACT_ELU) v = v > 0.f ? v : expf(v) - 1.f;
                    if (a.res) v += a.res[oo + n];
                    if (a.act & RN_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                    a.y[oo + n] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (128x128x32 tile): both operands go global -> LDS directly with
// buffer_load_dwordx4 ... lds (1 KiB per wave instruction), no staging VGPRs, no ds_write pass.
// The DMA writes lane-linear (wave base + lane*16 B), so the A tile is stored UNPADDED ([128][32] f32,
// 128-B rows) and bank conflicts are avoided by an XOR swizzle of the 16-B chunk index applied on the
// SOURCE side (lane (row, pc) fetches logical chunk pc ^ ((row>>1)&7)) and again on the fragment read;
// the map makes the 16 rows of every ds_read_b128 lane group hit 16 distinct 16-B slots.  SAME padding
// still comes from the buffer bounds check (offset >= 2^31 -> zeros are written to LDS).
// ------------------------------------------------------------------------------------------------
// BM = 128: the throughput tile.  BM = 64: same kernel with half-height tiles for small batches -- at M = 4096 (one
// frame) the 128-row tiling yields 256 workgroups for 512 slots.
template <int BM>
__global__ __launch_bounds__(256, 2)
void conv_igemm_glds_kernel(const IgemmArgs a)
{
    constexpr int BN = 128, BK = 32, NT = 256, WN = 2;
    constexpr int WTM = BM / 2, WTN = 64, TM = WTM / 32, TN = 2;
    constexpr int APW = BM / 32;                                  // A DMA instructions per wave (8 rows each)
    constexpr int ASZ = BM * BK, BSZ = BK * BN;                 // floats per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                   // [2][128][32]
    float* Bs = As + 2 * ASZ;                                     // [2][8][128][4]
    int4* rowinfo = reinterpret_cast<int4*>(Bs + 2 * BSZ);
    long long* outoff = reinterpret_cast<long long*>(rowinfo + BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    int tile;
    {
        const int nb = a.mtiles * a.ntiles, id = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = id & 7, within = id >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int tn_idx = tile % a.ntiles, tm_idx = tile / a.ntiles;
    const int m0 = tm_idx * BM, n0 = tn_idx * BN;

    for (int r = tid; r < BM; r += NT) {
        const int m = m0 + r;
        int4 ri; long long oo = -1;
        if (m < a.M) {
            int t = m;
            const int o2 = t % a.O2; t /= a.O2;
            const int o1 = t % a.O1; t /= a.O1;
            const int o0 = t % a.O0; const int b = t / a.O0;
            ri = make_int4(b, o0 * a.S0 - a.P0, o1 * a.S1 - a.P1, o2 * a.S2 - a.P2);
            oo = a.out_off + b * a.os_b + o0 * a.os0 + o1 * a.os1 + o2 * a.os2;
        } else {
            ri = make_int4(-1, 0, 0, 0);
        }
        rowinfo[r] = ri; outoff[r] = oo;
    }
    __syncthreads();

    // DMA assignment: wave w moves A rows [8*APW*w, +8*APW) as APW instructions of 8 rows; lane -> (row, chunk)
    // (fixed extents: with the dependent extent [APW] clang 22 silently drops the HOST stub of the kernel)
    static_assert(APW <= 4, "A DMA instructions per wave");
    int rb[4], r0[4], r1[4], r2[4];
    unsigned lc16[4];                                            // logical chunk byte offset within the row
#pragma unroll
    for (int p = 0; p < APW; ++p) {
        const int row = wave * (8 * APW) + p * 8 + (lane >> 3);
        const int4 ri = rowinfo[row];
        rb[p] = ri.x; r0[p] = ri.y; r1[p] = ri.z; r2[p] = ri.w;
        lc16[p] = (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    int t0 = 0, t1 = 0, t2 = 0, ct = 0;
    unsigned aoff[4];
#define RN_TAP_SETUP_G()                                                                                 \
    _Pragma("unroll") for (int p = 0; p < APW; ++p) {                                                      \
        const int i0 = r0[p] + t0, i1 = r1[p] + t1, i2 = r2[p] + t2;                                     \
        const bool ok = rb[p] >= 0 && (unsigned)i0 < (unsigned)a.I0 && (unsigned)i1 < (unsigned)a.I1 &&  \
                        (unsigned)i2 < (unsigned)a.I2;                                                   \
        const unsigned e = (unsigned)(((rb[p] * a.I0 + i0) * a.I1 + i1) * a.I2 + i2) * (unsigned)a.Cin;  \
        aoff[p] = ok ? e * 4u + lc16[p] : OOB;                                                           \
    }
    RN_TAP_SETUP_G();
    // weights: wave w, instruction j moves float4 indices [(4w+j)*64, +64) of the [8][128] tile
    unsigned boff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = (wave * 4 + j) * 64 + lane;
        boff[j] = (unsigned)(((size_t)(idx >> 7) * a.Npad + n0 + (idx & 127)) * 16);
    }
    const unsigned bstep = (unsigned)((size_t)(BK / 4) * a.Npad * 16);   // bytes per K-tile in the packed filter

    typedef __attribute__((address_space(3))) void lds_void;
#define RN_DMA(kt, stage)                                                                                \
    {                                                                                                    \
        const unsigned c0b = (unsigned)ct * (BK * 4);                                                    \
        _Pragma("unroll") for (int p = 0; p < APW; ++p)                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)(As + (stage) * ASZ + (wave * (8 * APW) + p * 8) * BK), \
                                                     16, aoff[p] + c0b, 0, 0, 0);                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                    \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void*)(Bs + (stage) * BSZ + (wave * 4 + j) * 256), \
                                                     16, boff[j] + (unsigned)(kt) * bstep, 0, 0, 0);     \
        if (++ct == a.ctiles) {                                                                          \
            ct = 0;                                                                                      \
            if (++t2 == a.K2) { t2 = 0; if (++t1 == a.K1) { t1 = 0; ++t0; } }                            \
            RN_TAP_SETUP_G();                                                                            \
        }                                                                                                \
    }

    RN_DMA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int swz = (li >> 1) & 7;
    int cur = 0;
    for (int kt = 0; kt < a.nk; ++kt) {
        if (kt + 1 < a.nk) RN_DMA(kt + 1, cur ^ 1);
        const float* Ab = As + cur * ASZ + (wm * WTM + li) * BK;
        const float* Bb = Bs + cur * BSZ + (lh * BN + wn * WTN + li) * 4;
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            f32x4 af[TM], bf[TN];
            const int ch = ((kb * 2 + lh) ^ swz) * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * BK + ch);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + (kb * 2 * BN + j * 32) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
#undef RN_DMA
#undef RN_TAP_SETUP_G

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + li;
        const bool nok = n < a.Cout;
        const float bv = (a.bias && nok) ? a.bias[n] : 0.f;
        const float av = (a.alpha && nok) ? a.alpha[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const long long oo = outoff[row];
                if (oo >= 0 && nok) {
                    float v = acc[i][j][r] + bv;
                    if (a.z) a.z[oo + n] = v;
                    if (a.act & RN_ACT_PRELU) v = fmaxf(v, 0.f) + av * fminf(v, 0.f);
                    if (a.act & RN_ACT_ELU) v = v > 0.f ? v : expf(v) - 1.f;
                    if (a.res) v += a.res[oo + n];
                    if (a.act & RN_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                    a.y[oo + n] = v;
                }
            }
        }
    }
}

template <int BM>
static int launch_glds(IgemmArgs& a, hipStream_t st)
{
    constexpr int BN = 128, BK = 32;
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = a.Npad / BN;
    a.ctiles = a.Cin / BK;
    a.nk = a.K0 * a.K1 * a.K2 * a.ctiles;
    const size_t lds = (size_t)2 * BM * BK * 4 + (size_t)2 * BK * BN * 4 + (size_t)BM * (16 + 8);
    auto kern = conv_igemm_glds_kernel<BM>;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(lds)); if (rc_ != RN_OK) return rc_; }
    const long long nb = (long long)a.mtiles * a.ntiles;
    if (nb <= 0 || nb > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv_igemm: bad grid %lld", nb);
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(256), lds, st, a);
    return rn_check_launch("conv_igemm_glds");
}

template <int BM, int BN, int BK, int WM, int WN>
static int launch_cfg(IgemmArgs& a, hipStream_t st)
{
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = a.Npad / BN;
    a.ctiles = a.Cin / BK;
    a.nk = a.K0 * a.K1 * a.K2 * a.ctiles;
    const size_t lds = (size_t)2 * BM * (BK + 4) * 4 + (size_t)2 * BK * BN * 4 + (size_t)BM * (16 + 8);
    auto kern = conv_igemm_kernel<BM, BN, BK, WM, WN>;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(lds)); if (rc_ != RN_OK) return rc_; }
    const long long nb = (long long)a.mtiles * a.ntiles;
    if (nb <= 0 || nb > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv_igemm: bad grid %lld", nb);
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(64 * WM * WN), lds, st, a);
    return rn_check_launch("conv_igemm");
}

bool rn_igemm_supported(const RnConvProblem& p)
{
    return p.Cin % 16 == 0 && p.Cout >= 16 && p.Npad % 32 == 0;
}

int rn_launch_conv_igemm(const RnConvProblem& p, hipStream_t st)
{
    if (!rn_igemm_supported(p)) return rn_set_error(RN_E_UNSUPPORTED, "conv_igemm: Cin=%d Cout=%d", p.Cin, p.Cout);
    {
        // The A operand is addressed with 32-bit byte offsets whose upper half is reserved for the
        // hardware zero-fill: inputs >= 2 GiB are processed in batch chunks that each fit the window.
        const long long per_item = (long long)p.I[0] * p.I[1] * p.I[2] * p.Cin * 4;
        if (per_item >= 0x80000000LL)
            return rn_set_error(RN_E_UNSUPPORTED, "conv_igemm: one batch item of %lld bytes exceeds the 2 GiB buffer window", per_item);
        if (per_item * p.B >= 0x80000000LL) {
            const int chunk = (int)(0x7fffffffLL / per_item);
            for (int b0 = 0; b0 < p.B; b0 += chunk) {
                RnConvProblem q = p;
                q.B = (p.B - b0 < chunk) ? p.B - b0 : chunk;
                q.x = p.x + (size_t)b0 * (per_item / 4);
                q.y = p.y + (size_t)b0 * p.os_b;
                if (p.residual) q.residual = p.residual + (size_t)b0 * p.os_b;
                if (p.preact) q.preact = p.preact + (size_t)b0 * p.os_b;
                const int rc = rn_launch_conv_igemm(q, st);
                if (rc != RN_OK) return rc;
            }
            return RN_OK;
        }
    }
    IgemmArgs a;
    a.x = p.x; a.w = p.w; a.bias = p.bias; a.alpha = p.alpha; a.res = p.residual; a.y = p.y; a.z = p.preact;
    const long long M = (long long)p.B * p.O[0] * p.O[1] * p.O[2];
    if (M <= 0 || M > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv_igemm: M=%lld", M);
    a.M = (int)M;
    const long long xb = (long long)p.B * p.I[0] * p.I[1] * p.I[2] * p.Cin * 4;
    if (xb >= 0x80000000LL) return rn_set_error(RN_E_UNSUPPORTED, "conv_igemm: input of %lld bytes exceeds the 2 GiB buffer window", xb);
    a.x_bytes = (unsigned)xb;
    {
        const long long wb = (long long)((p.K[0] * p.K[1] * p.K[2] * p.Cin + 3) / 4) * p.Npad * 16;
        a.w_bytes = wb < 0x80000000LL ? (unsigned)wb : 0u;
    }
    a.I0 = p.I[0]; a.I1 = p.I[1]; a.I2 = p.I[2]; a.Cin = p.Cin;
    a.O0 = p.O[0]; a.O1 = p.O[1]; a.O2 = p.O[2]; a.Cout = p.Cout; a.Npad = p.Npad;
    a.K0 = p.K[0]; a.K1 = p.K[1]; a.K2 = p.K[2];
    a.S0 = p.S[0]; a.S1 = p.S[1]; a.S2 = p.S[2];
    a.P0 = p.P[0]; a.P1 = p.P[1]; a.P2 = p.P[2];
    a.os_b = p.os_b; a.os0 = p.os[0]; a.os1 = p.os[1]; a.os2 = p.os[2]; a.out_off = p.out_off;
    a.act = p.act;
    const bool k32 = p.Cin % 32 == 0;
    // Measured on res2 (B=24, 1.855 TFLOP per launch): register-staged tiles 129 TFLOP/s, with the next tile's
    // ds_writes issued under the last MFMA group 133, LDS-DMA staging (conv_igemm_glds_kernel) 143 -- the default
    // where it applies (128-wide N, 32-channel K slices).  Ablations of the register-staged form on the same shape:
    // no global loads 142.5, no LDS stores 137.0, neither 152.8 TFLOP/s.
    if (p.Npad % 128 == 0) {
        if (k32 && a.w_bytes) {
            // small batches: half-height tiles once the 128-row tiling cannot fill 256 CUs x 2 workgroups twice over
            const long long grid128 = (long long)((a.M + 127) / 128) * (p.Npad / 128);
            return grid128 < 1024 ? launch_glds<64>(a, st) : launch_glds<128>(a, st);
        }
        return k32 ? launch_cfg<128, 128, 32, 2, 2>(a, st) : launch_cfg<128, 128, 16, 2, 2>(a, st);
    } else if (p.Npad % 64 == 0) {
        return k32 ? launch_cfg<128, 64, 32, 2, 2>(a, st) : launch_cfg<128, 64, 16, 2, 2>(a, st);
    }
    return k32 ? launch_cfg<128, 32, 32, 4, 1>(a, st) : launch_cfg<128, 32, 16, 4, 1>(a, st);
}
