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
// GEMM: 512 threads = 8 waves (4 along tiles x 2 along channels), a wave = 2 x 4 MFMA tiles of 32 x 32 (128 accumulators),
// U is the A operand (accumulator registers = 4 consecutive output channels: 16-byte stores), three LDS stages of 48 KiB,
// DMA two K steps ahead with counted vmcnt (the loads stay in flight across the one barrier of a step), persistent grid
// with the XCD-contiguous item order of the fp32 kernel.
#include "rn_common.h"
#include "wino_mats.h"
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int SB_ROW = 96;                                   // bytes per row and K step
constexpr int SB_BM = 256, SB_BN = 256;
constexpr int SB_VB = SB_BM * SB_ROW, SB_UB = SB_BN * SB_ROW, SB_STAGE = SB_VB + SB_UB;      // 24 + 24 KiB
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

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// 1. input transform V = B^T d B, written as three bf16 planes in the GEMM's row layout.  The arithmetic is that of
// wino_input_kernel (conv_wino43.hip) per element -- V itself is bit-identical, only its representation changes.
// thread = (tile, VW channels); a wave = 64 / (16 / VW) tiles x one 16-channel K-step group, so that the three stores of
// a (xi, plane) triple cover the wave's rows completely (96 contiguous bytes per tile, neighbouring tiles adjacent);
// a workgroup = those tiles x 64 channels.
template <class S, int VW>
__global__ __launch_bounds__(256)
void wino_input_bf3_kernel(const float* __restrict__ x, char* __restrict__ Vs, int H, int W, int C, int th, int tw,
                           long long T, unsigned ncb, unsigned nwg, unsigned nblk8, int pad_lo)
{
    typedef float vec __attribute__((ext_vector_type(VW)));
    constexpr int A = S::TA;
    constexpr int LPS = 16 / VW, TPW = 64 / LPS;               // lanes per K-step group; tiles per wave
    const unsigned blk = xcd_contiguous(blockIdx.x, nblk8);
    if (blk >= nwg) return;
    const unsigned cb = blk % ncb;
    const long long tg = blk / ncb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (int)cb * 64 + wave * 16 + (lane % LPS) * VW;
    const long long t = tg * TPW + lane / LPS;
    if (t >= T || c >= C) return;
    const int tx = (int)(t % tw), ty = (int)((t / tw) % th);
    const long long b = t / ((long long)tw * th);
    const int y0 = S::M * ty - pad_lo, x0 = S::M * tx - pad_lo;
    const float* xb = x + ((size_t)b * H * W) * C + c;
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
                const float cf = S::BT(i, k);
                if (cf != 0.f) acc += cf * d[k];
            }
            tt[i][col] = acc;
        }
    }
    // row (xi, s = c / 16, t): plane p at +32 p, chunk (c % 16) / 8 at +16 (chunk ^ bit 3 of t), channel at +2 (c % 8)
    const int s = c >> 4, within = c & 15;
    const unsigned pos = (unsigned)(within >> 3) ^ (unsigned)((t >> 3) & 1);
    const size_t xi_stride = (size_t)T * C * 6;                // bytes per xi: (C / 16) K steps x T rows x 96
    char* vb = Vs + ((size_t)s * T + t) * SB_ROW + pos * 16 + (within & 7) * 2;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j < A; ++j) {
            vec acc = vec(0.f);
#pragma unroll
            for (int k = 0; k < A; ++k) {
                const float cf = S::BT(j, k);
                if (cf != 0.f) acc += cf * tt[i][k];
            }
            float a[VW];
#pragma unroll
            for (int e = 0; e < VW; ++e) a[e] = acc[e];
            unsigned short p[3][VW];
            split3<VW>(a, p);
            char* dst = vb + (size_t)(i * A + j) * xi_stride;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if constexpr (VW == 2) {
                    *reinterpret_cast<unsigned*>(dst + q * 32) = (unsigned)p[q][0] | ((unsigned)p[q][1] << 16);
                } else {
                    uint2 o;
                    o.x = (unsigned)p[q][0] | ((unsigned)p[q][1] << 16);
                    o.y = (unsigned)p[q][2] | ((unsigned)p[q][3] << 16);
                    *reinterpret_cast<uint2*>(dst + q * 32) = o;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// filter transform U = G g G^T (double, as wino_pack_kernel), rounded to fp32 and split: Us [nxi][Cout/256][Cin/16][256][3][16]
template <class S>
__global__ __launch_bounds__(256)
void wino_pack_bf3_kernel(const float* __restrict__ w_tf, char* __restrict__ us, int Cin, int Cout, int transposed)
{
    constexpr int A = S::TA, R = S::R;
    const int nkg = Cin / 4, nblocks = Cout / 256, ksteps = Cin / 16;
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
                    for (int p_ = 0; p_ < R; ++p_) acc += S::G(i, p_) * (double)g[p_][q][r];
                    gg[i][q][r] = acc;
                }
        const int s = kg >> 2, within = (kg & 3) * 4;
        const unsigned pos = (unsigned)(within >> 3) ^ (unsigned)((slot >> 3) & 1);
        char* ub = us + (((size_t)nb * ksteps + s) * 256 + slot) * SB_ROW + pos * 16 + (within & 7) * 2;
        const size_t plane = (size_t)nblocks * ksteps * 256 * SB_ROW;      // bytes per xi
#pragma unroll
        for (int i = 0; i < A; ++i)
#pragma unroll
            for (int j = 0; j < A; ++j) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < R; ++q) acc += gg[i][q][r] * S::G(j, q);
                    o[r] = (float)acc;
                }
                unsigned short p[3][4];
                split3<4>(o, p);
                char* dst = ub + (size_t)(i * A + j) * plane;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    uint2 v;
                    v.x = (unsigned)p[q][0] | ((unsigned)p[q][1] << 16);
                    v.y = (unsigned)p[q][2] | ((unsigned)p[q][3] << 16);
                    *reinterpret_cast<uint2*>(dst + q * 32) = v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2. the GEMMs  M[xi] = V[xi] (T x Cin) . U[xi] (Cin x Cout) on split operands
struct Bf3GemmArgs {
    const char* V; const char* U; float* M;
    long long T;
    int Cin, Cout;
    int mblocks, nblocks, ksteps;   // 256-row blocks of T (the last may be ragged), 256-channel blocks, K steps of 16
    int nitems;                     // nxi * mblocks * nblocks
    unsigned v_step_bytes;          // T * 96: one (xi, K step) sub-plane of Vs
    unsigned m_bytes;               // one xi plane of M
    int probe;                      // RN_WINO_BF3_PROBE (timing experiments; results are wrong when set): 1 no DMA in the loop, 2 no stores
};

#define BF3_WAIT_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")

// TAG only names the kernel per layer class in profiler tables (0: F43, 1: F44, 2: F63 Cin >= 1024, 3: F63 narrower)
template <int TAG>
__global__ __launch_bounds__(512, 2)
void wino_gemm_bf3_kernel(const Bf3GemmArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [stage][V 256 x 96 | U 256 x 96]
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hb = lane >> 5;                       // 32x32x16 MFMA: lane = (row / column, k group of 8)
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned sw = (unsigned)((l32 >> 3) & 1);
    const unsigned vfrag = (unsigned)((wm * 64 + l32) * SB_ROW) + (((unsigned)hb ^ sw) << 4);
    const unsigned ufrag = (unsigned)(SB_VB + (wn * 128 + l32) * SB_ROW) + (((unsigned)hb ^ sw) << 4);
    const unsigned dma_lane = (unsigned)(wave * 1024 + lane * 16);

    struct Item { const char* vplane; const char* upanel; float* mplane; long long m0; int nb; };
    const int perm = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);   // XCD-contiguous slot in a round
    auto decode = [&](int r, Item& it) -> bool {
        const int L = r * (int)gridDim.x + perm;                      // gridDim.x is a multiple of 8
        if (L >= a.nitems) return false;
        const int nb = L % a.nblocks;
        const int mbx = L / a.nblocks;
        const int mb = mbx % a.mblocks, xi = mbx / a.mblocks;
        it.nb = nb;
        it.m0 = (long long)mb * SB_BM;
        it.vplane = a.V + (size_t)xi * a.ksteps * a.v_step_bytes;
        it.upanel = a.U + ((size_t)xi * a.nblocks + nb) * ((size_t)a.ksteps * SB_UB);
        it.mplane = a.M + (size_t)xi * a.T * a.Cout;
        return true;
    };
    // one K step of one item -> LDS stage `buf`: 3 + 3 wave instructions of 1 KiB.  Rows >= T of a V sub-plane lie beyond
    // its buffer window: zeros (their outputs are never stored).
    auto issue = [&](const Item& it, int s, int buf) {
        const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(it.vplane + (size_t)s * a.v_step_bytes), 0, a.v_step_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(it.upanel + (size_t)s * SB_UB), 0, SB_UB, 0x00020000);
        char* sb = smem + buf * SB_STAGE;
        const unsigned vo = (unsigned)(it.m0 * SB_ROW) + dma_lane;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (lds_void*)(sb + (wave + 8 * i) * 1024), 16, vo + i * 8192, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lds_void*)(sb + SB_VB + (wave + 8 * i) * 1024), 16, dma_lane + i * 8192, 0, 0, 0);
    };

    f32x16 acc[2][4];
    auto ldv = [&](const char* sb, int mt, bf16x8 (&v)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) v[p] = *reinterpret_cast<const bf16x8*>(sb + vfrag + mt * (32 * SB_ROW) + p * 32);
    };
    auto ldu = [&](const char* sb, int nt, bf16x8 (&u)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) u[p] = *reinterpret_cast<const bf16x8*>(sb + ufrag + nt * (32 * SB_ROW) + p * 32);
    };
    // one 32 x 32 tile and K step: the six piece products with i + j <= 2, smallest terms first
    auto grp = [&](const bf16x8 (&v)[3], const bf16x8 (&u)[3], f32x16& c) {
        constexpr int PU[6] = {2, 1, 0, 1, 0, 0}, PV[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int k = 0; k < 6; ++k) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[PU[k]], v[PV[k]], c, 0, 0, 0);
    };

    Item cur, nxt;
    if (!decode(0, cur)) return;
    bool have_next = decode(1, nxt);
    bf16x8 x0[3], x1[3], ua[3], ub[3];
    issue(cur, 0, 0);
    issue(cur, 1, 1);
    BF3_WAIT_BARRIER(6);
    ldv(smem, 0, x0);
    ldu(smem, 0, ua);
    int buf = 0;                                                      // stage of the current K step
    bool after_store = false;                                         // the previous item's 32 stores sit in the queue behind DMA(g+1)

    // One K step = 8 (row tile, channel tile) groups of 6 MFMAs in an order in which consecutive groups share an operand, so
    // that 4 fragment sets (2 V, 2 U: 48 registers) suffice: a group's new operand is read from LDS two groups ahead of its
    // use.  At entry v0 holds the fragments of row tile 0 and ua those of channel tile 0 (read under the previous step's last
    // group); v1 is free.  Before the last group the next stage is waited for (counted: the stage after it stays in flight),
    // the barrier says every wave has finished reading this stage, and the next step's first operands are read into the
    // two sets that have just become free -- the next step runs with the roles of v0 / v1 swapped.
    auto step = [&](int s, bf16x8 (&v0)[3], bf16x8 (&v1)[3]) {
        const char* sb = smem + buf * SB_STAGE;
        const int bn = buf == SB_NSTAGE - 1 ? 0 : buf + 1;
        const int b2 = bn == SB_NSTAGE - 1 ? 0 : bn + 1;
        const int s2 = s + 2;
        bool issued = false;
        if (!(a.probe & 1)) {
            if (s2 < a.ksteps) { issue(cur, s2, b2); issued = true; }
            else if (have_next) { issue(nxt, s2 - a.ksteps, b2); issued = true; }
        }
        ldv(sb, 1, v1);
        ldu(sb, 1, ub);
        grp(v0, ua, acc[0][0]);
        grp(v1, ua, acc[1][0]);
        __builtin_amdgcn_sched_barrier(0);
        ldu(sb, 2, ua);
        grp(v1, ub, acc[1][1]);
        grp(v0, ub, acc[0][1]);
        __builtin_amdgcn_sched_barrier(0);
        ldu(sb, 3, ub);
        grp(v0, ua, acc[0][2]);
        grp(v1, ua, acc[1][2]);
        grp(v1, ub, acc[1][3]);
        if (!issued) BF3_WAIT_BARRIER(0);
        else if (after_store) BF3_WAIT_BARRIER(38);
        else BF3_WAIT_BARRIER(6);
        if (s + 1 < a.ksteps || have_next) {
            const char* sn = smem + bn * SB_STAGE;
            ldv(sn, 0, v1);
            ldu(sn, 0, ua);
        }
        grp(v0, ub, acc[0][3]);
        __builtin_amdgcn_sched_barrier(0);
        buf = bn;
        after_store = false;
    };

    for (int r = 0;; ++r) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        for (int s = 0; s < a.ksteps; s += 2) {
            step(s, x0, x1);
            step(s + 1, x1, x0);
        }
        // D (32 x 32) = U-tile (rows: channels) x V-tile (cols: tile rows): register r of lane (l32, hb) is channel
        // (r & 3) + 8*(r >> 2) + 4*hb of the 32-channel tile, tile row l32 -> four 16-byte stores per MFMA tile
        if (!(a.probe & 2)) {
            const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(cur.mplane, 0, a.m_bytes, 0x00020000);
            const unsigned mo = (unsigned)(((cur.m0 + wm * 64 + l32) * a.Cout + cur.nb * SB_BN + wn * 128 + hb * 4) * 4);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 o = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), mrsrc,
                                                               mo + (unsigned)(mt * 32 * a.Cout * 4) + nt * 128 + g * 32, 0, 0);
                    }
            after_store = true;
        }
        if (!have_next) break;
        cur = nxt;
        have_next = decode(r + 2, nxt);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
bool rn_wino_bf3_supported(int scheme, int Cin, int Cout)
{
    static const bool off = getenv("RN_NO_WINOGRAD_BF3") != nullptr;
    return !off && rn_wino43_supported(scheme, Cin, Cout);
}

size_t rn_wino_bf3_packed_bytes(int scheme, int Cin, int Cout) { return (size_t)rn_wino_scheme_nxi(scheme) * Cin * Cout * 6; }

// workspace: Vs (nxi * T * Cin * 6 bytes, rounded up to 256) followed by M (nxi * T * Cout floats)
size_t rn_wino_bf3_v_bytes(int scheme, long long T, int Cin) { return ((size_t)rn_wino_scheme_nxi(scheme) * T * Cin * 6 + 255) / 256 * 256; }

size_t rn_wino_bf3_workspace_bytes(int scheme, int B, int H, int W, int Cin, int Cout)
{
    const int m = rn_wino_scheme_m(scheme);
    if (m == 0) return 0;
    const long long T = (long long)B * ((H + m - 1) / m) * ((W + m - 1) / m);
    return rn_wino_bf3_v_bytes(scheme, T, Cin) + (size_t)rn_wino_scheme_nxi(scheme) * T * Cout * 4;
}

int rn_launch_wino_pack_bf3(int scheme, const float* w_tf, void* us, int Cin, int Cout, int transposed, hipStream_t st)
{
    if (!rn_wino43_supported(scheme, Cin, Cout)) return rn_set_error(RN_E_UNSUPPORTED, "wino_pack_bf3: scheme=%d Cin=%d Cout=%d", scheme, Cin, Cout);
    const size_t tot = (size_t)(Cin / 4) * Cout;
    const unsigned nbw = (unsigned)((tot + 255) / 256 > 65536 ? 65536 : (tot + 255) / 256);
    char* u = static_cast<char*>(us);
    if (scheme == RN_WINO_F43) hipLaunchKernelGGL(wino_pack_bf3_kernel<WinoF43>, dim3(nbw), dim3(256), 0, st, w_tf, u, Cin, Cout, transposed);
    else if (scheme == RN_WINO_F44) hipLaunchKernelGGL(wino_pack_bf3_kernel<WinoF44>, dim3(nbw), dim3(256), 0, st, w_tf, u, Cin, Cout, transposed);
    else hipLaunchKernelGGL(wino_pack_bf3_kernel<WinoF63>, dim3(nbw), dim3(256), 0, st, w_tf, u, Cin, Cout, transposed);
    return rn_check_launch("wino_pack_bf3");
}

int rn_launch_wino_input_bf3(int scheme, const float* x, void* Vs, int B, int H, int W, int C, int pad_lo, hipStream_t st)
{
    const int m = rn_wino_scheme_m(scheme);
    if (m == 0 || C % 16 != 0) return rn_set_error(RN_E_INVALID, "wino_input_bf3: scheme %d, C %d", scheme, C);
    const int th = (H + m - 1) / m, tw = (W + m - 1) / m;
    const long long T = (long long)B * th * tw;
    const int vw = scheme == RN_WINO_F43 ? 4 : 2;
    const int tpw = 64 / (16 / vw);
    const unsigned ncb = (unsigned)((C + 63) / 64);
    const unsigned long long n = (unsigned long long)((T + tpw - 1) / tpw) * ncb;
    if (n > 0x7fffff00ULL) return rn_set_error(RN_E_UNSUPPORTED, "wino_input_bf3: grid too large");
    const unsigned nwg = (unsigned)n, nblk8 = (unsigned)((n + 7) / 8 * 8);
    char* v = static_cast<char*>(Vs);
    if (scheme == RN_WINO_F43)
        hipLaunchKernelGGL((wino_input_bf3_kernel<WinoF43, 4>), dim3(nblk8), dim3(256), 0, st, x, v, H, W, C, th, tw, T, ncb, nwg, nblk8, pad_lo);
    else if (scheme == RN_WINO_F44)
        hipLaunchKernelGGL((wino_input_bf3_kernel<WinoF44, 2>), dim3(nblk8), dim3(256), 0, st, x, v, H, W, C, th, tw, T, ncb, nwg, nblk8, pad_lo);
    else
        hipLaunchKernelGGL((wino_input_bf3_kernel<WinoF63, 2>), dim3(nblk8), dim3(256), 0, st, x, v, H, W, C, th, tw, T, ncb, nwg, nblk8, pad_lo);
    return rn_check_launch("wino_input_bf3");
}

template <int TAG>
static int wino_gemm_bf3_launch_t(const Bf3GemmArgs& a, hipStream_t st)
{
    const size_t lds = (size_t)SB_NSTAGE * SB_STAGE;
    auto kern = wino_gemm_bf3_kernel<TAG>;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc_ != RN_OK) return rc_; }
    const int n = a.nitems;
    hipLaunchKernelGGL(kern, dim3(n < 256 ? (unsigned)((n + 7) / 8 * 8) : 256u), dim3(512), lds, st, a);
    return rn_check_launch("wino_gemm_bf3");
}

int rn_launch_wino_gemm_bf3(int scheme, const void* Vs, const void* us, float* M, long long T, int Cin, int Cout, hipStream_t st)
{
    if (!rn_wino43_supported(scheme, Cin, Cout)) return rn_set_error(RN_E_UNSUPPORTED, "wino_gemm_bf3: scheme=%d Cin=%d Cout=%d", scheme, Cin, Cout);
    if (T < 1 || (T + SB_BM) * (long long)Cout * 4 >= 0xffffff00LL || T * (long long)Cout * 4 >= 0x7fffff00LL || (T + SB_BM) * (long long)SB_ROW >= 0x7fffff00LL)
        return rn_set_error(RN_E_UNSUPPORTED, "wino_gemm_bf3: a transform plane must stay below the 2 GiB buffer window");
    Bf3GemmArgs a;
    a.V = static_cast<const char*>(Vs); a.U = static_cast<const char*>(us); a.M = M; a.T = T; a.Cin = Cin; a.Cout = Cout;
    a.nblocks = Cout / SB_BN; a.ksteps = Cin / 16; a.mblocks = (int)((T + SB_BM - 1) / SB_BM);
    const long long nitems = (long long)rn_wino_scheme_nxi(scheme) * a.mblocks * a.nblocks;
    if (nitems > 0x3fffffff) return rn_set_error(RN_E_UNSUPPORTED, "wino_gemm_bf3: too many items");
    a.nitems = (int)nitems;
    a.v_step_bytes = (unsigned)(T * SB_ROW); a.m_bytes = (unsigned)(T * Cout * 4);
    { static const int probe = getenv("RN_WINO_BF3_PROBE") ? atoi(getenv("RN_WINO_BF3_PROBE")) : 0; a.probe = probe; }
    const int tag = scheme == RN_WINO_F43 ? 0 : scheme == RN_WINO_F44 ? 1 : Cin >= 1024 ? 2 : 3;
    switch (tag) {
    case 0: return wino_gemm_bf3_launch_t<0>(a, st);
    case 1: return wino_gemm_bf3_launch_t<1>(a, st);
    case 2: return wino_gemm_bf3_launch_t<2>(a, st);
    default: return wino_gemm_bf3_launch_t<3>(a, st);
    }
}

// x [B,H,W,Cin] -> y [B,H,W,Cout]; us from rn_launch_wino_pack_bf3; ws >= rn_wino_bf3_workspace_bytes(...) bytes
int rn_launch_conv_wino_bf3(int scheme, const float* x, const void* us, const float* bias, const float* alpha, const float* residual,
                            float* y, float* preact, void* ws, int B, int H, int W, int Cin, int Cout, int pad_lo, int act, hipStream_t st)
{
    if (!rn_wino43_supported(scheme, Cin, Cout)) return rn_set_error(RN_E_UNSUPPORTED, "conv_wino_bf3: scheme=%d Cin=%d Cout=%d", scheme, Cin, Cout);
    const int m = rn_wino_scheme_m(scheme);
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
            const int rc = rn_launch_conv_wino_bf3(scheme, x + xo, us, bias, alpha, residual ? residual + yo : nullptr, y + yo,
                                                   preact ? preact + yo : nullptr, ws, nb, H, W, Cin, Cout, pad_lo, act, st);
            if (rc != RN_OK) return rc;
        }
        return RN_OK;
    }
    char* Vs = static_cast<char*>(ws);
    float* M = reinterpret_cast<float*>(Vs + rn_wino_bf3_v_bytes(scheme, T, Cin));
    int rc = rn_launch_wino_input_bf3(scheme, x, Vs, B, H, W, Cin, pad_lo, st);
    if (rc != RN_OK) return rc;
    rc = rn_launch_wino_gemm_bf3(scheme, Vs, us, M, T, Cin, Cout, st);
    if (rc != RN_OK) return rc;
    return rn_launch_wino_output(scheme, M, bias, alpha, residual, y, preact, B, H, W, Cout, act, st);
}
