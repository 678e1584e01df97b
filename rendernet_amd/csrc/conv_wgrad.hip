// Filter gradient of every convolution flavour of the path on the gfx950 matrix cores, exact fp32
// (v_mfma_f32_32x32x2_f32).  This is the wgrad leg of the training step that TensorFlow's autodiff
// derives for tf.nn.conv3d / conv2d / conv2d_transpose (tools/layer_util.py:171,212,253;
// RenderNet_Shader.py:165-167 `AdamOptimizer(...).minimize(recon_loss)`).
//
// One formulation serves forward convs and transposed convs:
//     dw[t0,t1,t2][ca][cg] += sum_{b,o0,o1,o2} A[b, o0*S0-P0+t0, o1*S1-P1+t1, o2*S2-P2+t2, ca] * G[b,o0,o1,o2,cg]
//   forward conv  y = conv(x, w[k..,Cin,Cout]):            A = x,  G = dz           (dw in TF layout [k..,Cin,Cout])
//   transposed    y = convT(x, w[k..,Cout,Cin]), stride s: A = dz, G = x            (dw in TF layout [k..,Cout,Cin])
// out-of-range A coordinates contribute zero (TF SAME).
//
// GEMM view per tap: out[Ca x Cg] = A_tap^T [Ca x M] * G [M x Cg], M = B*O0*O1*O2 positions.  Both
// operands are "reduction-major" in memory (a position is a contiguous channel run), so tiles go
// global -> LDS with buffer_load_dwordx4 ... lds exactly as they lie ([BK positions][BM|BN channels],
// no transpose, no staging VGPRs) and MFMA fragments are conflict-free ds_read_b32 rows (lanes 0-31 =
// 32 consecutive channels of position k, lanes 32-63 = position k+1).  SAME padding, the channel tail
// and the end of the reduction range come from the buffer bounds check (offset >= 2^31 -> zeros).
// The reduction is split over `nsplit` workgroups per (tap, tile) -- and over the waves of a workgroup
// for narrow tiles -- and partial tiles are accumulated into dw with hardware fp32 atomics, so dw must
// be zero-initialised (or hold the running gradient) on entry.  blockIdx%8 = split%8 keeps the
// workgroups that stream the same positions on one XCD (one L2).
#include "rn_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_MAX_CHUNK = 3072;      // positions per workgroup (row table = 12 KiB of LDS)

struct WgradArgs {
    const float* a; const float* g; float* dw;
    unsigned a_bytes, g_bytes;
    int I0, I1, I2, Ca;
    int O0, O1, O2, Cg;
    int K0, K1, K2, S0, S1, S2, P0, P1, P2;
    int M;                       // positions in the reduction
    int mtiles, ntiles, nsplit, kchunk;
};

template <int BM, int BN, int BK, int WM, int WN, int WK>
__global__ __launch_bounds__(256, 2)
void conv_wgrad_kernel(const WgradArgs a)
{
    static_assert(WM * WN * WK == 4, "4 waves");
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int CA = BM / 4, CG = BN / 4;          // 16-B chunks per tile row
    constexpr int RA = 64 / CA, RG = 64 / CG;        // rows per wave DMA instruction
    constexpr int IAW = BK / RA / 4, IGW = BK / RG / 4;   // DMA instructions per wave per stage
    static_assert(IAW >= 1 && IGW >= 1 && TM >= 1 && TN >= 1, "tile shape");
    constexpr int ASZ = BK * BM, GSZ = BK * BN;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);      // [2][BK][BM]
    float* Gs = As + 2 * ASZ;                        // [2][BK][BN]
    unsigned* rowtab = reinterpret_cast<unsigned*>(Gs + 2 * GSZ);   // [WG_MAX_CHUNK] A-row byte offsets

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);

    int id = blockIdx.x;
    const int split = id % a.nsplit; id /= a.nsplit;
    const int tn = id % a.ntiles; id /= a.ntiles;
    const int tm = id % a.mtiles; const int tap = id / a.mtiles;
    const int t2 = tap % a.K2, t1 = (tap / a.K2) % a.K1, t0 = tap / (a.K2 * a.K1);
    const int ca0 = tm * BM, cg0 = tn * BN;
    const int kbeg = split * a.kchunk;
    const int kend = min(a.M, kbeg + a.kchunk);
    if (kbeg >= kend) return;
    const int nstage = (kend - kbeg + BK - 1) / BK;

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t arsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.a), 0, a.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t grsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, a.g_bytes, 0x00020000);

    // Row table: the byte offset of the A row of every position of this workgroup's reduction chunk for
    // ITS tap (or OOB where SAME padding / the end of the chunk applies), computed once -- an odometer
    // per thread, stepping 256 positions at a time -- and kept in LDS.  The per-stage DMA bookkeeping is
    // then one ds_read_b32 + select + add per instruction instead of a position -> (b,o0,o1,o2)
    // decomposition with bounds checks (measured: that block, vector or scalar, sat in front of every
    // stage's MFMAs -- see DESIGN.md).
    {
        const int nrows = nstage * BK;
        int pos = kbeg + tid;
        int r = pos;
        int o2 = r % a.O2; r /= a.O2;
        int o1 = r % a.O1; r /= a.O1;
        int o0 = r % a.O0; int b = r / a.O0;
        const int e2 = 256 % a.O2; r = 256 / a.O2;
        const int e1 = r % a.O1; r /= a.O1;
        const int e0 = r % a.O0; const int eb = r / a.O0;
        for (int i = tid; i < nrows; i += 256) {
            const int i0 = o0 * a.S0 - a.P0 + t0, i1 = o1 * a.S1 - a.P1 + t1, i2 = o2 * a.S2 - a.P2 + t2;
            const bool ok = pos < kend && (unsigned)i0 < (unsigned)a.I0 && (unsigned)i1 < (unsigned)a.I1 &&
                            (unsigned)i2 < (unsigned)a.I2;
            const unsigned e = (unsigned)(((b * a.I0 + i0) * a.I1 + i1) * a.I2 + i2) * (unsigned)a.Ca;
            rowtab[i] = ok ? e * 4u : OOB;
            pos += 256;
            int c;
            o2 += e2; c = o2 >= a.O2; o2 -= c ? a.O2 : 0;
            o1 += e1 + c; c = o1 >= a.O1; o1 -= c ? a.O1 : 0;
            o0 += e0 + c; c = o0 >= a.O0; o0 -= c ? a.O0 : 0;
            b += eb + c;
        }
    }
    __syncthreads();

    const int achunk = lane % CA, arow = lane / CA;
    const bool a_ch_ok = ca0 + achunk * 4 < a.Ca;
    const unsigned a_cbytes = (unsigned)(ca0 + achunk * 4) * 4u;
    const int gchunk = lane % CG, grow = lane / CG;
    const bool g_ch_ok = cg0 + gchunk * 4 < a.Cg;
    const unsigned g_cbytes = (unsigned)(cg0 + gchunk * 4) * 4u;
    const unsigned g_rbytes = (unsigned)a.Cg * 4u;

    typedef __attribute__((address_space(3))) void lds_void;
    // stage `st` of the chunk -> LDS stage buffer `buf`.  The A-row offsets of a stage are fetched from the
    // row table one stage ahead (noff[]), so a stage opens with its 8 DMA issues back to back: a table read
    // cannot be hoisted above an LDS-DMA by the compiler (both touch LDS), which serialised read -> wait ->
    // DMA four times at the head of every stage.
    static_assert(IAW <= 8, "noff");
    unsigned noff[8];   // fixed extent: with the dependent extent [IAW] clang 22 silently drops the HOST stub of this kernel
    auto load_tab = [&](int st) {
#pragma unroll
        for (int q = 0; q < IAW; ++q) noff[q] = rowtab[st * BK + (wave * IAW + q) * RA + arow];
    };
    auto issue_dma = [&](int st, int buf) {
#pragma unroll
        for (int q = 0; q < IAW; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_void*)(As + buf * ASZ + (wave * IAW + q) * RA * BM), 16,
                                                     a_ch_ok ? noff[q] + a_cbytes : OOB, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < IGW; ++q) {
            const int gp = kbeg + st * BK + (wave * IGW + q) * RG + grow;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(grsrc, (lds_void*)(Gs + buf * GSZ + (wave * IGW + q) * RG * BN), 16,
                                                     (g_ch_ok && gp < kend) ? (unsigned)gp * g_rbytes + g_cbytes : OOB,
                                                     0, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tab(0);
    issue_dma(0, 0);
    if (nstage > 1) load_tab(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (int s = 0; s < nstage; ++s) {
        if (s + 1 < nstage) {
            issue_dma(s + 1, cur ^ 1);
            if (s + 2 < nstage) load_tab(s + 2);
        }
        // lane li owns the TM (TN) ADJACENT channels li*TM.. of the wave tile, so that one ds_read_b64 feeds both
        // 32x32 sub-tiles (sub-tile i = channels {li*TM + i}: an interleaved row set, undone in the epilogue)
        const float* Ab = As + cur * ASZ + lh * BM + wm * WTM + li * TM;
        const float* Gb = Gs + cur * GSZ + lh * BN + wn * WTN + li * TN;
        // Fragment reads run ONE k-step ahead of the MFMAs that consume them (two register sets, fully
        // unrolled): issued behind the MFMAs of the same step, each read's LDS latency (~100+ cycles) would
        // open a bubble in the matrix pipe after every 4 MFMAs (256 cycles) -- measured 122 -> see DESIGN.md.
        constexpr int KS = BK / 2 / WK;
        typedef float fragA __attribute__((ext_vector_type(TM)));
        typedef float fragG __attribute__((ext_vector_type(TN)));
        fragA af[2];
        fragG bf[2];
        af[0] = *reinterpret_cast<const fragA*>(Ab + wk * 2 * BM);
        bf[0] = *reinterpret_cast<const fragG*>(Gb + wk * 2 * BN);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk + 1 < KS) {
                const int ks = (kk + 1) * WK + wk;
                af[(kk + 1) & 1] = *reinterpret_cast<const fragA*>(Ab + ks * 2 * BM);
                bf[(kk + 1) & 1] = *reinterpret_cast<const fragG*>(Gb + ks * 2 * BN);
            }
            __builtin_amdgcn_sched_barrier(0);     // keep the reads above the MFMAs (the scheduler sinks them back)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* dwt = a.dw + (size_t)tap * a.Ca * a.Cg;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int cg = cg0 + wn * WTN + li * TN + j;
        if (cg >= a.Cg) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ca = ca0 + wm * WTM + ((r & 3) + 8 * (r >> 2) + 4 * lh) * TM + i;
                if (ca < a.Ca) unsafeAtomicAdd(dwt + (size_t)ca * a.Cg + cg, acc[i][j][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Narrow-input variant (Ca*taps <= 1024 threads, Cg <= 16): e_conv1 (Cin = 1|5, 125 taps, 8 outputs),
// the wgrad of e_conv11 (A = dz with 1|3 channels).  One thread owns one (tap, ca) pair and CGT
// accumulators; the workgroup walks a chunk of positions, every thread gathering its own A sample
// while the G row is a wave-uniform (broadcast) load.
// ------------------------------------------------------------------------------------------------
template <int CGT>
__global__ __launch_bounds__(1024)
void conv_wgrad_small_kernel(const WgradArgs a)
{
    const int taps = a.K0 * a.K1 * a.K2;
    const int t = threadIdx.x;
    const bool live = t < taps * a.Ca;
    const int ca = live ? t % a.Ca : 0, tap = live ? t / a.Ca : 0;
    const int t2 = tap % a.K2, t1 = (tap / a.K2) % a.K1, t0 = tap / (a.K2 * a.K1);
    const int kbeg = blockIdx.x * a.kchunk;
    const int kend = min(a.M, kbeg + a.kchunk);
    float acc[CGT];
#pragma unroll
    for (int n = 0; n < CGT; ++n) acc[n] = 0.f;
    int pos = kbeg;
    int o2 = pos % a.O2; pos /= a.O2;
    int o1 = pos % a.O1; pos /= a.O1;
    int o0 = pos % a.O0; int b = pos / a.O0;
    for (int p = kbeg; p < kend; ++p) {
        const int i0 = o0 * a.S0 - a.P0 + t0, i1 = o1 * a.S1 - a.P1 + t1, i2 = o2 * a.S2 - a.P2 + t2;
        float av = 0.f;
        if (live && (unsigned)i0 < (unsigned)a.I0 && (unsigned)i1 < (unsigned)a.I1 && (unsigned)i2 < (unsigned)a.I2)
            av = a.a[((((size_t)b * a.I0 + i0) * a.I1 + i1) * a.I2 + i2) * a.Ca + ca];
        const float* gp = a.g + (size_t)p * a.Cg;
#pragma unroll
        for (int n = 0; n < CGT; ++n)
            if (n < a.Cg) acc[n] = fmaf(av, gp[n], acc[n]);
        if (++o2 == a.O2) { o2 = 0; if (++o1 == a.O1) { o1 = 0; if (++o0 == a.O0) { o0 = 0; ++b; } } }
    }
    if (live) {
        float* d = a.dw + ((size_t)tap * a.Ca + ca) * a.Cg;
#pragma unroll
        for (int n = 0; n < CGT; ++n)
            if (n < a.Cg) unsafeAtomicAdd(d + n, acc[n]);
    }
}

// ------------------------------------------------------------------------------------------------
// Narrow variant with LDS staging (taps*Cg <= 1024 threads, Ca <= 8): e_conv1 (A = x with 1 | 5 channels, 125 taps, G = dz
// with 8) and the wgrad of e_conv11 (A = dz with 1 | 3 channels, G = x with 16 | 32).  The kernel above walks its positions
// with one gathered A sample and Cg broadcast G loads from global memory per position and thread -- a chain of memory
// latencies (e_conv11 at crop 64: 2.0 ms for 107 MB of operands).  Here a workgroup stages a T0 x T1 x T2 tile of output
// positions -- its G rows and the A box they touch -- in LDS with coalesced row loads (zeros where SAME padding applies),
// thread (tap, cg) then runs over the tile's positions out of LDS with Ca accumulators, and the workgroup keeps
// accumulating over its tiles (id, id + grid, ...) before it adds its taps*Ca*Cg sums to dw with one atomic each.
// ------------------------------------------------------------------------------------------------
struct WgradTileArgs {
    WgradArgs w;
    int T0, T1, T2;              // output positions per tile
    int R0, R1, R2;              // A box of a tile: (T-1)*S + K
    int nt0, nt1, nt2, ntiles;   // tiles per item and dimension; tiles in all (B * nt0 * nt1 * nt2)
};

template <int CA>
__global__ __launch_bounds__(1024)
void conv_wgrad_tile_kernel(const WgradTileArgs ta)
{
    const WgradArgs& a = ta.w;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int abox = ta.R0 * ta.R1 * ta.R2 * CA;
    float* As = reinterpret_cast<float*>(smem);           // [R0][R1][R2][CA]
    float* Gs = As + ((abox + 3) & ~3);                   // [T0*T1*T2][Cg]
    const int taps = a.K0 * a.K1 * a.K2;
    const int t = threadIdx.x, nth = blockDim.x;
    const bool live = t < taps * a.Cg;
    const int cg = live ? t % a.Cg : 0, tap = live ? t / a.Cg : 0;
    const int t2 = tap % a.K2, t1 = (tap / a.K2) % a.K1, t0 = tap / (a.K2 * a.K1);
    const int tbase = ((t0 * ta.R1 + t1) * ta.R2 + t2) * CA;      // this thread's tap inside the A box
    const int P = ta.T0 * ta.T1 * ta.T2;
    float acc[CA];
#pragma unroll
    for (int c = 0; c < CA; ++c) acc[c] = 0.f;

    for (int tile = blockIdx.x; tile < ta.ntiles; tile += gridDim.x) {
        int r = tile;
        const int b2 = r % ta.nt2; r /= ta.nt2;
        const int b1 = r % ta.nt1; r /= ta.nt1;
        const int b0 = r % ta.nt0; const int b = r / ta.nt0;
        const int o0b = b0 * ta.T0, o1b = b1 * ta.T1, o2b = b2 * ta.T2;
        const int i0b = o0b * a.S0 - a.P0, i1b = o1b * a.S1 - a.P1, i2b = o2b * a.S2 - a.P2;
        __syncthreads();                                  // the previous tile is consumed
        // A box: rows of R2*CA floats (contiguous along the last input axis)
        const int rowa = ta.R2 * CA;
        for (int i = t; i < abox; i += nth) {
            const int e = i % rowa, rr = i / rowa;
            const int r1 = rr % ta.R1, r0 = rr / ta.R1;
            const int i0 = i0b + r0, i1 = i1b + r1, i2 = i2b + e / CA;
            float v = 0.f;
            if ((unsigned)i0 < (unsigned)a.I0 && (unsigned)i1 < (unsigned)a.I1 && (unsigned)i2 < (unsigned)a.I2)
                v = a.a[((((size_t)b * a.I0 + i0) * a.I1 + i1) * a.I2 + i2b) * CA + e];
            As[i] = v;
        }
        // G rows: T2*Cg floats per (q0, q1); positions past the end of an axis read as zero (they contribute nothing)
        const int rowg = ta.T2 * a.Cg;
        for (int i = t; i < P * a.Cg; i += nth) {
            const int e = i % rowg, rr = i / rowg;
            const int q1 = rr % ta.T1, q0 = rr / ta.T1;
            const int o0 = o0b + q0, o1 = o1b + q1, o2 = o2b + e / a.Cg;
            float v = 0.f;
            if (o0 < a.O0 && o1 < a.O1 && o2 < a.O2)
                v = a.g[((((size_t)b * a.O0 + o0) * a.O1 + o1) * a.O2 + o2b) * a.Cg + e];
            Gs[i] = v;
        }
        __syncthreads();
        if (live) {
            const float* gp = Gs + cg;
            for (int q0 = 0; q0 < ta.T0; ++q0)
                for (int q1 = 0; q1 < ta.T1; ++q1) {
                    const float* ap = As + tbase + ((q0 * a.S0 * ta.R1 + q1 * a.S1) * ta.R2) * CA;
                    const float* gq = gp + (size_t)((q0 * ta.T1 + q1) * ta.T2) * a.Cg;
#pragma unroll 4
                    for (int q2 = 0; q2 < ta.T2; ++q2) {
                        const float g = gq[q2 * a.Cg];
#pragma unroll
                        for (int c = 0; c < CA; ++c) acc[c] = fmaf(ap[q2 * a.S2 * CA + c], g, acc[c]);
                    }
                }
        }
    }
    if (live) {
#pragma unroll
        for (int c = 0; c < CA; ++c) unsafeAtomicAdd(a.dw + ((size_t)tap * CA + c) * a.Cg + cg, acc[c]);
    }
}

// picks a tile that fits LDS; returns RN_E_UNSUPPORTED when the shape is not for this kernel
static int launch_wgrad_tile(const WgradArgs& a, int B, hipStream_t st)
{
    static const bool off = getenv("RN_WGRAD_NO_TILE") != nullptr;
    const int taps = a.K0 * a.K1 * a.K2;
    if (off || taps * a.Cg > 1024 || a.Ca > 8 || (a.Ca != 1 && a.Ca != 3 && a.Ca != 5)) return RN_E_UNSUPPORTED;
    WgradTileArgs ta;
    ta.w = a;
    if (a.I2 == 1 && a.O2 == 1 && a.K2 == 1) {          // a 2-D layer [H][W][1][C] is the 3-D layer [1][H][W][C]: same memory, same taps
        WgradArgs& w = ta.w;
        w.I2 = a.I1; w.I1 = a.I0; w.I0 = 1; w.O2 = a.O1; w.O1 = a.O0; w.O0 = 1;
        w.K2 = a.K1; w.K1 = a.K0; w.K0 = 1; w.S2 = a.S1; w.S1 = a.S0; w.S0 = 1; w.P2 = a.P1; w.P1 = a.P0; w.P0 = 0;
    }
    {
        const WgradArgs& w = ta.w;
        // the last axis long (coalesced rows), a few rows of the others
        ta.T2 = w.O2 < 64 ? w.O2 : (w.S2 == 1 ? 64 : 32);
        ta.T1 = w.O1 < 8 ? w.O1 : (w.O0 == 1 ? 8 : 4);
        ta.T0 = w.O0 < 2 ? w.O0 : 2;
    }
    const WgradArgs& a_ = ta.w;
    for (;;) {
        ta.R0 = (ta.T0 - 1) * a_.S0 + a_.K0; ta.R1 = (ta.T1 - 1) * a_.S1 + a_.K1; ta.R2 = (ta.T2 - 1) * a_.S2 + a_.K2;
        const size_t lds = ((size_t)((ta.R0 * ta.R1 * ta.R2 * a.Ca + 3) & ~3) + (size_t)ta.T0 * ta.T1 * ta.T2 * a.Cg) * 4;
        if (lds <= 64 * 1024) break;
        if (ta.T0 > 1) ta.T0 = (ta.T0 + 1) / 2; else if (ta.T1 > 1) ta.T1 = (ta.T1 + 1) / 2; else if (ta.T2 > 8) ta.T2 /= 2; else return RN_E_UNSUPPORTED;
    }
    ta.nt0 = (a_.O0 + ta.T0 - 1) / ta.T0; ta.nt1 = (a_.O1 + ta.T1 - 1) / ta.T1; ta.nt2 = (a_.O2 + ta.T2 - 1) / ta.T2;
    const long long nt = (long long)B * ta.nt0 * ta.nt1 * ta.nt2;
    if (nt > 0x7fffffffLL) return RN_E_UNSUPPORTED;
    ta.ntiles = (int)nt;
    const size_t lds = ((size_t)((ta.R0 * ta.R1 * ta.R2 * a.Ca + 3) & ~3) + (size_t)ta.T0 * ta.T1 * ta.T2 * a.Cg) * 4;
    const int nth = (taps * a.Cg + 63) / 64 * 64;
    const int per_cu = nth > 512 ? 1 : 2;
    const unsigned grid = (unsigned)(nt < 256 * per_cu ? nt : 256 * per_cu);
    auto go = [&](auto kern) -> int {
        const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
        if (rc_ != RN_OK) return rc_;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(nth), lds, st, ta);
        return rn_check_launch("conv_wgrad_tile");
    };
    return a.Ca == 1 ? go(conv_wgrad_tile_kernel<1>) : a.Ca == 3 ? go(conv_wgrad_tile_kernel<3>) : go(conv_wgrad_tile_kernel<5>);
}

template <int BM, int BN, int BK, int WM, int WN, int WK>
static int launch_wgrad(WgradArgs& a, hipStream_t st)
{
    const int taps = a.K0 * a.K1 * a.K2;
    a.mtiles = (a.Ca + BM - 1) / BM;
    a.ntiles = (a.Cg + BN - 1) / BN;
    const long long tiles = (long long)taps * a.mtiles * a.ntiles;
    // enough workgroups to fill 256 CUs x 2 several times over, in multiples of 8 (one split residue
    // class per XCD), but never fewer than 4 stages of reduction per workgroup
    static const int target_wgs = getenv("RN_WGRAD_WGS") ? atoi(getenv("RN_WGRAD_WGS")) : 3072;
    long long ns = (target_wgs + tiles - 1) / tiles;
    ns = (ns + 7) / 8 * 8;
    const long long max_ns = (a.M + 4LL * BK - 1) / (4LL * BK);
    if (ns > max_ns) ns = max_ns;
    if (ns < 1) ns = 1;
    long long kc = (a.M + ns - 1) / ns;
    if (kc > WG_MAX_CHUNK) kc = WG_MAX_CHUNK;
    kc = (kc + BK - 1) / BK * BK;
    ns = (a.M + kc - 1) / kc;
    a.nsplit = (int)ns; a.kchunk = (int)kc;
    const long long nb = tiles * ns;
    if (nb <= 0 || nb > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv_wgrad: bad grid %lld", nb);
    const size_t lds = (size_t)2 * BK * (BM + BN) * 4 + (size_t)WG_MAX_CHUNK * 4;
    auto kern = conv_wgrad_kernel<BM, BN, BK, WM, WN, WK>;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(lds)); if (rc_ != RN_OK) return rc_; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(256), lds, st, a);
    return rn_check_launch("conv_wgrad");
}

// ------------------------------------------------------------------------------------------------
// 3x3x3, stride 1, 32 -> 32 channels (the 21 convs of the 3-D encoder's residual stack,
// RenderNet_Shader.py:48-64): depth-run variant.  The generic kernel gives one workgroup ONE tap, so the
// G rows (and the A rows, shifted) are re-read 27 times through L2 -- 8 flop per byte, L2-bound at
// 55-73 TFLOP/s.  Here a workgroup owns one (t0,t1) pair and accumulates its THREE depth taps from one
// staged slab: per stage 4 (b,h,w) columns x 32 depth positions of G (16 KiB) and the same columns of A
// shifted by (t0-1, t1-1) with one halo row on either side in depth (4 x 34 rows, 17 KiB; rows outside
// the volume come back as zeros from the buffer bounds check) -- tap t2 of position d is slab row d+t2.
// 24 flop per staged byte.  Wave w reduces column w: 16 k-steps x 3 taps per stage, three 32x32
// accumulators; the four waves' partials are summed through LDS and added to dw with fp32 atomics
// (3072 per workgroup).  blockIdx % 8 = split % 8: the nine pairs that stream the same slabs share an XCD.
// ------------------------------------------------------------------------------------------------
struct Wgrad3Args {
    const float* a; const float* g; float* dw;
    unsigned a_bytes, g_bytes;
    int H, W, D, ncols, ndch, nitems, ipw, nsplit;
};

// SPLIT (round 5; rn_conv3d_wgrad_split): the same reduction on the bf16 matrix pipe at fp32 accuracy -- every fp32 value of the two slabs
// as three bf16 pieces (exact sum), the six piece products with i + j <= 2, fp32 accumulation (conv_wino_bf3.hip has the arithmetic).  The
// pieces are made ON THE FLY from the fp32 slabs in LDS (the operands stay 4 bytes per element on their nine trips through L2): per 16
// depth positions a lane reads the 10 slab rows its three depth taps touch and 8 rows of G for its channel, splits them pairwise (positions
// e, e + 1 land in one register = one k pair of a v_mfma_f32_32x32x16_bf16 operand; the odd tap's operand is the even ones' registers
// shifted by 16 bits), and issues 3 taps x 6 products -- about 95 vector instructions per 18 MFMAs of 32 cycles, against 24 exact-fp32
// MFMAs of 64 cycles for the same positions.
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned wg_u32x4 __attribute__((ext_vector_type(4)));

// (a, b) -> the three packed bf16 pairs of their pieces (round to nearest even; the remainders are exact in fp32)
__device__ __forceinline__ void wg_split_pair(float a, float b, unsigned (&p)[3])
{
    wg_f32x2 v = {a, b};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const wg_bf16x2 h = __builtin_convertvector(v, wg_bf16x2);
        p[q] = __builtin_bit_cast(unsigned, h);
        if (q < 2) v -= __builtin_convertvector(h, wg_f32x2);
    }
}

template <bool SPLIT>
__global__ __launch_bounds__(256, 2)
void conv_wgrad_k3d32_kernel(const Wgrad3Args a)
{
    constexpr int C = 32, COLS = 4, TD = 32, ROWS = TD + 2, NROWS = COLS * ROWS;     // 136 slab rows
    constexpr int ASZ = NROWS * C;                      // 4352 floats: 17 DMA instructions of 8 rows
    constexpr int GSZ = COLS * TD * C;                  // 4096 floats: 16 DMA instructions
    constexpr int NAI = NROWS / 8;
    constexpr unsigned OOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void lds_void;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);         // [2][136][32]
    float* Gs = As + 2 * ASZ;                           // [2][128][32]
    __shared__ unsigned colA[2][COLS], colG[2][COLS];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    int id = blockIdx.x;
    const int xs = id & 7; id >>= 3;
    const int pair = id % 9, split = (id / 9) * 8 + xs;
    if (split >= a.nsplit) return;
    const int t0 = pair / 3, t1 = pair % 3;
    const int ibeg = split * a.ipw, iend = min(a.nitems, ibeg + a.ipw);
    if (ibeg >= iend) return;

    const __amdgpu_buffer_rsrc_t arsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.a), 0, a.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t grsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, a.g_bytes, 0x00020000);

    // DMA assignment.  A: wave w issues instructions q = w, w+4, ... (< 17); lane -> slab row 8q + lane/8,
    // 16-B chunk lane%8.  G: wave w issues q = 4w .. 4w+3; lane -> row 8q + lane/8.
    int acol[5], arow[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const int q = wave + 4 * p;
        const int rr = q * 8 + (lane >> 3);
        acol[p] = (q < NAI) ? rr / ROWS : 0;
        arow[p] = (q < NAI) ? rr % ROWS : -0x40000000;
    }
    const unsigned cbytes = (unsigned)((lane & 7) * 16);
    int gcol[4], grow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rr = (wave * 4 + j) * 8 + (lane >> 3);
        gcol[j] = rr / TD; grow[j] = rr % TD;
    }

    auto setup_cols = [&](int buf, int item) {
        if (tid < COLS) {
            const int col = (item / a.ndch) * COLS + tid;
            unsigned ao = OOB, go = OOB;
            if (item < iend && col < a.ncols) {
                const int w = col % a.W, h = (col / a.W) % a.H, b = col / (a.W * a.H);
                const int hh = h + t0 - 1, ww = w + t1 - 1;
                if ((unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W)
                    ao = (unsigned)((b * a.H + hh) * a.W + ww) * (unsigned)(a.D * C * 4);
                go = (unsigned)col * (unsigned)(a.D * C * 4);
            }
            colA[buf][tid] = ao; colG[buf][tid] = go;
        }
    };
    auto issue_dma = [&](int buf, int item, int stage) {
        const int d0 = (item % a.ndch) * TD;
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            if (p < 4 || wave == 0) {
                const unsigned cb = colA[buf][acol[p]];
                const int dd = d0 - 1 + arow[p];
                const unsigned off = ((cb & OOB) || (unsigned)dd >= (unsigned)a.D) ? OOB : cb + (unsigned)dd * (C * 4) + cbytes;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_void*)(As + stage * ASZ + (wave + 4 * p) * 256), 16, off, 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned cb = colG[buf][gcol[j]];
            const int dd = d0 + grow[j];
            const unsigned off = ((cb & OOB) || dd >= a.D) ? OOB : cb + (unsigned)dd * (C * 4) + cbytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(grsrc, (lds_void*)(Gs + stage * GSZ + (wave * 4 + j) * 256), 16, off, 0, 0, 0);
        }
    };

    f32x16 acc0, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }

    setup_cols(0, ibeg);
    setup_cols(1, ibeg + 1);
    __syncthreads();
    issue_dma(0, ibeg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (int item = ibeg; item < iend; ++item) {
        const int n = item - ibeg;
        if (item + 1 < iend) issue_dma((n + 1) & 1, item + 1, cur ^ 1);
        // the column table of item + 2 goes where item's was: item's DMA was issued one iteration ago, and the
        // table of item + 1 (read just above) sits in the other slot
        if constexpr (SPLIT) {
            // lane (li, lh): k group lh of a 16-position block = positions 16 kb + 8 lh + e, e = 0..7; slab row of tap t2 = position + t2
            const float* Ab = As + cur * ASZ + (wave * ROWS + 8 * lh) * C + li;
            const float* Gb = Gs + cur * GSZ + (wave * TD + 8 * lh) * C + li;
#pragma unroll
            for (int kb = 0; kb < TD / 16; ++kb) {
                float xv[10], gv[8];
#pragma unroll
                for (int e = 0; e < 10; ++e) xv[e] = Ab[(16 * kb + e) * C];
#pragma unroll
                for (int e = 0; e < 8; ++e) gv[e] = Gb[(16 * kb + e) * C];
                unsigned xp[5][3], gp[4][3];                      // [position pair][piece]
#pragma unroll
                for (int q = 0; q < 5; ++q) wg_split_pair(xv[2 * q], xv[2 * q + 1], xp[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) wg_split_pair(gv[2 * q], gv[2 * q + 1], gp[q]);
                wg_bf16x8 xa[3][3], gb[3];                        // [tap][piece], [piece]
#pragma unroll
                for (int p_ = 0; p_ < 3; ++p_) {
                    const wg_u32x4 t0v = {xp[0][p_], xp[1][p_], xp[2][p_], xp[3][p_]};
                    const wg_u32x4 t2v = {xp[1][p_], xp[2][p_], xp[3][p_], xp[4][p_]};
                    wg_u32x4 t1v;                                 // positions 1..8: the pairs (1,2) (3,4) (5,6) (7,8)
#pragma unroll
                    for (int q = 0; q < 4; ++q) t1v[q] = (xp[q][p_] >> 16) | (xp[q + 1][p_] << 16);
                    xa[0][p_] = __builtin_bit_cast(wg_bf16x8, t0v);
                    xa[1][p_] = __builtin_bit_cast(wg_bf16x8, t1v);
                    xa[2][p_] = __builtin_bit_cast(wg_bf16x8, t2v);
                    const wg_u32x4 gq = {gp[0][p_], gp[1][p_], gp[2][p_], gp[3][p_]};
                    gb[p_] = __builtin_bit_cast(wg_bf16x8, gq);
                }
                // the six piece products with i + j <= 2, smallest terms first; the three taps interleaved (MFMAs on one accumulator three apart)
                constexpr int PX[6] = {2, 1, 0, 1, 0, 0}, PG[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0][PX[k]], gb[PG[k]], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[1][PX[k]], gb[PG[k]], acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[2][PX[k]], gb[PG[k]], acc2, 0, 0, 0);
                }
            }
        } else {
        const float* Ab = As + cur * ASZ + (wave * ROWS + lh) * C + li;     // slab row (2kk + lh) + t2
        const float* Gb = Gs + cur * GSZ + (wave * TD + lh) * C + li;
        float g[2], x0[2], x1[2], x2[2];
        g[0] = Gb[0]; x0[0] = Ab[0]; x1[0] = Ab[C]; x2[0] = Ab[2 * C];
#pragma unroll
        for (int kk = 0; kk < TD / 2; ++kk) {
            if (kk + 1 < TD / 2) {
                const int o = (kk + 1) * 2 * C;
                g[(kk + 1) & 1] = Gb[o]; x0[(kk + 1) & 1] = Ab[o]; x1[(kk + 1) & 1] = Ab[o + C]; x2[(kk + 1) & 1] = Ab[o + 2 * C];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[kk & 1], g[kk & 1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[kk & 1], g[kk & 1], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x2[kk & 1], g[kk & 1], acc2, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        setup_cols(n & 1, item + 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // sum the four waves' partial tiles through LDS (the stage buffers are dead), then one atomic per output
    float* red = As;                                    // [4 waves][3 taps][32 ca][32 cg] = 48 KiB
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ca = (r & 3) + 8 * (r >> 2) + 4 * lh;
        red[((wave * 3 + 0) * C + ca) * C + li] = acc0[r];
        red[((wave * 3 + 1) * C + ca) * C + li] = acc1[r];
        red[((wave * 3 + 2) * C + ca) * C + li] = acc2[r];
    }
    __syncthreads();
    float* dwp = a.dw + (size_t)pair * 3 * C * C;       // taps (t0,t1,0..2) are adjacent in [3,3,3,Ca,Cg]
    for (int e = tid; e < 3 * C * C; e += 256) {
        const float v = (red[e] + red[3 * C * C + e]) + (red[2 * 3 * C * C + e] + red[3 * 3 * C * C + e]);
        unsafeAtomicAdd(dwp + e, v);
    }
}

// ... and with the three taps of one filter ROW (t0 fixed, t1 = 0..2) per workgroup (W % 4 == 0): the stage's four columns (b, h, w0 .. w0 + 3) of G
// serve all nine (t1, t2) taps of the row, and the A slab needs only the SIX columns w0 - 1 .. w0 + 4 of input row h + t0 - 1 instead of 3 x 4:
// 21.5 KiB staged per 54 MFMAs per wave instead of 33 KiB per 36 -- the nine-pair form above moves 1.8 GB through L2 for the crop-64 layer
// (6 TB/s at 0.29 ms: THAT bounds it), this one 0.8 GB.  Stage = 16 depth positions (one k block); wave w = column w0 + w; nine 32 x 32
// accumulators (144 registers); the four waves' partials are summed through LDS three taps at a time.
__global__ __launch_bounds__(256, 2)
void conv_wgrad_k3d32_row_kernel(const Wgrad3Args a)
{
    constexpr int C = 32, COLS = 4, ACOLS = COLS + 2, TD = 16, ROWS = TD + 2, NAROWS = ACOLS * ROWS;     // 108 slab rows of A
    constexpr int ASZ = 112 * C;                        // 14 DMA instructions of 8 rows (the last four rows are never read)
    constexpr int GSZ = COLS * TD * C;                  // 2048 floats: 8 DMA instructions
    constexpr int NAI = 14, NGI = 8;
    constexpr unsigned OOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void lds_void;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);         // [2][112][32]
    float* Gs = As + 2 * ASZ;                           // [2][64][32]
    __shared__ unsigned colA[2][ACOLS], colG[2][COLS];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    int id = blockIdx.x;
    const int xs = id & 7; id >>= 3;
    const int t0 = id % 3, split = (id / 3) * 8 + xs;
    if (split >= a.nsplit) return;
    const int ibeg = split * a.ipw, iend = min(a.nitems, ibeg + a.ipw);
    if (ibeg >= iend) return;

    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.a), 0, a.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, a.g_bytes, 0x00020000);

    // DMA assignment.  A: wave w issues instructions q = w, w + 4, ... (< 14); lane -> slab row 8 q + lane / 8 of the flat [6 columns][18 rows]
    // list, 16-byte chunk lane % 8.  G: wave w issues q = 2 w, 2 w + 1; lane -> row 8 q + lane / 8 of [4 columns][16 rows].
    int acol[4], arow[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int q = wave + 4 * p;
        const int rr = q * 8 + (lane >> 3);
        const bool ok = q < NAI && rr < NAROWS;
        acol[p] = ok ? rr / ROWS : 0;
        arow[p] = ok ? rr % ROWS : -0x40000000;
    }
    const unsigned cbytes = (unsigned)((lane & 7) * 16);
    int gcol[2], grow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rr = (wave * 2 + j) * 8 + (lane >> 3);
        gcol[j] = rr / TD; grow[j] = rr % TD;
    }

    // item = (group of four columns, depth chunk of 16): columns (b, h, w0 .. w0 + 3) -- W % 4 == 0, so a group never straddles two rows
    auto setup_cols = [&](int buf, int item) {
        if (tid < ACOLS) {
            const int col0 = (item / a.ndch) * COLS;
            unsigned ao = OOB, go = OOB;
            if (item < iend && col0 < a.ncols) {
                const int w0 = col0 % a.W, h = (col0 / a.W) % a.H, b = col0 / (a.W * a.H);
                const int hh = h + t0 - 1, ww = w0 - 1 + tid;
                if ((unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W)
                    ao = (unsigned)((b * a.H + hh) * a.W + ww) * (unsigned)(a.D * C * 4);
                if (tid < COLS) go = (unsigned)(col0 + tid) * (unsigned)(a.D * C * 4);
            }
            colA[buf][tid] = ao;
            if (tid < COLS) colG[buf][tid] = go;
        }
    };
    auto issue_dma = [&](int buf, int item, int stage) {
        const int d0 = (item % a.ndch) * TD;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (wave + 4 * p < NAI) {
                const unsigned cb = colA[buf][acol[p]];
                const int dd = d0 - 1 + arow[p];
                const unsigned off = ((cb & OOB) || (unsigned)dd >= (unsigned)a.D) ? OOB : cb + (unsigned)dd * (C * 4) + cbytes;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_void*)(As + stage * ASZ + (wave + 4 * p) * 256), 16, off, 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned cb = colG[buf][gcol[j]];
            const int dd = d0 + grow[j];
            const unsigned off = ((cb & OOB) || dd >= a.D) ? OOB : cb + (unsigned)dd * (C * 4) + cbytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(grsrc, (lds_void*)(Gs + stage * GSZ + (wave * 2 + j) * 256), 16, off, 0, 0, 0);
        }
    };
    static_assert(NGI == 8, "two G instructions per wave");

    f32x16 acc[3][3];                                   // [t1][t2]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    setup_cols(0, ibeg);
    setup_cols(1, ibeg + 1);
    __syncthreads();
    issue_dma(0, ibeg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (int item = ibeg; item < iend; ++item) {
        const int n = item - ibeg;
        if (item + 1 < iend) issue_dma((n + 1) & 1, item + 1, cur ^ 1);
        // lane (li, lh): k group lh of the 16-position block = depth positions 8 lh + e, e = 0..7; slab row of tap t2 = position + t2;
        // tap t1 of column w0 + wave reads A column wave + t1 of the six staged ones
        const float* Gb = Gs + cur * GSZ + (wave * TD + 8 * lh) * C + li;
        float gv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] = Gb[e * C];
        unsigned gp[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q) wg_split_pair(gv[2 * q], gv[2 * q + 1], gp[q]);
        wg_bf16x8 gb[3];
#pragma unroll
        for (int p_ = 0; p_ < 3; ++p_) {
            const wg_u32x4 gq = {gp[0][p_], gp[1][p_], gp[2][p_], gp[3][p_]};
            gb[p_] = __builtin_bit_cast(wg_bf16x8, gq);
        }
#pragma unroll
        for (int t1 = 0; t1 < 3; ++t1) {
            const float* Ab = As + cur * ASZ + ((wave + t1) * ROWS + 8 * lh) * C + li;
            float xv[10];
#pragma unroll
            for (int e = 0; e < 10; ++e) xv[e] = Ab[e * C];
            unsigned xp[5][3];
#pragma unroll
            for (int q = 0; q < 5; ++q) wg_split_pair(xv[2 * q], xv[2 * q + 1], xp[q]);
            wg_bf16x8 xa[3][3];                               // [t2][piece]
#pragma unroll
            for (int p_ = 0; p_ < 3; ++p_) {
                const wg_u32x4 e0 = {xp[0][p_], xp[1][p_], xp[2][p_], xp[3][p_]};
                const wg_u32x4 e2 = {xp[1][p_], xp[2][p_], xp[3][p_], xp[4][p_]};
                wg_u32x4 e1;
#pragma unroll
                for (int q = 0; q < 4; ++q) e1[q] = (xp[q][p_] >> 16) | (xp[q + 1][p_] << 16);
                xa[0][p_] = __builtin_bit_cast(wg_bf16x8, e0);
                xa[1][p_] = __builtin_bit_cast(wg_bf16x8, e1);
                xa[2][p_] = __builtin_bit_cast(wg_bf16x8, e2);
            }
            constexpr int PX[6] = {2, 1, 0, 1, 0, 0}, PG[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                acc[t1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0][PX[k]], gb[PG[k]], acc[t1][0], 0, 0, 0);
                acc[t1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[1][PX[k]], gb[PG[k]], acc[t1][1], 0, 0, 0);
                acc[t1][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[2][PX[k]], gb[PG[k]], acc[t1][2], 0, 0, 0);
            }
        }
        setup_cols(n & 1, item + 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // sum the four waves' partial tiles through LDS three taps (one t1) at a time (48 KiB: the stage buffers are dead), one atomic per output
    float* red = reinterpret_cast<float*>(smem);        // [4 waves][3 taps][32 ca][32 cg]
#pragma unroll
    for (int t1 = 0; t1 < 3; ++t1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ca = (r & 3) + 8 * (r >> 2) + 4 * lh;
            red[((wave * 3 + 0) * C + ca) * C + li] = acc[t1][0][r];
            red[((wave * 3 + 1) * C + ca) * C + li] = acc[t1][1][r];
            red[((wave * 3 + 2) * C + ca) * C + li] = acc[t1][2][r];
        }
        __syncthreads();
        float* dwp = a.dw + (size_t)(t0 * 3 + t1) * 3 * C * C;       // taps (t0, t1, 0..2) are adjacent in [3,3,3,Ca,Cg]
        for (int e = tid; e < 3 * C * C; e += 256) {
            const float v = (red[e] + red[3 * C * C + e]) + (red[2 * 3 * C * C + e] + red[3 * 3 * C * C + e]);
            unsafeAtomicAdd(dwp + e, v);
        }
        __syncthreads();
    }
}

static int launch_wgrad_k3d32_row(const float* A, const float* G, float* dw, int B, int H, int W, int D,
                                  unsigned a_bytes, unsigned g_bytes, hipStream_t st)
{
    Wgrad3Args a;
    a.a = A; a.g = G; a.dw = dw; a.a_bytes = a_bytes; a.g_bytes = g_bytes;
    a.H = H; a.W = W; a.D = D;
    a.ncols = B * H * W;
    a.ndch = (D + 15) / 16;
    a.nitems = (a.ncols / 4) * a.ndch;                  // W % 4 == 0: whole groups of four columns
    static const int target_wgs = getenv("RN_WGRAD_WGS") ? atoi(getenv("RN_WGRAD_WGS")) : 3072;
    int ns = (target_wgs + 2) / 3;
    if (ns > (a.nitems + 15) / 16) ns = (a.nitems + 15) / 16;      // at least 16 stages per workgroup (every workgroup ends with 9216 atomics)
    if (ns < 1) ns = 1;
    a.ipw = (a.nitems + ns - 1) / ns;
    a.nsplit = (a.nitems + a.ipw - 1) / a.ipw;
    const long long nb = (long long)((a.nsplit + 7) / 8) * 24;
    const size_t stage = (size_t)2 * (112 * 32 + 64 * 32) * 4, redb = (size_t)4 * 3 * 32 * 32 * 4;
    const size_t lds = stage > redb ? stage : redb;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_wgrad_k3d32_row_kernel), lds); if (rc_ != RN_OK) return rc_; }
    hipLaunchKernelGGL(conv_wgrad_k3d32_row_kernel, dim3((unsigned)nb), dim3(256), lds, st, a);
    return rn_check_launch("conv_wgrad_k3d32_row (bf16x3)");
}

static int launch_wgrad_k3d32(const float* A, const float* G, float* dw, int B, int H, int W, int D,
                              unsigned a_bytes, unsigned g_bytes, hipStream_t st, bool split = false)
{
    Wgrad3Args a;
    a.a = A; a.g = G; a.dw = dw; a.a_bytes = a_bytes; a.g_bytes = g_bytes;
    a.H = H; a.W = W; a.D = D;
    a.ncols = B * H * W;
    a.ndch = (D + 31) / 32;
    a.nitems = ((a.ncols + 3) / 4) * a.ndch;
    static const int target_wgs = getenv("RN_WGRAD_WGS") ? atoi(getenv("RN_WGRAD_WGS")) : 3072;
    int ns = (target_wgs + 8) / 9;
    if (ns > (a.nitems + 3) / 4) ns = (a.nitems + 3) / 4;          // at least 4 stages per workgroup
    if (ns < 1) ns = 1;
    a.ipw = (a.nitems + ns - 1) / ns;
    a.nsplit = (a.nitems + a.ipw - 1) / a.ipw;
    const long long nb = (long long)((a.nsplit + 7) / 8) * 72;
    const size_t lds = (size_t)2 * (136 * 32 + 128 * 32) * 4;
    void (*kern)(const Wgrad3Args) = split ? conv_wgrad_k3d32_kernel<true> : conv_wgrad_k3d32_kernel<false>;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(lds)); if (rc_ != RN_OK) return rc_; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(256), lds, st, a);
    return rn_check_launch(split ? "conv_wgrad_k3d32 (bf16x3)" : "conv_wgrad_k3d32");
}

// 3x3x3, stride 1, 32 -> 32 (x [B,H,W,D,32], dz likewise) on the bf16 pipe: dw [3,3,3,32,32] += ...; batch chunks keep the byte offsets below 2^31
bool rn_conv3d_wgrad_split_ok(int Cin, int Cout)
{
    static const bool off = getenv("RN_NO_WGRAD3D_SPLIT") != nullptr;
    return !off && Cin == 32 && Cout == 32;
}

int rn_launch_conv3d_wgrad_split(const float* x, const float* dz, float* dw, int B, int H, int W, int D, hipStream_t st)
{
    if (!x || !dz || !dw) return rn_set_error(RN_E_INVALID, "conv3d_wgrad_split: null pointer");
    if (B < 1 || H < 1 || W < 1 || D < 1) return rn_set_error(RN_E_INVALID, "conv3d_wgrad_split: bad sizes");
    const long long item = (long long)H * W * D * 32 * 4;
    if (item >= 0x80000000LL) return rn_set_error(RN_E_UNSUPPORTED, "conv3d_wgrad_split: one batch item exceeds the 2 GiB buffer window");
    const int chunk = (int)(0x7fffffffLL / item);
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = B - b0 < chunk ? B - b0 : chunk;
        const size_t off = (size_t)b0 * (item / 4);
        // W % 4 == 0 (every map of the path): the row form -- three taps of a filter row per workgroup; RN_WGRAD3D_ROW=0: the nine-pair form
        static const bool row_off = getenv("RN_WGRAD3D_ROW") != nullptr && atoi(getenv("RN_WGRAD3D_ROW")) == 0;
        const int rc = (W % 4 == 0 && !row_off)
            ? launch_wgrad_k3d32_row(x + off, dz + off, dw, nb, H, W, D, (unsigned)(item * nb), (unsigned)(item * nb), st)
            : launch_wgrad_k3d32(x + off, dz + off, dw, nb, H, W, D, (unsigned)(item * nb), (unsigned)(item * nb), st, true);
        if (rc != RN_OK) return rc;
    }
    return RN_OK;
}

// A [B,I0,I1,I2,Ca], G [B,O0,O1,O2,Cg] -> dw [K0,K1,K2,Ca,Cg] (accumulated)
int rn_launch_conv_wgrad(const float* A, const float* G, float* dw, int B, const int* I, int Ca,
                         const int* O, int Cg, const int* K, const int* S, const int* P, hipStream_t st)
{
    if (!A || !G || !dw) return rn_set_error(RN_E_INVALID, "conv_wgrad: null pointer");
    if (B < 1 || Ca < 1 || Cg < 1) return rn_set_error(RN_E_INVALID, "conv_wgrad: bad sizes");
    const long long a_item = (long long)I[0] * I[1] * I[2] * Ca * 4;
    const long long g_item = (long long)O[0] * O[1] * O[2] * Cg * 4;
    if (a_item >= 0x80000000LL || g_item >= 0x80000000LL)
        return rn_set_error(RN_E_UNSUPPORTED, "conv_wgrad: one batch item exceeds the 2 GiB buffer window");
    if (a_item * B >= 0x80000000LL || g_item * B >= 0x80000000LL) {
        // operands are addressed with 32-bit byte offsets (upper half = hardware zero-fill): the
        // gradient is additive over the batch, so process batch chunks that fit the window
        const long long big = a_item > g_item ? a_item : g_item;
        const int chunk = (int)(0x7fffffffLL / big);
        for (int b0 = 0; b0 < B; b0 += chunk) {
            const int nb = (B - b0 < chunk) ? B - b0 : chunk;
            const int rc = rn_launch_conv_wgrad(A + (size_t)b0 * (a_item / 4), G + (size_t)b0 * (g_item / 4), dw,
                                                nb, I, Ca, O, Cg, K, S, P, st);
            if (rc != RN_OK) return rc;
        }
        return RN_OK;
    }
    WgradArgs a;
    a.a = A; a.g = G; a.dw = dw;
    a.a_bytes = (unsigned)(a_item * B); a.g_bytes = (unsigned)(g_item * B);
    a.I0 = I[0]; a.I1 = I[1]; a.I2 = I[2]; a.Ca = Ca;
    a.O0 = O[0]; a.O1 = O[1]; a.O2 = O[2]; a.Cg = Cg;
    a.K0 = K[0]; a.K1 = K[1]; a.K2 = K[2];
    a.S0 = S[0]; a.S1 = S[1]; a.S2 = S[2];
    a.P0 = P[0]; a.P1 = P[1]; a.P2 = P[2];
    const long long M = (long long)B * O[0] * O[1] * O[2];
    if (M <= 0 || M > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv_wgrad: M=%lld", M);
    a.M = (int)M;
    const int taps = K[0] * K[1] * K[2];
    static const bool no_k3d = getenv("RN_WGRAD_NO_K3D") != nullptr;
    if (!no_k3d && Ca == 32 && Cg == 32 && K[0] == 3 && K[1] == 3 && K[2] == 3 && S[0] == 1 && S[1] == 1 && S[2] == 1 &&
        P[0] == 1 && P[1] == 1 && P[2] == 1 && O[0] == I[0] && O[1] == I[1] && O[2] == I[2])
        return launch_wgrad_k3d32(A, G, dw, B, I[0], I[1], I[2], a.a_bytes, a.g_bytes, st);
    if (Ca % 4 != 0 || Cg % 4 != 0 || Ca < 8) {
        {   // LDS-staged narrow kernel where its shape limits allow
            const int rct = launch_wgrad_tile(a, B, st);
            if (rct != RN_E_UNSUPPORTED) return rct;
        }
        // narrow path: one thread per (tap, ca)
        if ((long long)taps * Ca > 1024 || Cg > 32)
            return rn_set_error(RN_E_UNSUPPORTED, "conv_wgrad: Ca=%d Cg=%d taps=%d has no kernel", Ca, Cg, taps);
        long long nblk = 4096;
        long long kc = (M + nblk - 1) / nblk;
        if (kc < 64) kc = 64;
        nblk = (M + kc - 1) / kc;
        a.kchunk = (int)kc; a.nsplit = (int)nblk; a.mtiles = a.ntiles = 1;
        const int nt = (taps * Ca + 63) / 64 * 64;
        if (Cg <= 8) hipLaunchKernelGGL(conv_wgrad_small_kernel<8>, dim3((unsigned)nblk), dim3(nt), 0, st, a);
        else if (Cg <= 16) hipLaunchKernelGGL(conv_wgrad_small_kernel<16>, dim3((unsigned)nblk), dim3(nt), 0, st, a);
        else hipLaunchKernelGGL(conv_wgrad_small_kernel<32>, dim3((unsigned)nblk), dim3(nt), 0, st, a);   // e_conv11 of the stress net (w10 = 32)
        return rn_check_launch("conv_wgrad_small");
    }
    const bool wide_m = Ca > 32, wide_n = Cg > 32;
    if (wide_m && wide_n) return launch_wgrad<128, 128, 32, 2, 2, 1>(a, st);
    if (wide_m) return launch_wgrad<128, 32, 64, 2, 1, 2>(a, st);
    if (wide_n) return launch_wgrad<32, 128, 64, 1, 2, 2>(a, st);
    return launch_wgrad<32, 32, 128, 1, 1, 4>(a, st);
}
