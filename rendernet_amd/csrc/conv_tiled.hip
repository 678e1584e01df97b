// LDS-tiled direct (VALU) convolution for the channel-starved stem and tail of the path, which are memory-bound and
// hopeless MFMA shapes:  e_conv1 (5^3 stride 2, Cin 1 | 5 -> 8; RenderNet_Shader.py:36-39,
// RenderNet_Texture_Face_Normal.py:52-55), e_conv2 (3^3 stride (1,1,2), 8 -> 16; :40-43) and e_conv11 (4x4 stride-1
// transposed conv = flipped conv, 16 -> 1 | 3, + sigmoid; :125-131).
//
// The generic direct kernel (conv_direct.hip) reads every tap of every output straight from L1/L2: 125 x Cin scattered
// loads per output (e_conv1: 0.83 ms for 402 MB of traffic, 5.0 ms with the texture net's 5 input channels).  Here a
// workgroup stages the input box of its T0 x T1 x T2 output tile in LDS with coalesced row loads (zero-filled where SAME
// padding applies) and the filter beside it; every thread then owns ONE output position and all CO channels.
// A tile whose input box is entirely zero -- 98 % of the 128^3 resampled grid is -- skips the arithmetic: its outputs
// are the constant act(bias).
#include "rn_common.h"

typedef float f32x4s __attribute__((ext_vector_type(4)));

struct TiledArgs2 {
    const float* x; const float* w; const float* bias; const float* alpha; const float* res; float* y; float* z;
    int B, I0, I1, I2, O0, O1, O2, Cout, Npad;
    int P0, P1, P2;
    int nt0, nt1, nt2;          // tiles per dim
    int act;
};

template <int K0, int K1, int K2, int S0, int S1, int S2, int CIN, int CO, int T0, int T1, int T2>
__global__ __launch_bounds__(T0 * T1 * T2)
void conv_tiled_kernel(const TiledArgs2 a)
{
    constexpr int NTH = T0 * T1 * T2;
    constexpr int IT0 = (T0 - 1) * S0 + K0, IT1 = (T1 - 1) * S1 + K1, IT2 = (T2 - 1) * S2 + K2;
    constexpr int ROWE = IT2 * CIN;                       // floats in one (r0, r1) row of the box in global memory
    // LDS pitches chosen so that the lanes of a wave (t2 fastest, then t1) read distinct banks: a position takes PP floats
    // (S2*PP*t2 mod 32 distinct for 16 lanes), a row an odd number of floats, and rows of a stride-2 conv are skewed by one
    // float per output row
    constexpr int PP = (CIN % 2 == 0) ? CIN + 1 : CIN;
    constexpr int SKEW = (S1 % 2 == 0) ? 1 : 0;
    constexpr int ROW = (IT2 * PP + SKEW * (IT1 / 2 + 1)) | 1;
    constexpr int KTOT = K0 * K1 * K2 * CIN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);           // [KTOT][CO]
    float* xt = wl + KTOT * CO;                           // [IT0][IT1][ROW]

    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    const int bt2 = blk % a.nt2; blk /= a.nt2;
    const int bt1 = blk % a.nt1; blk /= a.nt1;
    const int bt0 = blk % a.nt0; const int b = blk / a.nt0;
    const int in0 = bt0 * T0 * S0 - a.P0, in1 = bt1 * T1 * S1 - a.P1, in2 = bt2 * T2 * S2 - a.P2;

    for (int i = tid; i < KTOT * CO; i += NTH) {
        const int k = i / CO, n = i % CO;
        wl[i] = (n < a.Cout) ? a.w[((size_t)(k >> 2) * a.Npad + n) * 4 + (k & 3)] : 0.f;
    }
    // input box: rows of ROWE contiguous floats (depth x channel) in global memory
    bool any = false;
    for (int i = tid; i < IT0 * IT1 * ROWE; i += NTH) {
        const int e = i % ROWE, r = i / ROWE;
        const int r1 = r % IT1, r0 = r / IT1;
        const int i0 = in0 + r0, i1 = in1 + r1, i2 = in2 + e / CIN;
        float v = 0.f;
        if ((unsigned)i0 < (unsigned)a.I0 && (unsigned)i1 < (unsigned)a.I1 && (unsigned)i2 < (unsigned)a.I2)
            v = a.x[((((size_t)b * a.I0 + i0) * a.I1 + i1) * a.I2 + in2) * CIN + e];
        any = any || v != 0.f;
        xt[r * ROW + SKEW * (r1 >> 1) + (e / CIN) * PP + e % CIN] = v;
    }
    const bool work = __syncthreads_or(any ? 1 : 0) != 0;

    const int t2 = tid % T2, t1 = (tid / T2) % T1, t0 = tid / (T2 * T1);
    const int o0 = bt0 * T0 + t0, o1 = bt1 * T1 + t1, o2 = bt2 * T2 + t2;
    float acc[CO];
#pragma unroll
    for (int n = 0; n < CO; ++n) acc[n] = 0.f;
    if (work) {
        for (int k0 = 0; k0 < K0; ++k0)
            for (int k1 = 0; k1 < K1; ++k1) {
                const int r1 = t1 * S1 + k1;
                const float* xr = xt + ((t0 * S0 + k0) * IT1 + r1) * ROW + SKEW * (r1 >> 1) + t2 * S2 * PP;
                const float* wr = wl + (size_t)((k0 * K1 + k1) * K2) * CIN * CO;
#pragma unroll
                for (int k2 = 0; k2 < K2; ++k2)
#pragma unroll
                    for (int c = 0; c < CIN; ++c) {
                        const float xv = xr[k2 * PP + c];
#pragma unroll
                        for (int n = 0; n < CO; ++n) acc[n] = fmaf(xv, wr[(k2 * CIN + c) * CO + n], acc[n]);
                    }
            }
    }
    if (o0 >= a.O0 || o1 >= a.O1 || o2 >= a.O2) return;
    const size_t oo = ((((size_t)b * a.O0 + o0) * a.O1 + o1) * a.O2 + o2) * a.Cout;
#pragma unroll
    for (int n = 0; n < CO; ++n) {
        if (n < a.Cout) {
            float v = acc[n] + (a.bias ? a.bias[n] : 0.f);
            if (a.z) a.z[oo + n] = v;
            if (a.act & RN_ACT_PRELU) v = fmaxf(v, 0.f) + (a.alpha ? a.alpha[n] : 0.f) * fminf(v, 0.f);
            if (a.act & RN_ACT_ELU) v = v > 0.f ? v : expf(v) - 1.f;
            if (a.res) v += a.res[oo + n];
            if (a.act & RN_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
            a.y[oo + n] = v;
        }
    }
}

template <int K0, int K1, int K2, int S0, int S1, int S2, int CIN, int CO, int T0, int T1, int T2>
static int launch_tiled(const RnConvProblem& p, hipStream_t st)
{
    constexpr int IT0 = (T0 - 1) * S0 + K0, IT1 = (T1 - 1) * S1 + K1, IT2 = (T2 - 1) * S2 + K2;
    TiledArgs2 a;
    a.x = p.x; a.w = p.w; a.bias = p.bias; a.alpha = p.alpha; a.res = p.residual; a.y = p.y; a.z = p.preact;
    a.B = p.B; a.I0 = p.I[0]; a.I1 = p.I[1]; a.I2 = p.I[2]; a.O0 = p.O[0]; a.O1 = p.O[1]; a.O2 = p.O[2];
    a.Cout = p.Cout; a.Npad = p.Npad; a.P0 = p.P[0]; a.P1 = p.P[1]; a.P2 = p.P[2];
    a.nt0 = (p.O[0] + T0 - 1) / T0; a.nt1 = (p.O[1] + T1 - 1) / T1; a.nt2 = (p.O[2] + T2 - 1) / T2;
    a.act = p.act;
    const long long nb = (long long)p.B * a.nt0 * a.nt1 * a.nt2;
    if (nb <= 0 || nb > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv_tiled: bad grid %lld", nb);
    constexpr int PP = (CIN % 2 == 0) ? CIN + 1 : CIN;
    constexpr int ROW = (IT2 * PP + ((S1 % 2 == 0) ? IT1 / 2 + 1 : 0)) | 1;
    const size_t lds = ((size_t)K0 * K1 * K2 * CIN * CO + (size_t)IT0 * IT1 * ROW + 16) * sizeof(float);
    auto kern = conv_tiled_kernel<K0, K1, K2, S0, S1, S2, CIN, CO, T0, T1, T2>;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(lds)); if (rc_ != RN_OK) return rc_; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(T0 * T1 * T2), lds, st, a);
    return rn_check_launch("conv_tiled");
}

// ------------------------------------------------------------------------------------------------------------------
// The 5-channel stem of the texture net (e_conv1: 5^3 stride 2, 5 -> 8; RenderNet_Texture_Face_Normal.py:52-55).  Its
// input box does not fit LDS at a useful tile size (5 channels x a (2T+3)^3 halo), and with one output per thread the
// 625 taps x 8 channels are bound by the broadcast filter reads (two ds_read_b128 per input value: 4.9-5.5 ms).  Here a
// thread owns OPT consecutive outputs along the depth axis: per (k0, k1) filter row it loads its window of
// ((OPT-1)*S2 + K2) positions x CIN floats -- contiguous in channels-last memory -- ONCE into registers, and every filter
// value read from LDS feeds OPT outputs.
// ------------------------------------------------------------------------------------------------------------------
template <int K0, int K1, int K2, int S0, int S1, int S2, int CIN, int CO, int OPT>
__global__ __launch_bounds__(256)
void conv_rows_kernel(const TiledArgs2 a)
{
    constexpr int KTOT = K0 * K1 * K2 * CIN;
    constexpr int WIN = ((OPT - 1) * S2 + K2) * CIN;       // floats in a thread's row window
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);           // [KTOT][CO]
    for (int i = threadIdx.x; i < KTOT * CO; i += 256) {
        const int k = i / CO, n = i % CO;
        wl[i] = (n < a.Cout) ? a.w[((size_t)(k >> 2) * a.Npad + n) * 4 + (k & 3)] : 0.f;
    }
    __syncthreads();
    const int nq = (a.O2 + OPT - 1) / OPT;                // output groups along depth
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= (long long)a.B * a.O0 * a.O1 * nq) return;
    long long t = m;
    const int q = (int)(t % nq); t /= nq;
    const int o1 = (int)(t % a.O1); t /= a.O1;
    const int o0 = (int)(t % a.O0); const int b = (int)(t / a.O0);
    const int i2base = q * OPT * S2 - a.P2;

    float acc[OPT][CO];
#pragma unroll
    for (int j = 0; j < OPT; ++j)
#pragma unroll
        for (int n = 0; n < CO; ++n) acc[j][n] = 0.f;
    for (int k0 = 0; k0 < K0; ++k0) {
        const int i0 = o0 * S0 - a.P0 + k0;
        if ((unsigned)i0 >= (unsigned)a.I0) continue;
        for (int k1 = 0; k1 < K1; ++k1) {
            const int i1 = o1 * S1 - a.P1 + k1;
            if ((unsigned)i1 >= (unsigned)a.I1) continue;
            const float* xr = a.x + (((size_t)b * a.I0 + i0) * a.I1 + i1) * a.I2 * CIN;
            float xw[WIN];
#pragma unroll
            for (int e = 0; e < WIN; ++e) {
                const int i2 = i2base + e / CIN;
                xw[e] = ((unsigned)i2 < (unsigned)a.I2) ? xr[(long long)i2base * CIN + e] : 0.f;
            }
            const float* wr = wl + (size_t)((k0 * K1 + k1) * K2) * CIN * CO;
#pragma unroll
            for (int k2 = 0; k2 < K2; ++k2)
#pragma unroll
                for (int c = 0; c < CIN; ++c) {
                    float wv[CO];
#pragma unroll
                    for (int n = 0; n < CO; ++n) wv[n] = wr[(k2 * CIN + c) * CO + n];
#pragma unroll
                    for (int j = 0; j < OPT; ++j) {
                        const float xv = xw[(j * S2 + k2) * CIN + c];
#pragma unroll
                        for (int n = 0; n < CO; ++n) acc[j][n] = fmaf(xv, wv[n], acc[j][n]);
                    }
                }
        }
    }
#pragma unroll
    for (int j = 0; j < OPT; ++j) {
        const int o2 = q * OPT + j;
        if (o2 >= a.O2) break;
        const size_t oo = ((((size_t)b * a.O0 + o0) * a.O1 + o1) * a.O2 + o2) * a.Cout;
#pragma unroll
        for (int n = 0; n < CO; ++n) {
            if (n < a.Cout) {
                float v = acc[j][n] + (a.bias ? a.bias[n] : 0.f);
                if (a.z) a.z[oo + n] = v;
                if (a.act & RN_ACT_PRELU) v = fmaxf(v, 0.f) + (a.alpha ? a.alpha[n] : 0.f) * fminf(v, 0.f);
                if (a.act & RN_ACT_ELU) v = v > 0.f ? v : expf(v) - 1.f;
                if (a.res) v += a.res[oo + n];
                if (a.act & RN_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                a.y[oo + n] = v;
            }
        }
    }
}

template <int K0, int K1, int K2, int S0, int S1, int S2, int CIN, int CO, int OPT>
static int launch_rows(const RnConvProblem& p, hipStream_t st)
{
    TiledArgs2 a;
    a.x = p.x; a.w = p.w; a.bias = p.bias; a.alpha = p.alpha; a.res = p.residual; a.y = p.y; a.z = p.preact;
    a.B = p.B; a.I0 = p.I[0]; a.I1 = p.I[1]; a.I2 = p.I[2]; a.O0 = p.O[0]; a.O1 = p.O[1]; a.O2 = p.O[2];
    a.Cout = p.Cout; a.Npad = p.Npad; a.P0 = p.P[0]; a.P1 = p.P[1]; a.P2 = p.P[2];
    a.nt0 = a.nt1 = a.nt2 = 0; a.act = p.act;
    const long long threads = (long long)p.B * p.O[0] * p.O[1] * ((p.O[2] + OPT - 1) / OPT);
    const long long nb = (threads + 255) / 256;
    if (nb <= 0 || nb > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv_rows: bad grid %lld", nb);
    const size_t lds = (size_t)K0 * K1 * K2 * CIN * CO * sizeof(float);
    auto kern = conv_rows_kernel<K0, K1, K2, S0, S1, S2, CIN, CO, OPT>;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(lds)); if (rc_ != RN_OK) return rc_; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(256), lds, st, a);
    return rn_check_launch("conv_rows");
}

// ------------------------------------------------------------------------------------------------------------------
// conv_stem_kernel: the 5^3 stride-2 stems with a handful of input channels (e_conv1: Cin 1 | 5 -> 8), depth rows staged in
// LDS.  A workgroup = RPW/4 waves owns RPW output rows (o0, RPW*g ..) of one item: wave w rows 4w .. 4w+3, a lane one row
// (lane >> 4) and FOUR consecutive output depths 4*(lane & 15) .. +3 (chunks of 64 depths when O2 > 64).  For every input plane
// i0 = 2*o0 - P0 + k0 the 35 input rows i1 those output rows touch go global -> LDS with whole-row coalesced loads
// ((I2 + 10) positions x CIN floats per row: SAME padding = zero positions on both sides; 3 of a row's 5 uses are by a
// neighbouring output row), together with a per-row "any non-zero" flag; a lane then walks its five rows: the 11-position
// window of its four outputs is read ONCE into registers (contiguous in the staged row), every filter value -- a wave-uniform
// 16-byte LDS broadcast per 4 output channels -- feeds four outputs.  Rows that are all zero for every lane of the wave are
// skipped (most of the resampled grid is empty).
// ------------------------------------------------------------------------------------------------------------------
template <int CIN, int CO, int RPW>
__global__ __launch_bounds__(RPW * 16)
void conv_stem_kernel(const TiledArgs2 a)
{
    constexpr int K = 5, S = 2, NROW = (RPW - 1) * S + K;          // RPW output rows per workgroup (RPW/4 waves), staged rows
    constexpr int NTH = RPW * 16, NW = RPW / 4;
    constexpr int NPOS = 4, WIN = ((NPOS - 1) * S + K) * CIN;         // 4 outputs per lane: 11 positions x CIN floats
    constexpr int KTOT = K * K * K * CIN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);                    // [KTOT][CO]
    constexpr int LH = 4;                                          // zero positions before the row (LH*CIN floats: 16-byte multiple)
    const int rowlen = (a.I2 + LH + 8) * CIN;                      // floats per staged row: LH zero positions before, 8 after (SAME
                                                                   // padding + the window of a lane whose last outputs are past O2)
    const int rowpitch = rowlen | 1;                               // odd pitch: the four rows of a wave start in different banks
    float* rows = wl + KTOT * CO;                                  // [NROW][rowpitch]
    int* flags = reinterpret_cast<int*>(rows + NROW * rowpitch);   // [NROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int blk = blockIdx.x;
    const int ng = (a.O1 + RPW - 1) / RPW;
    const int g = blk % ng; blk /= ng;
    const int o0 = blk % a.O0; const int b = blk / a.O0;
    const int rl = wave * 4 + (lane >> 4);                         // output row within the workgroup
    const int o1 = g * RPW + rl;

    for (int i = tid; i < KTOT * CO; i += NTH) {
        const int k = i / CO, n = i % CO;
        wl[i] = (n < a.Cout) ? a.w[((size_t)(k >> 2) * a.Npad + n) * 4 + (k & 3)] : 0.f;
    }
    for (int i = tid; i < NROW * rowpitch; i += NTH) rows[i] = 0.f;      // the halos stay zero for the whole kernel
    const int nf4 = a.I2 * CIN / 4;                                // 16-byte pieces of an input row (the launcher checks divisibility)
    const int nchunk = (a.O2 + 63) / 64;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int o2 = ch * 64 + (lane & 15) * NPOS;               // first of this lane's four output depths
        float acc[NPOS][CO];
#pragma unroll
        for (int q = 0; q < NPOS; ++q)
#pragma unroll
            for (int n = 0; n < CO; ++n) acc[q][n] = 0.f;
        // window start in the staged row: position 2*o2 - P2, the row itself starts at position -2; clamped so that lanes
        // past the end of the depth axis (their outputs are not stored) stay inside the row
        const int w0 = min((o2 * S - a.P2 + LH) * CIN, rowlen - WIN);
        for (int k0 = 0; k0 < K; ++k0) {
            const int i0 = o0 * S - a.P0 + k0;
            if ((unsigned)i0 >= (unsigned)a.I0) continue;          // uniform per workgroup
            __syncthreads();                                       // the previous plane's rows are consumed
            if (tid < NROW) flags[tid] = 0;
            __syncthreads();
            // stage the rows: row j <- input row i1 = 32*g - P1 + j.  Wave w takes rows w, w+4, ...; its (row, 16-byte piece)
            // pairs are walked in batches of 8 per lane with all loads of a batch issued before the first LDS write
            {
                const int nrw = (NROW - wave + NW - 1) / NW;       // rows of this wave
                const int total = nrw * nf4;
                const float* xp = a.x + (((size_t)b * a.I0 + i0) * a.I1) * a.I2 * CIN;
                for (int base = 0; base < total; base += 64 * 8) {
                    f32x4s v[8]; int dst[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int idx = base + u * 64 + lane;
                        dst[u] = -1;
                        v[u] = f32x4s{0.f, 0.f, 0.f, 0.f};
                        if (idx < total) {
                            const int r = idx / nf4, f = idx - r * nf4;
                            const int j = wave + NW * r;
                            const int i1 = g * RPW * S - a.P1 + j;
                            dst[u] = j * rowpitch + LH * CIN + f * 4;
                            if ((unsigned)i1 < (unsigned)a.I1)
                                v[u] = *reinterpret_cast<const f32x4s*>(xp + (size_t)i1 * a.I2 * CIN + f * 4);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (dst[u] >= 0) {
                            float* d = rows + dst[u];
                            d[0] = v[u][0]; d[1] = v[u][1]; d[2] = v[u][2]; d[3] = v[u][3];
                            if (v[u][0] != 0.f || v[u][1] != 0.f || v[u][2] != 0.f || v[u][3] != 0.f)
                                flags[(dst[u] - LH * CIN) / rowpitch] = 1;         // benign race: every writer stores 1
                        }
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int k1 = 0; k1 < K; ++k1) {
                const int j = rl * S + k1;
                if (!__any(flags[j] != 0)) continue;               // all four rows of this wave are empty here
                const float* xr = rows + j * rowpitch + w0;
                float xw[WIN];
#pragma unroll
                for (int e = 0; e < WIN; ++e) xw[e] = xr[e];
                const float* wr = wl + (size_t)((k0 * K + k1) * K) * CIN * CO;
#pragma unroll
                for (int e = 0; e < K * CIN; ++e) {
                    float wv[CO];
#pragma unroll
                    for (int n = 0; n < CO; ++n) wv[n] = wr[e * CO + n];
#pragma unroll
                    for (int q = 0; q < NPOS; ++q) {
                        const float xv = xw[q * S * CIN + e];
#pragma unroll
                        for (int n = 0; n < CO; ++n) acc[q][n] = fmaf(xv, wv[n], acc[q][n]);
                    }
                }
            }
        }
        if (o1 < a.O1) {
#pragma unroll
            for (int q = 0; q < NPOS; ++q) {
                if (o2 + q >= a.O2) break;
                const size_t oo = ((((size_t)b * a.O0 + o0) * a.O1 + o1) * a.O2 + o2 + q) * a.Cout;
#pragma unroll
                for (int n = 0; n < CO; ++n) {
                    if (n < a.Cout) {
                        float v = acc[q][n] + (a.bias ? a.bias[n] : 0.f);
                        if (a.z) a.z[oo + n] = v;
                        if (a.act & RN_ACT_PRELU) v = fmaxf(v, 0.f) + (a.alpha ? a.alpha[n] : 0.f) * fminf(v, 0.f);
                        if (a.act & RN_ACT_ELU) v = v > 0.f ? v : expf(v) - 1.f;
                        if (a.res) v += a.res[oo + n];
                        if (a.act & RN_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                        a.y[oo + n] = v;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// conv_stem_dc_kernel: conv_stem_kernel with the depth axis cut into chunks of DCH = 16*NPOS outputs PER WORKGROUP.
// PMC on conv_stem_kernel<5,8,16> (texture net stem, B=24): vector ALU 33 % busy, LDS 33 % busy (54 % of that bank
// conflicts), 38 % of the wave cycles waiting -- with 98 KB of staged rows there is one workgroup = one wave per SIMD on a CU
// and nothing to hide its LDS / staging latency behind.  Here a workgroup stages only the row SEGMENTS its 32 output depths
// touch (72 positions instead of 140: 51 KB + the 20 KB filter), a lane owns NPOS = 2 consecutive outputs, and TWO 256-thread
// workgroups fit a CU.  Same arithmetic per output (same tap order): bit-identical to conv_stem_kernel.
// ------------------------------------------------------------------------------------------------------------------
template <int CIN, int CO, int RPW, int NPOS>
__global__ __launch_bounds__(RPW * 16, 2)
void conv_stem_dc_kernel(const TiledArgs2 a)
{
    constexpr int K = 5, S = 2, NROW = (RPW - 1) * S + K;
    constexpr int NTH = RPW * 16, NW = RPW / 4;
    constexpr int DCH = 16 * NPOS;                                   // output depths per workgroup
    constexpr int WIN = ((NPOS - 1) * S + K) * CIN;                  // a lane's window in a staged row (floats)
    constexpr int NPS = ((DCH - 1) * S + K + 3 + 3) / 4 * 4;         // staged positions per row: the chunk's span + alignment slack, multiple of 4
    constexpr int ROWLEN = NPS * CIN, PITCH = ROWLEN | 1;            // odd pitch: the rows of a wave start in different banks
    constexpr int KTOT = K * K * K * CIN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);                      // [KTOT][CO]
    float* rows = wl + KTOT * CO;                                    // [NROW][PITCH]
    int* flags = reinterpret_cast<int*>(rows + NROW * PITCH);        // [NROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int blk = blockIdx.x;
    const int nch = (a.O2 + DCH - 1) / DCH;
    const int ch = blk % nch; blk /= nch;
    const int ng = (a.O1 + RPW - 1) / RPW;
    const int g = blk % ng; blk /= ng;
    const int o0 = blk % a.O0; const int b = blk / a.O0;
    const int rl = wave * 4 + (lane >> 4);
    const int o1 = g * RPW + rl;
    const int c0 = ch * DCH;                                          // first output depth of the chunk
    // first staged position: 2*c0 - P2 rounded down to a multiple of 4 (so that segment and row ends fall on 16-byte pieces;
    // the launcher checks I2 % 4 == 0).  May be negative: pieces outside [0, I2) stay zero = SAME padding.
    const int pstart = ((c0 * S - a.P2) & ~3);
    for (int i = tid; i < KTOT * CO; i += NTH) {
        const int k = i / CO, n = i % CO;
        wl[i] = (n < a.Cout) ? a.w[((size_t)(k >> 2) * a.Npad + n) * 4 + (k & 3)] : 0.f;
    }
    constexpr int NF4 = ROWLEN / 4;                                   // 16-byte pieces of a staged segment
    const int o2 = c0 + (lane & 15) * NPOS;                           // first of this lane's outputs
    const int w0 = (o2 * S - a.P2 - pstart) * CIN;                    // its window's first float in a staged row (>= 0, + WIN <= ROWLEN)
    float acc[NPOS][CO];
#pragma unroll
    for (int q = 0; q < NPOS; ++q)
#pragma unroll
        for (int n = 0; n < CO; ++n) acc[q][n] = 0.f;
    for (int k0 = 0; k0 < K; ++k0) {
        const int i0 = o0 * S - a.P0 + k0;
        if ((unsigned)i0 >= (unsigned)a.I0) continue;                 // uniform per workgroup
        __syncthreads();                                              // the previous plane's rows are consumed
        if (tid < NROW) flags[tid] = 0;
        __syncthreads();
        {
            const int nrw = (NROW - wave + NW - 1) / NW;              // rows of this wave
            const int total = nrw * NF4;
            const float* xp = a.x + (((size_t)b * a.I0 + i0) * a.I1) * a.I2 * CIN;
            for (int base = 0; base < total; base += 64 * 8) {
                f32x4s v[8]; int dst[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 64 + lane;
                    dst[u] = -1;
                    v[u] = f32x4s{0.f, 0.f, 0.f, 0.f};
                    if (idx < total) {
                        const int r = idx / NF4, f = idx - r * NF4;
                        const int j = wave + NW * r;
                        const int i1 = g * RPW * S - a.P1 + j;
                        const int gf = pstart * CIN + f * 4;          // first float of this piece in the global row
                        dst[u] = j * PITCH + f * 4;
                        if ((unsigned)i1 < (unsigned)a.I1 && gf >= 0 && gf + 3 < a.I2 * CIN)
                            v[u] = *reinterpret_cast<const f32x4s*>(xp + (size_t)i1 * a.I2 * CIN + gf);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (dst[u] >= 0) {
                        float* d = rows + dst[u];
                        d[0] = v[u][0]; d[1] = v[u][1]; d[2] = v[u][2]; d[3] = v[u][3];
                        if (v[u][0] != 0.f || v[u][1] != 0.f || v[u][2] != 0.f || v[u][3] != 0.f)
                            flags[dst[u] / PITCH] = 1;                // benign race: every writer stores 1
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k1 = 0; k1 < K; ++k1) {
            const int j = rl * S + k1;
            if (!__any(flags[j] != 0)) continue;                      // all four rows of this wave are empty here
            const float* xr = rows + j * PITCH + w0;
            float xw[WIN];
#pragma unroll
            for (int e = 0; e < WIN; ++e) xw[e] = xr[e];
            const float* wr = wl + (size_t)((k0 * K + k1) * K) * CIN * CO;
#pragma unroll
            for (int e = 0; e < K * CIN; ++e) {
                float wv[CO];
#pragma unroll
                for (int n = 0; n < CO; ++n) wv[n] = wr[e * CO + n];
#pragma unroll
                for (int q = 0; q < NPOS; ++q) {
                    const float xv = xw[q * S * CIN + e];
#pragma unroll
                    for (int n = 0; n < CO; ++n) acc[q][n] = fmaf(xv, wv[n], acc[q][n]);
                }
            }
        }
    }
    if (o1 < a.O1) {
#pragma unroll
        for (int q = 0; q < NPOS; ++q) {
            if (o2 + q >= a.O2) break;
            const size_t oo = ((((size_t)b * a.O0 + o0) * a.O1 + o1) * a.O2 + o2 + q) * a.Cout;
#pragma unroll
            for (int n = 0; n < CO; ++n) {
                if (n < a.Cout) {
                    float v = acc[q][n] + (a.bias ? a.bias[n] : 0.f);
                    if (a.z) a.z[oo + n] = v;
                    if (a.act & RN_ACT_PRELU) v = fmaxf(v, 0.f) + (a.alpha ? a.alpha[n] : 0.f) * fminf(v, 0.f);
                    if (a.act & RN_ACT_ELU) v = v > 0.f ? v : expf(v) - 1.f;
                    if (a.res) v += a.res[oo + n];
                    if (a.act & RN_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                    a.y[oo + n] = v;
                }
            }
        }
    }
}

template <int CIN, int CO, int RPW, int NPOS>
static int launch_stem_dc(const RnConvProblem& p, hipStream_t st)
{
    TiledArgs2 a;
    a.x = p.x; a.w = p.w; a.bias = p.bias; a.alpha = p.alpha; a.res = p.residual; a.y = p.y; a.z = p.preact;
    a.B = p.B; a.I0 = p.I[0]; a.I1 = p.I[1]; a.I2 = p.I[2]; a.O0 = p.O[0]; a.O1 = p.O[1]; a.O2 = p.O[2];
    a.Cout = p.Cout; a.Npad = p.Npad; a.P0 = p.P[0]; a.P1 = p.P[1]; a.P2 = p.P[2];
    a.nt0 = a.nt1 = a.nt2 = 0; a.act = p.act;
    constexpr int DCH = 16 * NPOS, NROW = (RPW - 1) * 2 + 5;
    constexpr int NPS = ((DCH - 1) * 2 + 5 + 3 + 3) / 4 * 4, PITCH = (NPS * CIN) | 1;
    const long long nb = (long long)p.B * p.O[0] * ((p.O[1] + RPW - 1) / RPW) * ((p.O[2] + DCH - 1) / DCH);
    if (nb <= 0 || nb > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv_stem_dc: bad grid %lld", nb);
    const size_t lds = ((size_t)125 * CIN * CO + (size_t)NROW * PITCH) * sizeof(float) + NROW * sizeof(int);
    // 16-byte pieces must not straddle a row end, windows must stay inside the staged segment (pad_before <= 3)
    if (lds > 80 * 1024 || p.I[2] % 4 != 0 || (p.I[2] * CIN) % 4 != 0 || p.P[2] > 3 || (reinterpret_cast<size_t>(p.x) & 15) != 0) return RN_E_UNSUPPORTED;
    auto kern = conv_stem_dc_kernel<CIN, CO, RPW, NPOS>;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc_ != RN_OK) return rc_; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(RPW * 16), lds, st, a);
    return rn_check_launch("conv_stem_dc");
}

template <int CIN, int CO, int RPW>
static int launch_stem(const RnConvProblem& p, hipStream_t st)
{
    TiledArgs2 a;
    a.x = p.x; a.w = p.w; a.bias = p.bias; a.alpha = p.alpha; a.res = p.residual; a.y = p.y; a.z = p.preact;
    a.B = p.B; a.I0 = p.I[0]; a.I1 = p.I[1]; a.I2 = p.I[2]; a.O0 = p.O[0]; a.O1 = p.O[1]; a.O2 = p.O[2];
    a.Cout = p.Cout; a.Npad = p.Npad; a.P0 = p.P[0]; a.P1 = p.P[1]; a.P2 = p.P[2];
    a.nt0 = a.nt1 = a.nt2 = 0; a.act = p.act;
    const long long nb = (long long)p.B * p.O[0] * ((p.O[1] + RPW - 1) / RPW);
    if (nb <= 0 || nb > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv_stem: bad grid %lld", nb);
    const int rowpitch = ((p.I[2] + 12) * CIN) | 1;
    constexpr int NROW = (RPW - 1) * 2 + 5;
    const size_t lds = ((size_t)125 * CIN * CO + (size_t)NROW * rowpitch) * sizeof(float) + NROW * sizeof(int);
    if (lds > 150 * 1024 || (p.I[2] * CIN) % 4 != 0 || (reinterpret_cast<size_t>(p.x) & 15) != 0) return RN_E_UNSUPPORTED;
    auto kern = conv_stem_kernel<CIN, CO, RPW>;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc_ != RN_OK) return rc_; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(RPW * 16), lds, st, a);
    return rn_check_launch("conv_stem");
}

// ------------------------------------------------------------------------------------------------------------------
// The tail of the decoder: e_conv11 -- slim.conv2d_transpose 4x4 stride 1, 16 -> 1 | 3 channels, + sigmoid (RenderNet_Shader.py:125-131;
// as a gridded problem: 4x4 taps over [B,H,W,16], pad_lo (2,2)).  403 MB read and 25 MB written at B = 24: an HBM kernel.  The LDS-tiled
// form above (one output per thread, the 19 x 19 x 16 input box staged per 16 x 16 tile) took 0.30 ms = 0.18 of HBM speed: 256 broadcast
// filter reads and 256 scalar LDS reads per output.  Here a WAVE owns a strip of 16 output columns and walks down the rows:
//   * lane = (pixel g of the 16, channel quad q): a wave load is 16 pixels x 64 B = 1 KiB contiguous, four of them (the four column
//     taps) per input row, the next row's in flight while this one multiplies;
//   * an input row feeds the FOUR output rows it belongs to (row taps t0 = 0..3): four running accumulators per output channel, the
//     slot of an output row = row % 4 -- compile-time, the row loop is unrolled by four;
//   * when an output row has its fourth input row, the four quads of a pixel are summed with two DPP-class shuffles and lane q = 0 of
//     every pixel stores: 64 contiguous bytes per wave and row at one output channel.
// NQ = Cin / 4 channel quads per pixel (4: the 16-channel tail of the 64^3 -> 512^2 nets; 8: the 32-channel one of the 128^3 -> 1024^2 config, which ran on
// the generic direct kernel at 13.3 ms per call until round 6): a wave covers 64 / NQ output columns.
template <int CO, int NQ>
__global__ __launch_bounds__(256)
void conv_tail_kernel(const TiledArgs2 a, int R, int ncs, int nrs)
{
    constexpr int CIN = 4 * NQ, PW = 64 / NQ;                // input channels; pixels (output columns) per wave
    __shared__ f32x4s wl[16 * NQ * CO];                   // [tap][quad][n] = the packed filter's (K quad, n) element: k = tap * CIN + 4 quad + e
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16 * NQ * CO; i += 256) {
        const int n = i % CO, tq = i / CO;
        wl[i] = (n < a.Cout) ? *reinterpret_cast<const f32x4s*>(a.w + ((size_t)tq * a.Npad + n) * 4) : f32x4s{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    long long id = (long long)blockIdx.x * 4 + wave;                  // strip: (image, row strip, column strip)
    if (id >= (long long)a.B * nrs * ncs) return;
    const int cs = (int)(id % ncs); id /= ncs;
    const int rs = (int)(id % nrs); const int b = (int)(id / nrs);
    const int g = lane / NQ, q = lane % NQ;
    const int col = cs * PW + g, row0 = rs * R;                       // R % 4 == 0, so row0 % 4 == 0
    const int H = a.I0, W = a.I1;
    const float* xb = a.x + (size_t)b * H * W * CIN + 4 * q;
    // column taps: input column col - P1 + t1
    bool cok[4];
    int coff[4];
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1) {
        const int ic = col - a.P1 + t1;
        cok[t1] = (unsigned)ic < (unsigned)W;
        coff[t1] = (cok[t1] ? ic : 0) * CIN;
    }
    auto load_row = [&](int u, f32x4s (&xv)[4]) {                     // u = input row + P0
        const int ir = u - a.P0;
        const bool rok = (unsigned)ir < (unsigned)H && u < row0 + R + 3;
        const float* xr = xb + (size_t)(rok ? ir : 0) * W * CIN;
#pragma unroll
        for (int t1 = 0; t1 < 4; ++t1)
            xv[t1] = (rok && cok[t1]) ? *reinterpret_cast<const f32x4s*>(xr + coff[t1]) : f32x4s{0.f, 0.f, 0.f, 0.f};
    };
    typedef float f32x2s __attribute__((ext_vector_type(2)));
    f32x2s acc[4][CO];                                                // two partial sums per accumulator: packed FMAs (v_pk_fma_f32) on channel pairs
#pragma unroll
    for (int sl = 0; sl < 4; ++sl)
#pragma unroll
        for (int n = 0; n < CO; ++n) acc[sl][n] = f32x2s{0.f, 0.f};
    const bool colok = col < a.O1;
    // input rows u .. u + 3 live in a ring of four register rows (row u in xr[u % 4], compile-time after unrolling): while row u
    // multiplies, rows u + 1 .. u + 3 are in flight -- with one row ahead the waves waited on memory 70 % of their cycles (0.205 ms)
    f32x4s xr[4][4];
    load_row(row0, xr[0]);
    load_row(row0 + 1, xr[1]);
    load_row(row0 + 2, xr[2]);
    const int uend = row0 + R + 3;                                    // u = row0 .. row0 + R + 2 (output rows u - t0 inside the strip only)
    for (int u0 = row0; u0 < uend; u0 += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = u0 + j;
            if (u >= uend) break;                                     // uniform
            load_row(u + 3, xr[(j + 3) & 3]);                         // (rows beyond the strip's last input row load nothing)
#pragma unroll
            for (int t0 = 0; t0 < 4; ++t0) {
                const int o0 = u - t0;
                if (o0 < row0 || o0 >= row0 + R) continue;            // uniform: the strip's first / last three input rows
                const int sl = (j - t0) & 3;                          // = o0 % 4, compile-time after unrolling
#pragma unroll
                for (int t1 = 0; t1 < 4; ++t1)
#pragma unroll
                    for (int n = 0; n < CO; ++n) {
                        const f32x4s w = wl[((t0 * 4 + t1) * NQ + q) * CO + n];
                        acc[sl][n] = __builtin_elementwise_fma(f32x2s{xr[j][t1][0], xr[j][t1][1]}, f32x2s{w[0], w[1]}, acc[sl][n]);
                        acc[sl][n] = __builtin_elementwise_fma(f32x2s{xr[j][t1][2], xr[j][t1][3]}, f32x2s{w[2], w[3]}, acc[sl][n]);
                    }
            }
            // output row u - 3 has all four of its input rows
            const int od = u - 3;
            if (od >= row0) {                                         // uniform
                const int sl = (j + 1) & 3;                           // = (j - 3) & 3
#pragma unroll
                for (int n = 0; n < CO; ++n) {
                    float v = acc[sl][n][0] + acc[sl][n][1];
                    v += __shfl_xor(v, 1);
                    v += __shfl_xor(v, 2);
                    if (NQ == 8) v += __shfl_xor(v, 4);
                    acc[sl][n] = f32x2s{0.f, 0.f};
                    if (q == 0 && colok && od < a.O0 && n < a.Cout) {
                        const size_t oo = (((size_t)b * a.O0 + od) * a.O1 + col) * a.Cout + n;
                        v += a.bias ? a.bias[n] : 0.f;
                        if (a.z) a.z[oo] = v;
                        if (a.act & RN_ACT_PRELU) v = fmaxf(v, 0.f) + (a.alpha ? a.alpha[n] : 0.f) * fminf(v, 0.f);
                        if (a.act & RN_ACT_ELU) v = v > 0.f ? v : expf(v) - 1.f;
                        if (a.res) v += a.res[oo];
                        if (a.act & RN_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                        a.y[oo] = v;
                    }
                }
            }
        }
    }
}

template <int CO, int NQ>
static int launch_tail(const RnConvProblem& p, hipStream_t st)
{
    TiledArgs2 a;
    a.x = p.x; a.w = p.w; a.bias = p.bias; a.alpha = p.alpha; a.res = p.residual; a.y = p.y; a.z = p.preact;
    a.B = p.B; a.I0 = p.I[0]; a.I1 = p.I[1]; a.I2 = 1; a.O0 = p.O[0]; a.O1 = p.O[1]; a.O2 = 1;
    a.Cout = p.Cout; a.Npad = p.Npad; a.P0 = p.P[0]; a.P1 = p.P[1]; a.P2 = 0;
    a.nt0 = a.nt1 = a.nt2 = 0; a.act = p.act;
    static const int r_env = getenv("RN_TAIL_ROWS") ? atoi(getenv("RN_TAIL_ROWS")) : 0;
    const int R = (r_env >= 4 && r_env % 4 == 0) ? r_env : 32;        // rows per strip (three halo rows re-read per strip; measured 8: 0.210, 16: 0.196, 32: 0.191, 64: 0.215 ms)
    constexpr int PW = 64 / NQ;
    const int ncs = (p.O[1] + PW - 1) / PW, nrs = (p.O[0] + R - 1) / R;
    const long long nw = (long long)p.B * nrs * ncs;
    if (nw <= 0 || (nw + 3) / 4 > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv_tail: bad grid %lld", nw);
    hipLaunchKernelGGL((conv_tail_kernel<CO, NQ>), dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, st, a, R, ncs, nrs);
    return rn_check_launch("conv_tail");
}

static bool is_plain(const RnConvProblem& p)
{
    // contiguous channels-last output covering the whole grid (no sub-pixel phase addressing)
    return p.out_off == 0 && p.os[2] == p.Cout && p.os[1] == (long long)p.O[2] * p.Cout &&
           p.os[0] == (long long)p.O[1] * p.O[2] * p.Cout && p.os_b == (long long)p.O[0] * p.O[1] * p.O[2] * p.Cout;
}

// returns RN_E_UNSUPPORTED (without setting an error) when no instantiation matches: the caller falls back
int rn_launch_conv_tiled(const RnConvProblem& p, hipStream_t st)
{
    static const bool off = getenv("RN_NO_TILED") != nullptr;
    if (off || !is_plain(p)) return RN_E_UNSUPPORTED;
    const bool k555s2 = p.K[0] == 5 && p.K[1] == 5 && p.K[2] == 5 && p.S[0] == 2 && p.S[1] == 2 && p.S[2] == 2;
    static const bool stem1 = getenv("RN_STEM_KERNEL_CIN1") != nullptr;
    if (stem1 && k555s2 && p.Cout == 8 && p.Cin == 1 && p.P[2] <= 2 && (p.O[2] - 1) * 2 - p.P[2] + 4 <= p.I[2] + 1) {
        static const int rpw = getenv("RN_STEM_RPW") ? atoi(getenv("RN_STEM_RPW")) : 16;
        const int rc = rpw == 16 ? launch_stem<1, 8, 16>(p, st) : rpw == 4 ? launch_stem<1, 8, 4>(p, st) : launch_stem<1, 8, 8>(p, st);
        if (rc != RN_E_UNSUPPORTED) return rc;
    }
    if (k555s2 && p.Cout == 8 && p.Cin == 1) return launch_tiled<5, 5, 5, 2, 2, 2, 1, 8, 4, 4, 16>(p, st);
    // the texture net's 5-channel stem: generic direct kernel 5.0 ms, LDS-tiled with one output per thread 5.5 ms (tile 4x4x16),
    // conv_rows_kernel (four outputs per thread, row windows read straight from global memory) 4.9 ms, conv_stem_kernel
    // (rows staged in LDS by whole-row loads, below) 2.45 ms on dense input -- less on the mostly empty resampled grid
    static const bool no_stem = getenv("RN_NO_STEM_KERNEL") != nullptr;
    static const bool no_stem_dc = getenv("RN_NO_STEM_DC") != nullptr;
    if (!no_stem && !no_stem_dc && k555s2 && p.Cout == 8 && p.Cin == 5) {       // depth-chunked: two workgroups per CU
        // measured (B=24, dense input): 16 rows x 32 depths, 2 outputs per lane 1.66-1.69 ms; 1 output per lane (three
        // workgroups per CU, half the filter reuse) 1.80; 8 rows x 32 depths 2.06; conv_stem_kernel (one workgroup per CU) 2.42
        const int rc = launch_stem_dc<5, 8, 16, 2>(p, st);
        if (rc != RN_E_UNSUPPORTED) return rc;
    }
    if (!no_stem && k555s2 && p.Cout == 8 && p.Cin == 5 && p.P[2] <= 2 && (p.O[2] - 1) * 2 - p.P[2] + 4 <= p.I[2] + 1) {             // row-staged stem (conv_stem_kernel)
        static const int rpw = getenv("RN_STEM_RPW") ? atoi(getenv("RN_STEM_RPW")) : 16;
        const int rc = rpw == 16 ? launch_stem<5, 8, 16>(p, st) : rpw == 4 ? launch_stem<5, 8, 4>(p, st) : launch_stem<5, 8, 8>(p, st);
        if (rc != RN_E_UNSUPPORTED) return rc;
    }
    if (k555s2 && p.Cout == 8 && p.Cin == 5) return launch_rows<5, 5, 5, 2, 2, 2, 5, 8, 4>(p, st);
    // (e_conv2 -- 3^3 stride (1,1,2), 8 -> 16 -- measured 0.98 ms tiled vs 0.79 ms with the generic direct kernel: its 8-float
    //  channel runs already coalesce, and 16 accumulators x 216 taps leave the tile's 4 waves per CU latency-bound.  Not routed here.
    //  Round 6 built two more forms, both parity-green and neither faster than the generic kernel's 0.57 ms in the step: an fp32-MFMA
    //  kernel with LDS-staged depth lines (0.53-0.55 ms, launch-bound) and a packed-FMA strip kernel, one wave per pair of output
    //  depth lines (0.59-0.63 ms at 254 registers): profiles/r06e_econv2_mfma.txt.)
    static const bool tiled_e2 = getenv("RN_TILED_ECONV2") != nullptr;
    if (tiled_e2 && p.K[0] == 3 && p.K[1] == 3 && p.K[2] == 3 && p.S[0] == 1 && p.S[1] == 1 && p.S[2] == 2 && p.Cin == 8 && p.Cout == 16)
        return launch_tiled<3, 3, 3, 1, 1, 2, 8, 16, 4, 4, 16>(p, st);
    const bool k44 = p.K[0] == 4 && p.K[1] == 4 && p.K[2] == 1 && p.S[0] == 1 && p.S[1] == 1 && p.S[2] == 1 && p.I[2] == 1;
    // e_conv11: the strip kernel (RN_NO_TAIL_KERNEL=1: the LDS-tiled one); it needs 16-byte aligned pixels and the output grid = the input grid
    static const bool no_tail = getenv("RN_NO_TAIL_KERNEL") != nullptr;
    const bool tail_ok = !no_tail && p.O[0] == p.I[0] && p.O[1] == p.I[1] && p.P[0] >= 0 && p.P[0] <= 3 && p.P[1] >= 0 && p.P[1] <= 3 &&
                         (reinterpret_cast<uintptr_t>(p.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.w) & 15) == 0;
    if (k44 && p.Cin == 16 && p.Cout == 1) return tail_ok ? launch_tail<1, 4>(p, st) : launch_tiled<4, 4, 1, 1, 1, 1, 16, 1, 16, 16, 1>(p, st);
    if (k44 && p.Cin == 16 && p.Cout == 3) return tail_ok ? launch_tail<3, 4>(p, st) : launch_tiled<4, 4, 1, 1, 1, 1, 16, 3, 16, 16, 1>(p, st);
    if (k44 && p.Cin == 32 && p.Cout == 1 && tail_ok) return launch_tail<1, 8>(p, st);            // (the 128^3 -> 1024^2 config's tail; else: the generic direct kernel)
    if (k44 && p.Cin == 32 && p.Cout == 3 && tail_ok) return launch_tail<3, 8>(p, st);
    return RN_E_UNSUPPORTED;
}
