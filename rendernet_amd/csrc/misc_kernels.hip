// Weight packing, fully-connected and the demo's Phong composite (all memory-bound helpers).
#include "rn_common.h"
#include <math.h>

// ---------------------------------------------------------------------------------------------
// Weight packing: TF filter layout -> [phase][ceil(K/4)][Npad][4], k = tap*Cin + c.
//   RN_PACK_CONV     : w_tf[k0,k1,k2,Cin,Cout]                  (tools/layer_util.py:162,243)
//   RN_PACK_CONVT_S1 : w_tf[k0,k1,k2,Cout,Cin], taps flipped    (tools/layer_util.py:201,284)
//   RN_PACK_CONVT_S2 : w_tf[4,4,(4),Cout,Cin] -> 2^nd phases of 2 taps per dim:
//                      phase 0 (even outputs) uses filter taps {3,1}, phase 1 (odd) taps {2,0}
// ---------------------------------------------------------------------------------------------
struct PackArgs {
    const float* w_tf; float* w_packed;
    int kind, ndim, K0, K1, K2, E0, E1, E2, Cin, Cout, Npad, Kq, nphase;
};

__global__ void pack_weights_kernel(const PackArgs a)
{
    const size_t per_phase = (size_t)a.Kq * a.Npad * 4;
    const size_t total = per_phase * a.nphase;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int phase = (int)(idx / per_phase);
        size_t rem = idx - (size_t)phase * per_phase;
        const int r = (int)(rem & 3); rem >>= 2;
        const int n = (int)(rem % a.Npad);
        const int kq = (int)(rem / a.Npad);
        const int k = kq * 4 + r;
        const int Keff = a.E0 * a.E1 * a.E2 * a.Cin;
        float v = 0.f;
        if (k < Keff && n < a.Cout) {
            const int c = k % a.Cin;
            int tap = k / a.Cin;
            const int t2 = tap % a.E2; tap /= a.E2;
            const int t1 = tap % a.E1; const int t0 = tap / a.E1;
            if (a.kind == RN_PACK_CONV) {
                v = a.w_tf[((((size_t)t0 * a.K1 + t1) * a.K2 + t2) * a.Cin + c) * a.Cout + n];
            } else {
                int s0, s1, s2;
                if (a.kind == RN_PACK_CONVT_S1) {
                    s0 = a.K0 - 1 - t0; s1 = a.K1 - 1 - t1; s2 = a.K2 - 1 - t2;
                } else {
                    // phase bits, most significant = dim 0
                    int p0, p1, p2;
                    if (a.ndim == 3) { p0 = (phase >> 2) & 1; p1 = (phase >> 1) & 1; p2 = phase & 1; }
                    else { p0 = (phase >> 1) & 1; p1 = phase & 1; p2 = 0; }
                    s0 = (p0 ? 2 : 3) - 2 * t0;
                    s1 = (p1 ? 2 : 3) - 2 * t1;
                    s2 = (a.ndim == 3) ? (p2 ? 2 : 3) - 2 * t2 : 0;
                }
                v = a.w_tf[((((size_t)s0 * a.K1 + s1) * a.K2 + s2) * a.Cout + n) * a.Cin + c];
            }
        }
        a.w_packed[idx] = v;
    }
}

// Winograd F(2x2,3x3) filter transform for csrc/conv_wino.hip: U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],
// packed [Cout/NB][KD*Cin/16][16 xi][4 kq][NB n][4 r] with channel c' = step*16 + kq*4 + r (one contiguous piece per
// (n-block, 16-channel step)); the n-block width NB is 32 when Cout % 32 == 0 and 16 otherwise -- the rule the kernel's
// launcher uses to pick one or two 16-channel MFMA tiles per wave.  RN_PACK_CONV_WINO reads w_tf[3,3,(3,)Cin,Cout]; RN_PACK_CONVT_S1_WINO reads
// w_tf[3,3,(3,)Cout,Cin] with the taps flipped (a stride-1 transposed conv = the input gradient of a 3x3(x3) conv).
// 3-D filters (KD = 3): the transform runs over the first two filter dims only, the depth tap t2 joins the channel:
// c' = t2*Cin + c -- the kernel walks the 3*Cin contiguous floats of three depth slices (see conv_wino.hip).
__global__ void pack_wino_kernel(const float* __restrict__ w_tf, float* __restrict__ u, int Cin, int Cout, int KD, int transposed)
{
    const size_t total = (size_t)16 * KD * Cin * Cout;
    const int nstep = KD * Cin / 16;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int NB = Cout % 32 == 0 ? 32 : 16;
        size_t rem = idx;
        const int r = (int)(rem & 3); rem >>= 2;
        const int n = (int)(rem % NB); rem /= NB;
        const int kq = (int)(rem & 3); rem >>= 2;
        const int xi = (int)(rem & 15); rem >>= 4;
        const int step = (int)(rem % nstep);
        const int nb = (int)(rem / nstep);
        const int ce = step * 16 + kq * 4 + r, co = nb * NB + n;
        const int t2 = ce / Cin, c = ce - t2 * Cin;
        const int i = xi >> 2, j = xi & 3;
        float g[3][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                g[p][q] = transposed ? w_tf[((size_t)(((2 - p) * 3 + (2 - q)) * KD + (KD - 1 - t2)) * Cout + co) * Cin + c]
                                     : w_tf[((size_t)((p * 3 + q) * KD + t2) * Cin + c) * Cout + co];
        float row[3];                               // (G g)[i][q]
#pragma unroll
        for (int q = 0; q < 3; ++q)
            row[q] = i == 0 ? g[0][q] : i == 1 ? 0.5f * ((g[0][q] + g[1][q]) + g[2][q])
                   : i == 2 ? 0.5f * ((g[0][q] - g[1][q]) + g[2][q]) : g[2][q];
        u[idx] = j == 0 ? row[0] : j == 1 ? 0.5f * ((row[0] + row[1]) + row[2])
               : j == 2 ? 0.5f * ((row[0] - row[1]) + row[2]) : row[2];
    }
}

// 4x4 filters for the same kernel (MODE 1): the filter is the sum of FOUR 2x2 sub-filters h_ab = g[2a:2a+2, 2b:2b+2], each
// transformed with F(2x2,2x2): U_ab = G2 h_ab G2^T, G2 = [[1,0],[1,1],[0,1]] (9 planes; 36 multiplies per 2x2 outputs and
// channel pair instead of 64).  Packed [Cout/NB][(Cin/16)*4][9 xi][4 kq][NB n][4 r]: K step s = cstep*4 + (2a+b), channel
// c = cstep*16 + kq*4 + r; NB = 64 when Cout % 64 == 0, else 32 (rn_wino_ntiles).  RN_PACK_CONV_WINO4 reads
// w_tf[4,4,Cin,Cout]; RN_PACK_CONVT_S1_WINO4 reads w_tf[4,4,Cout,Cin] with the taps flipped.
__global__ void pack_wino4_kernel(const float* __restrict__ w_tf, float* __restrict__ u, int Cin, int Cout, int NB, int transposed)
{
    const size_t total = (size_t)36 * Cin * Cout;
    const int nstep = Cin / 16 * 4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t rem = idx;
        const int r = (int)(rem & 3); rem >>= 2;
        const int n = (int)(rem % NB); rem /= NB;
        const int kq = (int)(rem & 3); rem >>= 2;
        const int xi = (int)(rem % 9); rem /= 9;
        const int step = (int)(rem % nstep);
        const int nb = (int)(rem / nstep);
        const int sub = step & 3, c = (step >> 2) * 16 + kq * 4 + r, co = nb * NB + n;
        const int a = sub >> 1, b = sub & 1, i = xi / 3, j = xi % 3;
        float h[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int t0 = 2 * a + p, t1 = 2 * b + q;
                h[p][q] = transposed ? w_tf[((size_t)((3 - t0) * 4 + (3 - t1)) * Cout + co) * Cin + c]
                                     : w_tf[((size_t)(t0 * 4 + t1) * Cin + c) * Cout + co];
            }
        float row[2];                               // (G2 h)[i][q]
#pragma unroll
        for (int q = 0; q < 2; ++q) row[q] = i == 0 ? h[0][q] : i == 1 ? h[0][q] + h[1][q] : h[1][q];
        u[idx] = j == 0 ? row[0] : j == 1 ? row[0] + row[1] : row[1];
    }
}

// Stride-2 4x4 transposed conv as four F(2x2,2x2) phase convs (conv_wino.hip MODE 2).  Per axis y[2m+pa] = x[m-1+pa] w[3-pa] +
// x[m+pa] w[1-pa] (TF SAME: crop 1), i.e. phase pa is the correlation of the input window starting at m-1+pa with the 2-tap
// filter h[p] = w[3 - pa - 2p].  Packed [4 phases][Cout/NB][Cin/16][9 xi][4 kq][NB n][4 r], U = G2 h G2^T with
// G2 = [[1,0],[1,1],[0,1]]; reads the TF conv_transpose filter w_tf[4,4,Cout,Cin]; channel c = cstep*16 + kq*4 + r.
__global__ void pack_wino_s2_kernel(const float* __restrict__ w_tf, float* __restrict__ u, int Cin, int Cout, int NB)
{
    const size_t total = (size_t)36 * Cin * Cout;
    const int nstep = Cin / 16, nblocks = Cout / NB;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t rem = idx;
        const int r = (int)(rem & 3); rem >>= 2;
        const int n = (int)(rem % NB); rem /= NB;
        const int kq = (int)(rem & 3); rem >>= 2;
        const int xi = (int)(rem % 9); rem /= 9;
        const int step = (int)(rem % nstep); rem /= nstep;
        const int nb = (int)(rem % nblocks);
        const int ph = (int)(rem / nblocks);
        const int c = step * 16 + kq * 4 + r, co = nb * NB + n;
        const int pa = ph >> 1, pb = ph & 1, i = xi / 3, j = xi % 3;
        float h[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                h[p][q] = w_tf[((size_t)((3 - pa - 2 * p) * 4 + (3 - pb - 2 * q)) * Cout + co) * Cin + c];
        float row[2];                               // (G2 h)[i][q]
#pragma unroll
        for (int q = 0; q < 2; ++q) row[q] = i == 0 ? h[0][q] : i == 1 ? h[0][q] + h[1][q] : h[1][q];
        u[idx] = j == 0 ? row[0] : j == 1 ? row[0] + row[1] : row[1];
    }
}

static bool is_wino43_kind(int kind)
{
    return kind == RN_PACK_CONV_WINO43 || kind == RN_PACK_CONVT_S1_WINO43 || kind == RN_PACK_CONV_WINO44 || kind == RN_PACK_CONVT_S1_WINO44 ||
           kind == RN_PACK_CONV_WINO63 || kind == RN_PACK_CONVT_S1_WINO63;
}
static int wino43_scheme(int kind)
{
    return (kind == RN_PACK_CONV_WINO44 || kind == RN_PACK_CONVT_S1_WINO44) ? RN_WINO_F44
         : (kind == RN_PACK_CONV_WINO63 || kind == RN_PACK_CONVT_S1_WINO63) ? RN_WINO_F63 : RN_WINO_F43;
}
static bool is_wino_kind(int kind) { return kind == RN_PACK_CONV_WINO || kind == RN_PACK_CONVT_S1_WINO; }
static bool is_wino4_kind(int kind) { return kind == RN_PACK_CONV_WINO4 || kind == RN_PACK_CONVT_S1_WINO4 || kind == RN_PACK_CONVT_S2_WINO; }

static int wino4_pack_check(int ndim, const int* kdims, int Cin, int Cout)
{
    if (!kdims || ndim != 2 || kdims[0] != 4 || kdims[1] != 4)
        return rn_set_error(RN_E_UNSUPPORTED, "pack: the 4x4 Winograd packs need a 2-D 4x4 filter");
    if (Cin < 16 || Cout < 16 || Cin % 16 != 0 || Cout % 16 != 0)
        return rn_set_error(RN_E_UNSUPPORTED, "pack: the 4x4 Winograd packs need Cin and Cout to be multiples of 16 (got %d, %d)", Cin, Cout);
    return RN_OK;
}

static int wino43_pack_check(int kind, int ndim, const int* kdims, int Cin, int Cout)
{
    const int r = rn_wino_scheme_r(wino43_scheme(kind));
    if (!kdims || ndim != 2 || kdims[0] != r || kdims[1] != r)
        return rn_set_error(RN_E_UNSUPPORTED, "pack: this Winograd pack needs a 2-D %dx%d filter", r, r);
    if (Cin < 32 || Cin % 32 != 0 || Cout < 256 || Cout % 256 != 0)
        return rn_set_error(RN_E_UNSUPPORTED, "pack: the three-launch Winograd packs need Cin %% 32 == 0 and Cout %% 256 == 0 (got %d, %d)", Cin, Cout);
    return RN_OK;
}

static int wino_pack_check(int ndim, const int* kdims, int Cin, int Cout)
{
    if (!kdims || (ndim != 2 && ndim != 3) || kdims[0] != 3 || kdims[1] != 3 || (ndim == 3 && kdims[2] != 3))
        return rn_set_error(RN_E_UNSUPPORTED, "pack: the Winograd packs need a 3x3 or 3x3x3 filter");
    if (Cin < 16 || Cout < 16 || Cin % 16 != 0 || Cout % 16 != 0)
        return rn_set_error(RN_E_UNSUPPORTED, "pack: the Winograd packs need Cin and Cout to be multiples of 16 (got %d, %d)", Cin, Cout);
    return RN_OK;
}

static int pack_geometry(int kind, int ndim, const int* kdims, int Cin, int Cout, PackArgs& a)
{
    if (!kdims || (ndim != 2 && ndim != 3) || Cin < 1 || Cout < 1)
        return rn_set_error(RN_E_INVALID, "pack: bad arguments");
    a.kind = kind; a.ndim = ndim;
    a.K0 = kdims[0]; a.K1 = kdims[1]; a.K2 = (ndim == 3) ? kdims[2] : 1;
    if (a.K0 < 1 || a.K1 < 1 || a.K2 < 1) return rn_set_error(RN_E_INVALID, "pack: bad kernel dims");
    a.E0 = a.K0; a.E1 = a.K1; a.E2 = a.K2; a.nphase = 1;
    if (kind == RN_PACK_CONVT_S2) {
        if (a.K0 != 4 || a.K1 != 4 || (ndim == 3 && a.K2 != 4))
            return rn_set_error(RN_E_UNSUPPORTED, "pack: stride-2 transposed conv needs k=4");
        a.E0 = 2; a.E1 = 2; a.E2 = (ndim == 3) ? 2 : 1; a.nphase = (ndim == 3) ? 8 : 4;
    } else if (kind != RN_PACK_CONV && kind != RN_PACK_CONVT_S1) {
        return rn_set_error(RN_E_INVALID, "pack: unknown kind %d", kind);
    }
    a.Cin = Cin; a.Cout = Cout; a.Npad = rn_round_up(Cout, 32);
    a.Kq = (a.E0 * a.E1 * a.E2 * Cin + 3) / 4;
    return RN_OK;
}

extern "C" size_t rn_packed_weight_floats(int kind, int ndim, const int* kdims, int Cin, int Cout)
{
    if (is_wino_kind(kind)) return wino_pack_check(ndim, kdims, Cin, Cout) == RN_OK ? (size_t)16 * (ndim == 3 ? 3 : 1) * Cin * Cout : 0;
    if (is_wino4_kind(kind)) return wino4_pack_check(ndim, kdims, Cin, Cout) == RN_OK ? (size_t)36 * Cin * Cout : 0;
    if (is_wino43_kind(kind)) return wino43_pack_check(kind, ndim, kdims, Cin, Cout) == RN_OK ? (size_t)rn_wino_scheme_nxi(wino43_scheme(kind)) * Cin * Cout : 0;
    PackArgs a;
    if (pack_geometry(kind, ndim, kdims, Cin, Cout, a) != RN_OK) return 0;
    return (size_t)a.nphase * a.Kq * a.Npad * 4;
}

extern "C" int rn_pack_weights(int kind, int ndim, const int* kdims, int Cin, int Cout,
                               const float* w_tf, float* w_packed, void* stream)
{
    if (is_wino_kind(kind)) {
        const int rcw = wino_pack_check(ndim, kdims, Cin, Cout);
        if (rcw != RN_OK) return rcw;
        if (!w_tf || !w_packed) return rn_set_error(RN_E_INVALID, "pack: null pointer");
        const int KD = ndim == 3 ? 3 : 1;
        const size_t tot = (size_t)16 * KD * Cin * Cout;
        const unsigned nbw = (unsigned)((tot + 255) / 256 > 65536 ? 65536 : (tot + 255) / 256);
        hipLaunchKernelGGL(pack_wino_kernel, dim3(nbw), dim3(256), 0, (hipStream_t)stream, w_tf, w_packed, Cin, Cout, KD,
                           kind == RN_PACK_CONVT_S1_WINO ? 1 : 0);
        return rn_check_launch("pack_wino");
    }
    if (is_wino4_kind(kind)) {
        const int rcw = wino4_pack_check(ndim, kdims, Cin, Cout);
        if (rcw != RN_OK) return rcw;
        if (!w_tf || !w_packed) return rn_set_error(RN_E_INVALID, "pack: null pointer");
        const size_t tot = (size_t)36 * Cin * Cout;
        const unsigned nbw = (unsigned)((tot + 255) / 256 > 65536 ? 65536 : (tot + 255) / 256);
        if (kind == RN_PACK_CONVT_S2_WINO)
            hipLaunchKernelGGL(pack_wino_s2_kernel, dim3(nbw), dim3(256), 0, (hipStream_t)stream, w_tf, w_packed, Cin, Cout,
                               16 * rn_wino_ntiles(2, Cout));
        else
            hipLaunchKernelGGL(pack_wino4_kernel, dim3(nbw), dim3(256), 0, (hipStream_t)stream, w_tf, w_packed, Cin, Cout,
                               16 * rn_wino_ntiles(1, Cout), kind == RN_PACK_CONVT_S1_WINO4 ? 1 : 0);
        return rn_check_launch("pack_wino4");
    }
    if (is_wino43_kind(kind)) {
        const int rcw = wino43_pack_check(kind, ndim, kdims, Cin, Cout);
        if (rcw != RN_OK) return rcw;
        if (!w_tf || !w_packed) return rn_set_error(RN_E_INVALID, "pack: null pointer");
        return rn_launch_wino_pack(wino43_scheme(kind), w_tf, w_packed, Cin, Cout,
                                   (kind == RN_PACK_CONVT_S1_WINO43 || kind == RN_PACK_CONVT_S1_WINO44 || kind == RN_PACK_CONVT_S1_WINO63) ? 1 : 0,
                                   (hipStream_t)stream);
    }
    PackArgs a;
    int rc = pack_geometry(kind, ndim, kdims, Cin, Cout, a);
    if (rc != RN_OK) return rc;
    if (!w_tf || !w_packed) return rn_set_error(RN_E_INVALID, "pack: null pointer");
    a.w_tf = w_tf; a.w_packed = w_packed;
    const size_t total = (size_t)a.nphase * a.Kq * a.Npad * 4;
    const unsigned nb = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
    return rn_check_launch("pack_weights");
}

// ---------------------------------------------------------------------------------------------
// fully_connected (tools/layer_util.py:311-343): y = act(x @ w + bias).  HBM-bound on w
// ([in,out], 104 MB for the texture decoder's 199 -> 131072): one thread per output column,
// columns coalesced across lanes, the <= 8 batch rows of a chunk accumulate in registers,
// x is read with wave-uniform addresses.
// ---------------------------------------------------------------------------------------------
template <int BC>
__global__ __launch_bounds__(256)
void fc_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
               const float* __restrict__ alpha, float* __restrict__ y, float* __restrict__ preact,
               int B, int in_f, int out_f, int act)
{
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int b0 = blockIdx.y * BC;
    if (col >= out_f) return;
    float acc[BC];
#pragma unroll
    for (int i = 0; i < BC; ++i) acc[i] = 0.f;
    for (int k = 0; k < in_f; ++k) {
        const float wv = w[(size_t)k * out_f + col];
#pragma unroll
        for (int i = 0; i < BC; ++i)
            if (b0 + i < B) acc[i] = fmaf(x[(size_t)(b0 + i) * in_f + k], wv, acc[i]);
    }
    const float bv = bias ? bias[col] : 0.f;
    const float av = alpha ? alpha[col] : 0.f;
#pragma unroll
    for (int i = 0; i < BC; ++i) {
        if (b0 + i < B) {
            float v = acc[i] + bv;
            if (preact) preact[(size_t)(b0 + i) * out_f + col] = v;
            if (act & RN_ACT_PRELU) v = fmaxf(v, 0.f) + av * fminf(v, 0.f);
            if (act & RN_ACT_ELU) v = v > 0.f ? v : expf(v) - 1.f;
            if (act & RN_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
            y[(size_t)(b0 + i) * out_f + col] = v;
        }
    }
}

extern "C" int rn_fully_connected_fwd(const float* x, const float* w, const float* bias, const float* alpha,
                                      float* y, int B, int in_features, int out_features, int act, void* stream)
{
    if (!x || !w || !y || B < 1 || in_features < 1 || out_features < 1)
        return rn_set_error(RN_E_INVALID, "rn_fully_connected_fwd: bad argument");
    dim3 grid((out_features + 255) / 256, (B + 7) / 8);
    hipLaunchKernelGGL(fc_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, alpha, y, (float*)nullptr, B,
                       in_features, out_features, act);
    return rn_check_launch("fully_connected");
}

extern "C" int rn_fully_connected_fwd_train(const float* x, const float* w, const float* bias, const float* alpha,
                                            float* y, float* preact, int B, int in_features, int out_features, int act,
                                            void* stream)
{
    if (!x || !w || !y || B < 1 || in_features < 1 || out_features < 1)
        return rn_set_error(RN_E_INVALID, "rn_fully_connected_fwd_train: bad argument");
    dim3 grid((out_features + 255) / 256, (B + 7) / 8);
    hipLaunchKernelGGL(fc_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, alpha, y, preact, B,
                       in_features, out_features, act);
    return rn_check_launch("fully_connected_train");
}

// Backward of y = x @ w (tools/layer_util.py:311-343; bias / PReLU go through rn_epilogue_bwd):
//   dw[k][col] += sum_b x[b][k] * dz[b][col]   one thread per column, the batch kept in registers BC rows at a time
//   dx[b][k]    = sum_col dz[b][col] * w[k][col]  one workgroup per (k, 8 batch rows), reduced over the columns
template <int BC>
__global__ __launch_bounds__(256)
void fc_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ dw,
                     int B, int in_f, int out_f)
{
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= out_f) return;
    for (int b0 = 0; b0 < B; b0 += BC) {
        float g[BC];
#pragma unroll
        for (int i = 0; i < BC; ++i) g[i] = (b0 + i < B) ? dz[(size_t)(b0 + i) * out_f + col] : 0.f;
        for (int k = 0; k < in_f; ++k) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < BC; ++i)
                if (b0 + i < B) s = fmaf(x[(size_t)(b0 + i) * in_f + k], g[i], s);
            dw[(size_t)k * out_f + col] += s;             // this thread owns column `col` of dw
        }
    }
}

template <int BC>
__global__ __launch_bounds__(256)
void fc_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ dx,
                     int B, int in_f, int out_f)
{
    __shared__ float red[4][BC];
    const int k = blockIdx.x, b0 = blockIdx.y * BC;
    float acc[BC];
#pragma unroll
    for (int i = 0; i < BC; ++i) acc[i] = 0.f;
    for (int col = threadIdx.x; col < out_f; col += 256) {
        const float wv = w[(size_t)k * out_f + col];
#pragma unroll
        for (int i = 0; i < BC; ++i)
            if (b0 + i < B) acc[i] = fmaf(dz[(size_t)(b0 + i) * out_f + col], wv, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < BC; ++i) {
        float v = acc[i];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < BC && b0 + threadIdx.x < B)
        dx[(size_t)(b0 + threadIdx.x) * in_f + k] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

extern "C" int rn_fully_connected_bwd(const float* x, const float* w, const float* dz, float* dx, float* dw,
                                      int B, int in_features, int out_features, void* stream)
{
    if (!dz || B < 1 || in_features < 1 || out_features < 1 || (!dx && !dw))
        return rn_set_error(RN_E_INVALID, "rn_fully_connected_bwd: bad argument");
    if ((dw && !x) || (dx && !w)) return rn_set_error(RN_E_INVALID, "rn_fully_connected_bwd: dw needs x, dx needs w");
    hipStream_t st = (hipStream_t)stream;
    if (dw) hipLaunchKernelGGL(fc_wgrad_kernel<8>, dim3((out_features + 255) / 256), dim3(256), 0, st, x, dz, dw, B, in_features, out_features);
    if (dx) hipLaunchKernelGGL(fc_dgrad_kernel<8>, dim3(in_features, (B + 7) / 8), dim3(256), 0, st, dz, w, dx, B, in_features, out_features);
    return rn_check_launch("fully_connected_bwd");
}

// ---------------------------------------------------------------------------------------------
// Stand-alone PReLU (tools/layer_util.py:27-45); the hot path uses the fused conv epilogues.
// ---------------------------------------------------------------------------------------------
__global__ void prelu_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                             float* __restrict__ y, size_t n, int C)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = fmaxf(v, 0.f) + alpha[i % C] * fminf(v, 0.f);
    }
}

extern "C" int rn_prelu_fwd(const float* x, const float* alpha, float* y, size_t n, int C, void* stream)
{
    if (!x || !alpha || !y || C < 1) return rn_set_error(RN_E_INVALID, "rn_prelu_fwd: bad argument");
    if (n == 0) return RN_OK;
    const size_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(prelu_kernel, dim3((unsigned)(nb > 16384 ? 16384 : nb)), dim3(256), 0, (hipStream_t)stream,
                       x, alpha, y, n, C);
    return rn_check_launch("prelu");
}

// ---------------------------------------------------------------------------------------------
// Phong composite.  NumPy flavour of the demo (tools/Phong_shading.py:202-228, :162-200, :138-160) and the
// differentiable TF flavour of the inverse-rendering graph (:46-111, :23-44); they differ only in the mask:
//   n = (img-0.5)/|img-0.5| ; d = clip(k_d * max(n . l^, 0) * col, 0, 1)
//   mask = sigmoid(255*s - thr):  RN_PHONG_NP_BLACK  s = |img|,       thr 150   (np_mask,        :138-148)
//                                 RN_PHONG_NP_WHITE  s = |1 - img|,   thr 80    (np_mask_white,  :150-160)
//                                 RN_PHONG_TF_BLACK  s = |img|,       thr 80    (tf_mask,        :23-32)
//                                 RN_PHONG_TF_WHITE  s = sqrt(3)-|img|, thr 80  (tf_mask_white,  :34-44)
//                                 RN_PHONG_NO_MASK   out = clip(ambient + d)
//   shading = clip(mask*(ambient + d) + (1-mask), 0, 1) ;  out = shading * albedo  when albedo is given
//   (compos_pred = img_pred * shading, Reconstruct_RenderNet_Face.py:377-378).
// Backward: TF's gradients of the same graph (clip_by_value / maximum pass the gradient inside their range,
// tf.norm's gradient is v/|v|), d/d img and d/d light_dir (reduced per batch item), d/d albedo.
// ---------------------------------------------------------------------------------------------
struct PhongPix {
    float nx, ny, nz, nn;        // unit normal, |img - 0.5|
    float lx, ly, lz, ln;        // unit light, |light|
    float dot, m, s;             // n.l, mask, mask argument s
    float dc[3], comp[3];        // k_d*d*col before the clip; composite before the clip
};

__device__ __forceinline__ void phong_eval(float r, float g, float bl, const float* ld, const float* lc,
                                           float ambient, float k_diffuse, int mode, PhongPix& p, float* sh)
{
    p.lx = ld[0]; p.ly = ld[1]; p.lz = ld[2];
    p.ln = sqrtf(p.lx * p.lx + p.ly * p.ly + p.lz * p.lz);
    p.lx /= p.ln; p.ly /= p.ln; p.lz /= p.ln;
    const float vx = r - 0.5f, vy = g - 0.5f, vz = bl - 0.5f;
    p.nn = sqrtf(vx * vx + vy * vy + vz * vz);
    p.nx = vx / p.nn; p.ny = vy / p.nn; p.nz = vz / p.nn;
    p.dot = p.nx * p.lx + p.ny * p.ly + p.nz * p.lz;
    const float d = fmaxf(p.dot, 0.f);
    float thr = 80.f;
    if (mode == RN_PHONG_NP_BLACK) { p.s = sqrtf(r * r + g * g + bl * bl); thr = 150.f; }
    else if (mode == RN_PHONG_TF_BLACK) p.s = sqrtf(r * r + g * g + bl * bl);
    else if (mode == RN_PHONG_NP_WHITE) p.s = sqrtf((1.f - r) * (1.f - r) + (1.f - g) * (1.f - g) + (1.f - bl) * (1.f - bl));
    else p.s = 1.7320508075688772f - sqrtf(r * r + g * g + bl * bl);
    p.m = (mode == RN_PHONG_NO_MASK) ? 1.f : 1.f / (1.f + expf(-(255.f * p.s - thr)));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        p.dc[c] = k_diffuse * (d * lc[c]);
        const float D = fminf(fmaxf(p.dc[c], 0.f), 1.f);
        p.comp[c] = (mode == RN_PHONG_NO_MASK) ? ambient + D : p.m * (ambient + D) + (1.f - p.m);
        sh[c] = fminf(fmaxf(p.comp[c], 0.f), 1.f);
    }
}

__global__ void phong_kernel(const float* __restrict__ img, const float* __restrict__ light_dir,
                             const float* __restrict__ light_col, const float* __restrict__ albedo,
                             float ambient, float k_diffuse, float* __restrict__ out, int B, int HW, int mode)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long long)B * HW) return;
    const int b = (int)(p / HW);
    PhongPix px; float sh[3];
    phong_eval(img[p * 3 + 0], img[p * 3 + 1], img[p * 3 + 2], light_dir + b * 3, light_col + b * 3,
               ambient, k_diffuse, mode, px, sh);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[p * 3 + c] = albedo ? sh[c] * albedo[p * 3 + c] : sh[c];
}

// grid = (blocks per item, B): a block's pixels share the light, so d/d light is one block reduction + 3 atomics
__global__ __launch_bounds__(256)
void phong_bwd_kernel(const float* __restrict__ img, const float* __restrict__ light_dir,
                      const float* __restrict__ light_col, const float* __restrict__ albedo,
                      float ambient, float k_diffuse, const float* __restrict__ dout,
                      float* __restrict__ dimg, float* __restrict__ dlight, float* __restrict__ dalbedo,
                      int HW, int mode)
{
    __shared__ float red[3][4];
    const int b = blockIdx.y;
    float gl[3] = {0.f, 0.f, 0.f};                      // d loss / d unit light, this thread's pixels
    PhongPix px;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < HW; q += gridDim.x * 256) {
        const long long p = (long long)b * HW + q;
        const float r = img[p * 3 + 0], g = img[p * 3 + 1], bl = img[p * 3 + 2];
        float sh[3];
        phong_eval(r, g, bl, light_dir + b * 3, light_col + b * 3, ambient, k_diffuse, mode, px, sh);
        float dd = 0.f, dm = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float go = dout[p * 3 + c];
            if (albedo) {
                if (dalbedo) dalbedo[p * 3 + c] = go * sh[c];
                go *= albedo[p * 3 + c];
            }
            const float gc = (px.comp[c] >= 0.f && px.comp[c] <= 1.f) ? go : 0.f;
            const float D = fminf(fmaxf(px.dc[c], 0.f), 1.f);
            if (mode != RN_PHONG_NO_MASK) dm += gc * (ambient + D - 1.f);
            const float gD = (mode == RN_PHONG_NO_MASK) ? gc : gc * px.m;
            if (px.dc[c] >= 0.f && px.dc[c] <= 1.f) dd += gD * k_diffuse * light_col[b * 3 + c];
        }
        const float gdot = px.dot > 0.f ? dd : 0.f;
        // n = v/|v|:  dv = (dn - n (n.dn)) / |v|,  dn = gdot * l^
        const float ndl = px.dot;                       // n . l^
        float gi[3] = {gdot * (px.lx - px.nx * ndl) / px.nn, gdot * (px.ly - px.ny * ndl) / px.nn,
                       gdot * (px.lz - px.nz * ndl) / px.nn};
        gl[0] += gdot * px.nx; gl[1] += gdot * px.ny; gl[2] += gdot * px.nz;
        if (mode != RN_PHONG_NO_MASK) {
            const float gs = 255.f * dm * px.m * (1.f - px.m);
            if (mode == RN_PHONG_NP_WHITE) {
                const float f = -gs / px.s;
                gi[0] += f * (1.f - r); gi[1] += f * (1.f - g); gi[2] += f * (1.f - bl);
            } else {
                const float nrm = sqrtf(r * r + g * g + bl * bl);
                const float f = (mode == RN_PHONG_TF_WHITE ? -gs : gs) / nrm;
                gi[0] += f * r; gi[1] += f * g; gi[2] += f * bl;
            }
        }
        if (dimg) { dimg[p * 3 + 0] = gi[0]; dimg[p * 3 + 1] = gi[1]; dimg[p * 3 + 2] = gi[2]; }
    }
    if (!dlight) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = gl[c];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
        if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // l^ = l/|l|:  dl = (dl^ - l^ (l^ . dl^)) / |l|
        float lx = light_dir[b * 3 + 0], ly = light_dir[b * 3 + 1], lz = light_dir[b * 3 + 2];
        const float ln = sqrtf(lx * lx + ly * ly + lz * lz);
        lx /= ln; ly /= ln; lz /= ln;
        const float g0 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const float g1 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const float g2 = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        const float ldg = lx * g0 + ly * g1 + lz * g2;
        unsafeAtomicAdd(dlight + b * 3 + 0, (g0 - lx * ldg) / ln);
        unsafeAtomicAdd(dlight + b * 3 + 1, (g1 - ly * ldg) / ln);
        unsafeAtomicAdd(dlight + b * 3 + 2, (g2 - lz * ldg) / ln);
    }
}

static int phong_args_ok(const float* normals, const float* light_dir, const float* light_col, int B, int H, int W, int mode)
{
    return normals && light_dir && light_col && B >= 1 && H >= 1 && W >= 1 && mode >= RN_PHONG_NP_BLACK && mode <= RN_PHONG_NO_MASK;
}

extern "C" int rn_phong_composite_ex_fwd(const float* normals, const float* light_dir, const float* light_col,
                                         const float* albedo, float ambient, float k_diffuse, float* out,
                                         int B, int H, int W, int mask_mode, void* stream)
{
    if (!phong_args_ok(normals, light_dir, light_col, B, H, W, mask_mode) || !out)
        return rn_set_error(RN_E_INVALID, "rn_phong_composite_ex_fwd: bad argument");
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(phong_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       normals, light_dir, light_col, albedo, ambient, k_diffuse, out, B, H * W, mask_mode);
    return rn_check_launch("phong_composite");
}

extern "C" int rn_phong_composite_fwd(const float* normals, const float* light_dir, const float* light_col,
                                      float ambient, float k_diffuse, float* out, int B, int H, int W, void* stream)
{
    return rn_phong_composite_ex_fwd(normals, light_dir, light_col, nullptr, ambient, k_diffuse, out, B, H, W,
                                     RN_PHONG_NP_BLACK, stream);
}

extern "C" int rn_phong_composite_bwd(const float* normals, const float* light_dir, const float* light_col,
                                      const float* albedo, float ambient, float k_diffuse, const float* dout,
                                      float* dnormals, float* dlight_dir, float* dalbedo,
                                      int B, int H, int W, int mask_mode, void* stream)
{
    if (!phong_args_ok(normals, light_dir, light_col, B, H, W, mask_mode) || !dout)
        return rn_set_error(RN_E_INVALID, "rn_phong_composite_bwd: bad argument");
    if (dalbedo && !albedo) return rn_set_error(RN_E_INVALID, "rn_phong_composite_bwd: dalbedo without albedo");
    const int HW = H * W;
    int bx = (HW + 256 * 8 - 1) / (256 * 8);           // 8 pixels per thread: few atomics per item
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(phong_bwd_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       normals, light_dir, light_col, albedo, ambient, k_diffuse, dout, dnormals, dlight_dir, dalbedo,
                       HW, mask_mode);
    return rn_check_launch("phong_composite_bwd");
}
