// Filter gradient of the wide stride-1 3x3 / 4x4 2-D convs through Winograd F(4x4,3x3) / F(4x4,4x4) with the multiply stage on the
// bf16 matrix pipe at fp32 accuracy -- tf.nn.conv2d_backprop_filter of the res_block_2d / *_skip convs (tools/layer_util.py:101-104,
// RenderNet_Shader.py:71-84, :91-99) and of e_conv5 / e_conv6 (:86-88, :101-103) in the training step; the split counterpart of
// conv_wino43_wgrad.hip:
//
//     dg = G^T [ sum_tiles (B^T d B) .* (A dY A^T) ] G
//
// The sum over the tiles is the GEMM dU[xi] (Cin x Cout) = V[xi]^T (Cin x T) . dM[xi] (T x Cout): the REDUCTION runs over the
// tiles, so the operands of conv_wino_bf3.hip's GEMM kernel -- rows of [3 pieces][16 consecutive k] per output row -- must hold 16
// consecutive TILES of one channel.  Two transform kernels write exactly that, and the forward GEMM kernel runs unchanged with
// the roles (rows, K, columns) = (input channels, tiles, output channels):
//   1. wino_input_bf3t_kernel   x  [B,H,W,Cin]  -> Vt  [xi*KS + ks][Tk/16][Cin][3][16]            (V = B^T d B, split, tile-major rows)
//   2. wino_dout_bf3t_kernel    dz [B,H,W,Cout] -> dMt [xi*KS + ks][Cout/256][Tk/16][256][3][16]  (dM = A dY A^T, split)
//   3. wino_gemm_bf3_kernel     (conv_wino_bf3.hip)  -> dUp [xi*KS + ks][Cin][Cout] fp32
//   4. wino_dfilter_bf3_kernel  dw [R,R,Cin,Cout] += G^T (sum_ks dUp) G
// KS = K splits: the tiles are cut into KS runs of Tk (a multiple of 32, zero tiles behind T) so that planes x blocks fill whole rounds
// of the persistent GEMM grid (res2 at crop 64: 36 x 16 = 576 blocks = 2.25 rounds -> KS = 2: 4.5, the half round as half items).
// Arithmetic: V and dM are the fp32 values of the exact path (same formulas), each the exact sum of three bf16 pieces; six piece
// products per product, fp32 accumulation (conv_wino_bf3.hip explains the error class).
#include "rn_common.h"
#include "wino_mats.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TB_ROW = 96;                         // a GEMM operand row: 3 planes x 16 bf16
constexpr int TB_TILES = 16, TB_CH = 32;           // a transform workgroup: one K step of tiles x 32 channels; thread = (tile pair, channel)
constexpr int TB_PITCH = 112;                      // LDS row pitch (96 + 16: rows stay 16-byte aligned, 2-way write conflicts at worst)
constexpr int TB_SEG = TB_CH * TB_PITCH;           // one xi of a workgroup: 32 rows
constexpr int TB_PANEL = 256 * TB_ROW;             // one K step of one 256-row block of the U-side operand: 24 KiB

__device__ __forceinline__ unsigned xcd_contiguous(unsigned blk, unsigned nblk8) { return (blk & 7u) * (nblk8 >> 3) + (blk >> 3); }

// two fp32 values (the same channel of two neighbouring tiles) -> three words of two bf16 pieces each
__device__ __forceinline__ void split3_pair(float a, float b, unsigned (&w)[3])
{
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    f32x2_t v = {a, b};
#pragma unroll
    for (int q = 0; q < 3; ++q) {                          // one packed conversion per piece pair; its halves widened again feed a packed subtraction
        w[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
        if (q < 2) {
            f32x2_t h;
            h[0] = __builtin_bit_cast(float, w[q] << 16);
            h[1] = __builtin_bit_cast(float, w[q] & 0xffff0000u);
            v -= h;
        }
    }
}

// Writes the A x A values of this thread's two tiles (one channel) as GEMM rows.  val(i, j, e): value xi = (i, j) of tile e.
// Row i of the xi grid at a time goes through LDS ([j][channel row][3][16 tiles], chunk swapped in rows with bit 3 set: the
// GEMM's bank rule) and leaves as 16-byte stores: per xi the workgroup's 32 rows are 3 KiB CONTIGUOUS in both operand layouts.
template <int A, class F>
__device__ __forceinline__ void emit_rows(char* xch, char* gbase, size_t xi_stride, int tid, F&& val)
{
    const int c = tid & 31, tp = tid >> 5;                                   // channel row of the workgroup, tile pair (tiles 2 tp, 2 tp + 1)
    const unsigned sw = (unsigned)((c >> 3) & 1);
    const unsigned wofs = (unsigned)(c * TB_PITCH) + ((((unsigned)tp >> 2) ^ sw) << 4) + (unsigned)(tp & 3) * 4;
    constexpr int BUF = A * TB_SEG;
    // the way out: one xi = 32 rows x 6 chunks = 192 16-byte chunks = three store instructions of a whole wave; wave w takes the
    // columns j = w, w + 4 (its xi address is scalar arithmetic), the lane's row / chunk are fixed for the kernel
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned loff[3], goff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int rr = (tid & 63) + 64 * k, row = rr / 6, ch = rr - row * 6;
        loff[k] = (unsigned)(row * TB_PITCH + ch * 16);
        goff[k] = (unsigned)(rr * 16);
    }
#pragma unroll
    for (int i = 0; i < A; ++i) {
        char* buf = xch + (i & 1) * BUF;
#pragma unroll
        for (int j = 0; j < A; ++j) {
            unsigned w[3];
            split3_pair(val(i, j, 0), val(i, j, 1), w);
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<unsigned*>(buf + j * TB_SEG + wofs + q * 32) = w[q];
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < (A + 3) / 4; ++jj) {
            const int j = wv + 4 * jj;
            if (j >= A) break;
            char* gb = gbase + (size_t)(i * A + j) * xi_stride;
            const char* lb = buf + j * TB_SEG;
#pragma unroll
            for (int k = 0; k < 3; ++k) *reinterpret_cast<u32x4*>(gb + goff[k]) = *reinterpret_cast<const u32x4*>(lb + loff[k]);
        }
        // (two buffers: the next row's writes go to the other one; its read-out is separated from this row's by that row's barrier)
    }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// 1. V = B^T d B of the input tile (the forward's input transform, same formulas as wino_input_kernel), one channel of two
// neighbouring tiles per thread.  grid: (tile group of 16) x (channel block of 32); tiles >= T are zero tiles.
template <class S>
__global__ __launch_bounds__(256)
void wino_input_bf3t_kernel(const float* __restrict__ x, char* __restrict__ Vt, int H, int W, int C, int th, int tw, long long T,
                            int tk, int ks, unsigned ncb, unsigned nwg, unsigned nblk8, int pad_lo)
{
    constexpr int A = S::TA;
    __shared__ __attribute__((aligned(16))) char xch[2 * A * TB_SEG];
    const unsigned blk = xcd_contiguous(blockIdx.x, nblk8);
    if (blk >= nwg) return;                                    // (whole workgroups only: no barrier is skipped)
    const unsigned cb = blk % ncb;
    const long long g = blk / ncb;                             // tile group over all K splits: tiles 16 g .. 16 g + 15 of the padded run
    const int tid = threadIdx.x, c = (int)cb * TB_CH + (tid & 31), tp = tid >> 5;
    const int gps = tk / 16;                                   // groups per split
    const int split = (int)(g / gps), gi = (int)(g % gps);
    float v[2][A][A];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const long long t = (long long)split * tk + gi * 16 + 2 * tp + e;
        const bool live = t < T;
        const long long tc = live ? t : 0;
        const int tx = (int)(tc % tw), ty = (int)((tc / tw) % th);
        const long long b = tc / ((long long)tw * th);
        const int y0 = S::M * ty - pad_lo, x0 = S::M * tx - pad_lo;
        const float* xb = x + ((size_t)b * H * W) * C + c;
        float tt[A][A];                                        // (B^T d)[i][col]
#pragma unroll
        for (int col = 0; col < A; ++col) {
            float d[A];
            const int ix = x0 + col;
#pragma unroll
            for (int r = 0; r < A; ++r) {
                const int iy = y0 + r;
                const bool ok = live && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                d[r] = ok ? xb[((size_t)iy * W + ix) * C] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < A; ++i) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < A; ++k) {
                    const float cf = S::BT(i, k);
                    if (cf != 0.f) acc += cf * d[k];
                }
                tt[i][col] = acc;
            }
        }
#pragma unroll
        for (int i = 0; i < A; ++i)
#pragma unroll
            for (int j = 0; j < A; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < A; ++k) {
                    const float cf = S::BT(j, k);
                    if (cf != 0.f) acc += cf * tt[i][k];
                }
                v[e][i][j] = acc;
            }
    }
    // Vt [xi * KS + split][tk / 16][C][96]: this workgroup's rows = channels 32 cb .. 32 cb + 31 of group gi
    const size_t plane = (size_t)gps * C * TB_ROW;
    char* gbase = Vt + (size_t)split * plane + ((size_t)gi * C + (size_t)cb * TB_CH) * TB_ROW;
    emit_rows<A>(xch, gbase, (size_t)ks * plane, tid, [&](int i, int j, int e) { return v[e][i][j]; });
}

// 2. dM = A dY A^T: the 4x4 tile of the output gradient -> A x A (the adjoint of the output transform; formulas of wino_dout_kernel)
template <class S>
__global__ __launch_bounds__(256)
void wino_dout_bf3t_kernel(const float* __restrict__ dz, char* __restrict__ dMt, int H, int W, int C, int th, int tw, long long T,
                           int tk, int ks, unsigned ncb, unsigned nwg, unsigned nblk8)
{
    constexpr int A = S::TA;
    __shared__ __attribute__((aligned(16))) char xch[2 * A * TB_SEG];
    const unsigned blk = xcd_contiguous(blockIdx.x, nblk8);
    if (blk >= nwg) return;
    const unsigned cb = blk % ncb;
    const long long g = blk / ncb;
    const int tid = threadIdx.x, c = (int)cb * TB_CH + (tid & 31), tp = tid >> 5;
    const int gps = tk / 16;
    const int split = (int)(g / gps), gi = (int)(g % gps);
    float m[2][A][A];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const long long t = (long long)split * tk + gi * 16 + 2 * tp + e;
        const bool live = t < T;
        const long long tc = live ? t : 0;
        const int tx = (int)(tc % tw), ty = (int)((tc / tw) % th);
        const long long b = tc / ((long long)tw * th);
        const float* zb = dz + ((size_t)b * H * W) * C + c;
        float tt[A][4];                                        // (A dY)[i][q]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float d[4];
            const int ox = 4 * tx + q;
#pragma unroll
            for (int p_ = 0; p_ < 4; ++p_) {
                const int oy = 4 * ty + p_;
                d[p_] = (live && oy < H && ox < W) ? zb[((size_t)oy * W + ox) * C] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < A; ++i) {
                float acc = 0.f;
#pragma unroll
                for (int p_ = 0; p_ < 4; ++p_) {
                    const float cf = S::AT(p_, i);
                    if (cf != 0.f) acc += cf * d[p_];
                }
                tt[i][q] = acc;
            }
        }
#pragma unroll
        for (int i = 0; i < A; ++i)
#pragma unroll
            for (int j = 0; j < A; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float cf = S::AT(q, j);
                    if (cf != 0.f) acc += cf * tt[i][q];
                }
                m[e][i][j] = acc;
            }
    }
    // dMt [xi * KS + split][C / 256][tk / 16][256][96]: rows = channels (32 cb) % 256 .. + 31 of block (32 cb) / 256, group gi
    const size_t plane = (size_t)(C / 256) * gps * TB_PANEL;
    const int cblk = (int)(cb * TB_CH) / 256, crow = (int)(cb * TB_CH) % 256;
    char* gbase = dMt + (size_t)split * plane + ((size_t)cblk * gps + gi) * TB_PANEL + (size_t)crow * TB_ROW;
    emit_rows<A>(xch, gbase, (size_t)ks * plane, tid, [&](int i, int j, int e) { return m[e][i][j]; });
}

// 4. dw [R,R,Cin,Cout] += G^T (sum over the K splits of dUp[xi]) G.  thread = 4 consecutive output channels of one input channel.
template <class S>
__global__ __launch_bounds__(256)
void wino_dfilter_bf3_kernel(const float* __restrict__ dUp, float* __restrict__ dw, int Cin, int Cout, int ks)
{
    constexpr int A = S::TA, R = S::R;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t plane = (size_t)Cin * Cout;
    if (idx * 4 >= plane) return;
    const float* ub = dUp + idx * 4;
    f32x4 e[R][A];                                             // (G^T dU)[a][j]
#pragma unroll
    for (int a_ = 0; a_ < R; ++a_)
#pragma unroll
        for (int j = 0; j < A; ++j) e[a_][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j < A; ++j) {
            f32x4 u = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < ks; ++s) u += *reinterpret_cast<const f32x4*>(ub + ((size_t)(i * A + j) * ks + s) * plane);
#pragma unroll
            for (int a_ = 0; a_ < R; ++a_) {
                const float cf = (float)S::G(i, a_);
                if (cf != 0.f) e[a_][j] += cf * u;
            }
        }
#pragma unroll
    for (int a_ = 0; a_ < R; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < R; ++b_) {
            f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < A; ++j) {
                const float cf = (float)S::G(j, b_);
                if (cf != 0.f) w += cf * e[a_][j];
            }
            f32x4* d = reinterpret_cast<f32x4*>(dw + (size_t)(a_ * R + b_) * plane + idx * 4);
            *d += w;
        }
}

// ---------------------------------------------------------------------------------------------------------------------
namespace {
// K splits and tiles per split (a multiple of 16) for T tiles: the smallest KS <= 4 whose planes x blocks fill >= 85 % of whole
// rounds of 256 workgroups (a half round counts: the GEMM launcher runs a last partial round as half items), with >= 64 tiles each.
void wgrad_plan(int nxi, long long T, int Cin, int Cout, int& ks, int& tk)
{
    static const int forced = getenv("RN_WINO_BF3_WGRAD_SPLIT") ? atoi(getenv("RN_WINO_BF3_WGRAD_SPLIT")) : 0;
    const long long blocks = (long long)nxi * (Cin / 256) * (Cout / 256);
    int best = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= 4; ++s) {
        if (s > 1 && (T + s - 1) / s < 64) break;
        const long long items2 = blocks * s * 2;                                   // in half items
        const double eff = (double)items2 / (double)((items2 + 255) / 256 * 256);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = s; }
        if (eff >= 0.85) { best = s; break; }
    }
    ks = (forced >= 1 && forced <= 8) ? forced : best;
    const long long per = (T + ks - 1) / ks;
    tk = (int)((per + 31) / 32 * 32);                           // an even number of K steps, at least two (what the GEMM kernel's forward use guarantees it)
}
}  // namespace

bool rn_wino_bf3_wgrad_supported(int scheme, int Cin, int Cout)
{
    static const bool off = getenv("RN_NO_WINOGRAD_BF3_WGRAD") != nullptr;
    return !off && rn_wino43_wgrad_supported(scheme, Cin, Cout) && rn_wino_bf3_supported(scheme, 256, 256);
}

namespace {
size_t wgrad_ws_bytes(int nxi, long long T, int Cin, int Cout)
{
    int ks, tk;
    wgrad_plan(nxi, T, Cin, Cout, ks, tk);
    const size_t planes = (size_t)nxi * ks;
    return planes * tk * ((size_t)Cin + Cout) * 6 + planes * (size_t)Cin * Cout * 4 + 256;
}
}  // namespace

size_t rn_wino_bf3_wgrad_workspace_bytes(int scheme, int B, int H, int W, int Cin, int Cout)
{
    const long long per = (long long)((H + 3) / 4) * ((W + 3) / 4), T = (long long)B * per;
    const int nxi = rn_wino_scheme_nxi(scheme);
    const int cmax = Cin > Cout ? Cin : Cout;
    const long long lim = rn_wino43_plane_limit();
    if (per * cmax * 6 < lim && T * cmax * 6 >= lim) {          // batch chunks: the launcher plans each chunk on its own
        const int chunk = (int)((lim - 1) / (per * cmax * 6));
        const size_t a = wgrad_ws_bytes(nxi, (long long)chunk * per, Cin, Cout);
        const size_t b = B % chunk ? wgrad_ws_bytes(nxi, (long long)(B % chunk) * per, Cin, Cout) : 0;
        return a > b ? a : b;
    }
    return wgrad_ws_bytes(nxi, T, Cin, Cout);
}

// x [B,H,W,Cin], dz [B,H,W,Cout] -> dw [R,R,Cin,Cout] += conv2d_backprop_filter (RxR, stride 1, SAME); scheme F43: R = 3, F44: R = 4
int rn_launch_conv_wino_bf3_wgrad(int scheme, const float* x, const float* dz, float* dw, void* ws, int B, int H, int W, int Cin,
                                  int Cout, hipStream_t st)
{
    if (!rn_wino_bf3_wgrad_supported(scheme, Cin, Cout))
        return rn_set_error(RN_E_UNSUPPORTED, "conv_wino_bf3_wgrad: scheme=%d Cin=%d Cout=%d", scheme, Cin, Cout);
    const int nxi = rn_wino_scheme_nxi(scheme);
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const long long T = (long long)B * th * tw;
    if (T < 1) return rn_set_error(RN_E_INVALID, "conv_wino_bf3_wgrad: bad sizes");
    const int cmax = Cin > Cout ? Cin : Cout;
    const long long lim = rn_wino43_plane_limit();
    if ((long long)th * tw * cmax * 6 >= lim)
        return rn_set_error(RN_E_UNSUPPORTED, "conv_wino_bf3_wgrad: one image's transform plane exceeds the 2 GiB buffer window");
    if (T * cmax * 6 >= lim) {                                  // batch chunks (dw accumulates)
        const int chunk = (int)((lim - 1) / ((long long)th * tw * cmax * 6));
        for (int b0 = 0; b0 < B; b0 += chunk) {
            const int nb = B - b0 < chunk ? B - b0 : chunk;
            const int rc = rn_launch_conv_wino_bf3_wgrad(scheme, x + (size_t)b0 * H * W * Cin, dz + (size_t)b0 * H * W * Cout, dw, ws,
                                                         nb, H, W, Cin, Cout, st);
            if (rc != RN_OK) return rc;
        }
        return RN_OK;
    }
    int ks, tk;
    wgrad_plan(nxi, T, Cin, Cout, ks, tk);
    const size_t planes = (size_t)nxi * ks;
    char* Vt = static_cast<char*>(ws);
    char* dMt = Vt + planes * tk * (size_t)Cin * 6;
    float* dUp = reinterpret_cast<float*>(dMt + planes * tk * (size_t)Cout * 6);
    const unsigned groups = (unsigned)(ks * (tk / 16));
    const int pad_lo = 1;                                       // SAME padding of the 3x3 and of the 4x4 (1, 2) conv alike
    {
        const unsigned ncb = (unsigned)(Cin / TB_CH), nwg = groups * ncb, nblk8 = (nwg + 7) / 8 * 8;
        if (scheme == RN_WINO_F43)
            hipLaunchKernelGGL(wino_input_bf3t_kernel<WinoF43>, dim3(nblk8), dim3(256), 0, st, x, Vt, H, W, Cin, th, tw, T, tk, ks, ncb, nwg, nblk8, pad_lo);
        else
            hipLaunchKernelGGL(wino_input_bf3t_kernel<WinoF44>, dim3(nblk8), dim3(256), 0, st, x, Vt, H, W, Cin, th, tw, T, tk, ks, ncb, nwg, nblk8, pad_lo);
        const int rc = rn_check_launch("wino_input_bf3t");
        if (rc != RN_OK) return rc;
    }
    {
        const unsigned ncb = (unsigned)(Cout / TB_CH), nwg = groups * ncb, nblk8 = (nwg + 7) / 8 * 8;
        if (scheme == RN_WINO_F43)
            hipLaunchKernelGGL(wino_dout_bf3t_kernel<WinoF43>, dim3(nblk8), dim3(256), 0, st, dz, dMt, H, W, Cout, th, tw, T, tk, ks, ncb, nwg, nblk8);
        else
            hipLaunchKernelGGL(wino_dout_bf3t_kernel<WinoF44>, dim3(nblk8), dim3(256), 0, st, dz, dMt, H, W, Cout, th, tw, T, tk, ks, ncb, nwg, nblk8);
        const int rc = rn_check_launch("wino_dout_bf3t");
        if (rc != RN_OK) return rc;
    }
    // rows = input channels, K = the tiles of one split, columns = output channels
    int rc = rn_launch_gemm_bf3_planes((int)planes, 4, Vt, dMt, dUp, Cin, tk, Cout, st);
    if (rc != RN_OK) return rc;
    {
        const size_t n = (size_t)Cin * (Cout / 4);
        if (scheme == RN_WINO_F43) hipLaunchKernelGGL(wino_dfilter_bf3_kernel<WinoF43>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dUp, dw, Cin, Cout, ks);
        else hipLaunchKernelGGL(wino_dfilter_bf3_kernel<WinoF44>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dUp, dw, Cin, Cout, ks);
        return rn_check_launch("wino_dfilter_bf3");
    }
}
