// C ABI of librendernet_hip.so: argument checking, TF "SAME" geometry, lowering of every conv
// flavour to RnConvProblem, and the dispatch between the MFMA implicit-GEMM kernel and the
// direct VALU kernel.  See include/rendernet_hip.h for the contract.
#include "rn_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

int rn_set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int rn_check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rn_set_error(RN_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return RN_OK;
}

int rn_ensure_dynamic_lds(const void* kernel, size_t bytes)
{
    struct Slot { const void* k; int dev; size_t bytes; };
    static thread_local Slot cache[64];
    static thread_local int used = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (int i = 0; i < used; ++i)
        if (cache[i].k == kernel && cache[i].dev == dev && cache[i].bytes >= bytes) return RN_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return rn_set_error(RN_E_LAUNCH, "hipFuncSetAttribute(%zu B of LDS): %s", bytes, hipGetErrorString(e)); }
    if (used < 64) cache[used++] = Slot{kernel, dev, bytes};
    return RN_OK;
}

extern "C" int rn_version(void) { return RN_VERSION; }
extern "C" const char* rn_last_error(void) { return g_err; }

// TF SAME: out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0); pad_before = total/2
static inline void same_geom(int in, int k, int s, int& out, int& pad_lo)
{
    out = (in + s - 1) / s;
    int total = (out - 1) * s + k - in;
    if (total < 0) total = 0;
    pad_lo = total / 2;
}

static int dispatch(const RnConvProblem& p, hipStream_t st)
{
    static const bool no_drun = getenv("RN_NO_DRUN") != nullptr;
    if (!no_drun && rn_drun_supported(p)) return rn_launch_conv3d_drun(p, st);
    if (rn_igemm_supported(p) && p.Cout >= 8) return rn_launch_conv_igemm(p, st);
    const int rc = rn_launch_conv_tiled(p, st);          // LDS-tiled kernels of the stem / tail shapes
    if (rc != RN_E_UNSUPPORTED) return rc;
    if (rn_igemm_supported(p)) return rn_launch_conv_igemm(p, st);
    return rn_launch_conv_direct(p, st);
}

static int conv_fwd_nd(const float* x, const float* w, const float* bias, const float* alpha,
                       const float* residual, float* y, int B, const int* I, int Cin, int Cout,
                       const int* k, const int* s, int act, hipStream_t st, const char* who, float* preact = nullptr)
{
    if (!x || !w || !y) return rn_set_error(RN_E_INVALID, "%s: null pointer", who);
    if (B < 1 || Cin < 1 || Cout < 1) return rn_set_error(RN_E_INVALID, "%s: bad sizes", who);
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "%s: PReLU needs alpha", who);
    RnConvProblem p;
    p.x = x; p.w = w; p.bias = bias; p.alpha = alpha; p.residual = residual; p.y = y; p.preact = preact;
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.Npad = rn_round_up(Cout, 32);
    for (int d = 0; d < 3; ++d) {
        if (I[d] < 1 || k[d] < 1 || s[d] < 1) return rn_set_error(RN_E_INVALID, "%s: bad geometry", who);
        p.I[d] = I[d]; p.K[d] = k[d]; p.S[d] = s[d];
        same_geom(I[d], k[d], s[d], p.O[d], p.P[d]);
    }
    p.os[2] = Cout;
    p.os[1] = (long long)p.O[2] * Cout;
    p.os[0] = (long long)p.O[1] * p.os[1];
    p.os_b = (long long)p.O[0] * p.os[0];
    p.out_off = 0;
    p.act = act;
    return dispatch(p, st);
}

extern "C" int rn_conv3d_fwd(const float* x, const float* w_packed, const float* bias, const float* alpha,
                             const float* residual, float* y, int B, int H, int W, int D, int Cin, int Cout,
                             const int* ksize, const int* stride, int act, void* stream)
{
    if (!ksize || !stride) return rn_set_error(RN_E_INVALID, "rn_conv3d_fwd: null ksize/stride");
    const int I[3] = {H, W, D};
    return conv_fwd_nd(x, w_packed, bias, alpha, residual, y, B, I, Cin, Cout, ksize, stride, act,
                       (hipStream_t)stream, "rn_conv3d_fwd");
}

extern "C" int rn_conv2d_fwd(const float* x, const float* w_packed, const float* bias, const float* alpha,
                             const float* residual, float* y, int B, int H, int W, int Cin, int Cout,
                             const int* ksize, const int* stride, int act, void* stream)
{
    if (!ksize || !stride) return rn_set_error(RN_E_INVALID, "rn_conv2d_fwd: null ksize/stride");
    const int I[3] = {H, W, 1}, k[3] = {ksize[0], ksize[1], 1}, s[3] = {stride[0], stride[1], 1};
    return conv_fwd_nd(x, w_packed, bias, alpha, residual, y, B, I, Cin, Cout, k, s, act,
                       (hipStream_t)stream, "rn_conv2d_fwd");
}

extern "C" int rn_conv2d_wino_supported(int Cin, int Cout) { return rn_wino_supported(Cin, Cout) ? 1 : 0; }

extern "C" int rn_conv2d_wino_fwd(const float* x, const float* w_wino, const float* bias, const float* alpha,
                                  const float* residual, float* y, float* preact,
                                  int B, int H, int W, int Cin, int Cout, int act, void* stream)
{
    if (!x || !w_wino || !y) return rn_set_error(RN_E_INVALID, "rn_conv2d_wino_fwd: null pointer");
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return rn_set_error(RN_E_INVALID, "rn_conv2d_wino_fwd: bad sizes");
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "rn_conv2d_wino_fwd: PReLU needs alpha");
    return rn_launch_conv_wino(x, w_wino, bias, alpha, residual, y, preact, B, H, W, 1, 1, Cin, Cout, act, 0, 1, (hipStream_t)stream);
}

extern "C" int rn_conv2d_wino4_supported(int Cin, int Cout) { return rn_wino4_supported(Cin, Cout) ? 1 : 0; }

extern "C" int rn_conv2d_wino4_fwd(const float* x, const float* w_wino4, const float* bias, const float* alpha,
                                   const float* residual, float* y, float* preact,
                                   int B, int H, int W, int Cin, int Cout, int transposed, int act, void* stream)
{
    if (!x || !w_wino4 || !y) return rn_set_error(RN_E_INVALID, "rn_conv2d_wino4_fwd: null pointer");
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return rn_set_error(RN_E_INVALID, "rn_conv2d_wino4_fwd: bad sizes");
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "rn_conv2d_wino4_fwd: PReLU needs alpha");
    // TF SAME for k = 4, s = 1: pad_before 1; the stride-1 transposed conv is the flipped conv with pad_before 4-1-1 = 2
    return rn_launch_conv_wino(x, w_wino4, bias, alpha, residual, y, preact, B, H, W, 1, 1, Cin, Cout, act, 1, transposed ? 2 : 1,
                               (hipStream_t)stream);
}

extern "C" int rn_conv2d_transpose_s2_wino_supported(int Cin, int Cout)
{
    static const bool off = getenv("RN_NO_WINOGRAD_S2") != nullptr;
    return (!off && rn_wino4_supported(Cin, Cout)) ? 1 : 0;
}

extern "C" int rn_conv2d_transpose_s2_wino_fwd(const float* x, const float* w_wino_s2, const float* bias, const float* alpha,
                                               const float* residual, float* y, float* preact,
                                               int B, int H, int W, int Cin, int Cout, int act, void* stream)
{
    if (!x || !w_wino_s2 || !y) return rn_set_error(RN_E_INVALID, "rn_conv2d_transpose_s2_wino_fwd: null pointer");
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return rn_set_error(RN_E_INVALID, "rn_conv2d_transpose_s2_wino_fwd: bad sizes");
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "rn_conv2d_transpose_s2_wino_fwd: PReLU needs alpha");
    return rn_launch_conv_wino(x, w_wino_s2, bias, alpha, residual, y, preact, B, H, W, 1, 1, Cin, Cout, act, 2, 1, (hipStream_t)stream);
}

extern "C" int rn_conv3d_wino_supported(int Cin, int Cout) { return rn_wino3d_supported(Cin, Cout) ? 1 : 0; }

extern "C" int rn_conv3d_wino_fwd(const float* x, const float* w_wino, const float* bias, const float* alpha,
                                  const float* residual, float* y, float* preact,
                                  int B, int H, int W, int D, int Cin, int Cout, int act, void* stream)
{
    if (!x || !w_wino || !y) return rn_set_error(RN_E_INVALID, "rn_conv3d_wino_fwd: null pointer");
    if (B < 1 || H < 1 || W < 1 || D < 1 || Cin < 1 || Cout < 1) return rn_set_error(RN_E_INVALID, "rn_conv3d_wino_fwd: bad sizes");
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "rn_conv3d_wino_fwd: PReLU needs alpha");
    return rn_launch_conv_wino(x, w_wino, bias, alpha, residual, y, preact, B, H, W, D, 3, Cin, Cout, act, 0, 1, (hipStream_t)stream);
}

// Transposed conv, TF SAME with output = in*s (input-gradient of the SAME forward conv):
//   y[o] = sum_{i,k : i*s + k - pb = o} x[i] w[k],  pb = pad_before of the forward conv.
// s = 1: forward conv with the filter flipped and pad_lo = k-1-pb.
// s = 2, k = 4 (pb = 1): even outputs o=2q   use x[q-1]*w[3] + x[q]*w[1]   (2 taps, pad_lo 1)
//                        odd  outputs o=2q+1 use x[q]*w[2]   + x[q+1]*w[0] (2 taps, pad_lo 0)
// i.e. 2^nd sub-pixel phases, each a dense 2-tap-per-dim conv written with output stride 2.
static int convT_nd(const float* x, const float* w, const float* bias, const float* alpha,
                    const float* residual, float* y, int B, const int* I, int nd, int Cin, int Cout,
                    int ksize, int stride, int act, hipStream_t st, const char* who, float* preact = nullptr)
{
    if (!x || !w || !y) return rn_set_error(RN_E_INVALID, "%s: null pointer", who);
    if (B < 1 || Cin < 1 || Cout < 1 || ksize < 1) return rn_set_error(RN_E_INVALID, "%s: bad sizes", who);
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "%s: PReLU needs alpha", who);
    RnConvProblem p;
    p.x = x; p.bias = bias; p.alpha = alpha; p.residual = residual; p.y = y; p.preact = preact;
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.Npad = rn_round_up(Cout, 32);
    p.act = act;
    int Ofull[3];
    for (int d = 0; d < 3; ++d) {
        p.I[d] = I[d]; p.S[d] = 1;
        Ofull[d] = (d < nd) ? I[d] * stride : 1;
    }
    long long fs[3];   // full-resolution output strides
    fs[2] = Cout; fs[1] = (long long)Ofull[2] * Cout; fs[0] = (long long)Ofull[1] * fs[1];
    p.os_b = (long long)Ofull[0] * fs[0];
    if (stride == 1) {
        const int pb = (ksize - 1) / 2;
        for (int d = 0; d < 3; ++d) {
            p.K[d] = (d < nd) ? ksize : 1;
            p.P[d] = (d < nd) ? ksize - 1 - pb : 0;
            p.O[d] = Ofull[d];
            p.os[d] = fs[d];
        }
        p.out_off = 0; p.w = w;
        return dispatch(p, st);
    }
    if (stride != 2 || ksize != 4) return rn_set_error(RN_E_UNSUPPORTED, "%s: need (k=4,s=2) or s=1", who);
    const int nphase = 1 << nd;
    const int ktaps = 1 << nd;                       // 2 taps per dim
    const size_t per_phase = (size_t)((ktaps * Cin + 3) / 4) * p.Npad * 4;
    for (int ph = 0; ph < nphase; ++ph) {
        p.out_off = 0;
        for (int d = 0; d < 3; ++d) {
            if (d < nd) {
                const int bit = (ph >> (nd - 1 - d)) & 1;
                p.K[d] = 2; p.P[d] = bit ? 0 : 1; p.O[d] = I[d];
                p.os[d] = 2 * fs[d];
                p.out_off += (long long)bit * fs[d];
            } else {
                p.K[d] = 1; p.P[d] = 0; p.O[d] = 1; p.os[d] = fs[d];
            }
        }
        p.w = w + (size_t)ph * per_phase;
        int rc = dispatch(p, st);
        if (rc != RN_OK) return rc;
    }
    return RN_OK;
}

extern "C" int rn_conv2d_transpose_fwd(const float* x, const float* w_packed, const float* bias,
                                       const float* alpha, const float* residual, float* y,
                                       int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                                       int act, void* stream)
{
    const int I[3] = {H, W, 1};
    return convT_nd(x, w_packed, bias, alpha, residual, y, B, I, 2, Cin, Cout, ksize, stride, act,
                    (hipStream_t)stream, "rn_conv2d_transpose_fwd");
}

extern "C" int rn_conv3d_transpose_fwd(const float* x, const float* w_packed, const float* bias,
                                       const float* alpha, const float* residual, float* y,
                                       int B, int H, int W, int D, int Cin, int Cout, int ksize, int stride,
                                       int act, void* stream)
{
    const int I[3] = {H, W, D};
    return convT_nd(x, w_packed, bias, alpha, residual, y, B, I, 3, Cin, Cout, ksize, stride, act,
                    (hipStream_t)stream, "rn_conv3d_transpose_fwd");
}

// projection_unit (tools/layer_util.py:8-22): [B,H,W,D,C] is read as [B,H,W,1,D*C] -- the
// reshape :20 costs nothing in channels-last -- and the 1x1 slim.conv2d + PReLU :21 is one
// implicit-GEMM launch with M = B*H*W, K = N = D*C.
extern "C" int rn_projection_fwd(const float* x, const float* w_packed, const float* bias, const float* alpha,
                                 float* y, int B, int H, int W, int D, int C, void* stream)
{
    if (D < 1 || C < 1) return rn_set_error(RN_E_INVALID, "rn_projection_fwd: bad sizes");
    const int F = D * C;
    const int I[3] = {H, W, 1}, k[3] = {1, 1, 1}, s[3] = {1, 1, 1};
    return conv_fwd_nd(x, w_packed, bias, alpha, nullptr, y, B, I, F, F, k, s,
                       alpha ? RN_ACT_PRELU : RN_ACT_NONE, (hipStream_t)stream, "rn_projection_fwd");
}


// =================================================================================================
// Training step (RenderNet_Shader.py:159-167): forward entry points that also emit the pre-activation,
// input gradients (dgrad) and filter gradients (wgrad) of every conv flavour.
// =================================================================================================
extern "C" int rn_conv3d_fwd_train(const float* x, const float* w_packed, const float* bias, const float* alpha,
                                   const float* residual, float* y, float* preact, int B, int H, int W, int D,
                                   int Cin, int Cout, const int* ksize, const int* stride, int act, void* stream)
{
    if (!ksize || !stride) return rn_set_error(RN_E_INVALID, "rn_conv3d_fwd_train: null ksize/stride");
    const int I[3] = {H, W, D};
    return conv_fwd_nd(x, w_packed, bias, alpha, residual, y, B, I, Cin, Cout, ksize, stride, act,
                       (hipStream_t)stream, "rn_conv3d_fwd_train", preact);
}

extern "C" int rn_conv2d_fwd_train(const float* x, const float* w_packed, const float* bias, const float* alpha,
                                   const float* residual, float* y, float* preact, int B, int H, int W,
                                   int Cin, int Cout, const int* ksize, const int* stride, int act, void* stream)
{
    if (!ksize || !stride) return rn_set_error(RN_E_INVALID, "rn_conv2d_fwd_train: null ksize/stride");
    const int I[3] = {H, W, 1}, k[3] = {ksize[0], ksize[1], 1}, s[3] = {stride[0], stride[1], 1};
    return conv_fwd_nd(x, w_packed, bias, alpha, residual, y, B, I, Cin, Cout, k, s, act,
                       (hipStream_t)stream, "rn_conv2d_fwd_train", preact);
}

extern "C" int rn_conv2d_transpose_fwd_train(const float* x, const float* w_packed, const float* bias,
                                             const float* alpha, const float* residual, float* y, float* preact,
                                             int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                                             int act, void* stream)
{
    const int I[3] = {H, W, 1};
    return convT_nd(x, w_packed, bias, alpha, residual, y, B, I, 2, Cin, Cout, ksize, stride, act,
                    (hipStream_t)stream, "rn_conv2d_transpose_fwd_train", preact);
}

extern "C" int rn_conv3d_transpose_fwd_train(const float* x, const float* w_packed, const float* bias,
                                             const float* alpha, const float* residual, float* y, float* preact,
                                             int B, int H, int W, int D, int Cin, int Cout, int ksize, int stride,
                                             int act, void* stream)
{
    const int I[3] = {H, W, D};
    return convT_nd(x, w_packed, bias, alpha, residual, y, B, I, 3, Cin, Cout, ksize, stride, act,
                    (hipStream_t)stream, "rn_conv3d_transpose_fwd_train", preact);
}

// dgrad of a SAME forward conv: dx[i] = sum_{o,t : o*s - pb + t = i} dz[o] * w[t]  (= TF's
// conv*_backprop_input).  stride 1: a forward conv over dz with the flipped filter and
// pad_lo = k-1-pb (w packed with RN_PACK_CONVT_S1 from the SAME TF tensor -- a conv filter
// [k..,Cin,Cout] read as a transposed-conv filter [k..,Cout_T=Cin,Cin_T=Cout]); strided: direct gather
// kernel over the forward pack (RN_PACK_CONV), Cin <= 16.
static int conv_dgrad_nd(const float* dz, const float* w, float* dx, int B, const int* I, int Cin, int Cout,
                         const int* k, const int* s, hipStream_t st, const char* who)
{
    if (!dz || !w || !dx) return rn_set_error(RN_E_INVALID, "%s: null pointer", who);
    if (B < 1 || Cin < 1 || Cout < 1) return rn_set_error(RN_E_INVALID, "%s: bad sizes", who);
    int O[3], P[3];
    bool unit = true;
    for (int d = 0; d < 3; ++d) {
        if (I[d] < 1 || k[d] < 1 || s[d] < 1) return rn_set_error(RN_E_INVALID, "%s: bad geometry", who);
        same_geom(I[d], k[d], s[d], O[d], P[d]);
        unit = unit && s[d] == 1;
    }
    if (!unit) return rn_launch_conv_dgrad_direct(dz, w, dx, B, I, Cin, O, Cout, k, s, P, st);
    RnConvProblem p;
    p.x = dz; p.w = w; p.bias = nullptr; p.alpha = nullptr; p.residual = nullptr; p.y = dx; p.preact = nullptr;
    p.B = B; p.Cin = Cout; p.Cout = Cin; p.Npad = rn_round_up(Cin, 32);
    for (int d = 0; d < 3; ++d) {
        p.I[d] = O[d]; p.O[d] = I[d]; p.K[d] = k[d]; p.S[d] = 1; p.P[d] = k[d] - 1 - P[d];
    }
    p.os[2] = Cin;
    p.os[1] = (long long)I[2] * Cin;
    p.os[0] = (long long)I[1] * p.os[1];
    p.os_b = (long long)I[0] * p.os[0];
    p.out_off = 0; p.act = RN_ACT_NONE;
    return dispatch(p, st);
}

extern "C" int rn_conv3d_dgrad(const float* dz, const float* w_packed, float* dx, int B, int H, int W, int D,
                               int Cin, int Cout, const int* ksize, const int* stride, void* stream)
{
    if (!ksize || !stride) return rn_set_error(RN_E_INVALID, "rn_conv3d_dgrad: null ksize/stride");
    const int I[3] = {H, W, D};
    return conv_dgrad_nd(dz, w_packed, dx, B, I, Cin, Cout, ksize, stride, (hipStream_t)stream, "rn_conv3d_dgrad");
}

extern "C" int rn_conv2d_dgrad(const float* dz, const float* w_packed, float* dx, int B, int H, int W,
                               int Cin, int Cout, const int* ksize, const int* stride, void* stream)
{
    if (!ksize || !stride) return rn_set_error(RN_E_INVALID, "rn_conv2d_dgrad: null ksize/stride");
    const int I[3] = {H, W, 1}, k[3] = {ksize[0], ksize[1], 1}, s[3] = {stride[0], stride[1], 1};
    return conv_dgrad_nd(dz, w_packed, dx, B, I, Cin, Cout, k, s, (hipStream_t)stream, "rn_conv2d_dgrad");
}

// dgrad of a transposed conv y = convT(x, w[k..,Cout,Cin], stride s): the SAME forward conv over dz
// [B, H*s, W*s(, D*s), Cout] with that TF tensor read as a conv filter [k.., in = Cout, out = Cin]
// (packed with RN_PACK_CONV), stride s.
extern "C" int rn_conv2d_transpose_dgrad(const float* dz, const float* w_packed, float* dx, int B, int H, int W,
                                         int Cin, int Cout, int ksize, int stride, void* stream)
{
    const int I[3] = {H * stride, W * stride, 1}, k[3] = {ksize, ksize, 1}, s[3] = {stride, stride, 1};
    return conv_fwd_nd(dz, w_packed, nullptr, nullptr, nullptr, dx, B, I, Cout, Cin, k, s, RN_ACT_NONE,
                       (hipStream_t)stream, "rn_conv2d_transpose_dgrad");
}

extern "C" int rn_conv3d_transpose_dgrad(const float* dz, const float* w_packed, float* dx, int B, int H, int W, int D,
                                         int Cin, int Cout, int ksize, int stride, void* stream)
{
    const int I[3] = {H * stride, W * stride, D * stride}, k[3] = {ksize, ksize, ksize}, s[3] = {stride, stride, stride};
    return conv_fwd_nd(dz, w_packed, nullptr, nullptr, nullptr, dx, B, I, Cout, Cin, k, s, RN_ACT_NONE,
                       (hipStream_t)stream, "rn_conv3d_transpose_dgrad");
}

// wgrad: dw (TF layout, ACCUMULATED -- zero it first) of a SAME forward conv ...
static int conv_wgrad_nd(const float* x, const float* dz, float* dw, int B, const int* I, int Cin, int Cout,
                         const int* k, const int* s, hipStream_t st, const char* who)
{
    if (!x || !dz || !dw) return rn_set_error(RN_E_INVALID, "%s: null pointer", who);
    int O[3], P[3];
    for (int d = 0; d < 3; ++d) {
        if (I[d] < 1 || k[d] < 1 || s[d] < 1) return rn_set_error(RN_E_INVALID, "%s: bad geometry", who);
        same_geom(I[d], k[d], s[d], O[d], P[d]);
    }
    return rn_launch_conv_wgrad(x, dz, dw, B, I, Cin, O, Cout, k, s, P, st);
}

extern "C" int rn_conv3d_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W, int D,
                               int Cin, int Cout, const int* ksize, const int* stride, void* stream)
{
    if (!ksize || !stride) return rn_set_error(RN_E_INVALID, "rn_conv3d_wgrad: null ksize/stride");
    const int I[3] = {H, W, D};
    return conv_wgrad_nd(x, dz, dw, B, I, Cin, Cout, ksize, stride, (hipStream_t)stream, "rn_conv3d_wgrad");
}

extern "C" int rn_conv3d_wgrad_split_supported(int Cin, int Cout) { return rn_conv3d_wgrad_split_ok(Cin, Cout) ? 1 : 0; }
extern "C" int rn_conv3d_wgrad_split(const float* x, const float* dz, float* dw, int B, int H, int W, int D, int Cin, int Cout, void* stream)
{
    if (!rn_conv3d_wgrad_split_ok(Cin, Cout)) return rn_set_error(RN_E_UNSUPPORTED, "rn_conv3d_wgrad_split: Cin=%d Cout=%d (3x3x3, stride 1, 32 -> 32 only)", Cin, Cout);
    return rn_launch_conv3d_wgrad_split(x, dz, dw, B, H, W, D, (hipStream_t)stream);
}

extern "C" int rn_conv2d_wino43_supported(int Cin, int Cout) { return rn_wino43_supported(RN_WINO_F43, Cin, Cout) ? 1 : 0; }
extern "C" int rn_conv2d_wino44_supported(int Cin, int Cout) { return rn_wino43_supported(RN_WINO_F44, Cin, Cout) ? 1 : 0; }
extern "C" int rn_conv2d_wino63_supported(int Cin, int Cout) { return rn_wino43_supported(RN_WINO_F63, Cin, Cout) ? 1 : 0; }
extern "C" size_t rn_conv2d_wino63_workspace_floats(int B, int H, int W, int Cin, int Cout)
{
    if (B < 1 || H < 1 || W < 1 || !rn_wino43_supported(RN_WINO_F63, Cin, Cout)) return 0;
    return rn_wino43_workspace_floats(RN_WINO_F63, B, H, W, Cin, Cout);
}
extern "C" size_t rn_conv2d_wino43_workspace_floats(int B, int H, int W, int Cin, int Cout)
{
    if (B < 1 || H < 1 || W < 1 || !rn_wino43_supported(RN_WINO_F43, Cin, Cout)) return 0;
    return rn_wino43_workspace_floats(RN_WINO_F43, B, H, W, Cin, Cout);
}
extern "C" size_t rn_conv2d_wino44_workspace_floats(int B, int H, int W, int Cin, int Cout)
{
    if (B < 1 || H < 1 || W < 1 || !rn_wino43_supported(RN_WINO_F44, Cin, Cout)) return 0;
    return rn_wino43_workspace_floats(RN_WINO_F44, B, H, W, Cin, Cout);
}
static int wino4x_fwd(int scheme, const char* who, const float* x, const float* w, const float* bias, const float* alpha,
                      const float* residual, float* y, float* preact, float* workspace, int B, int H, int W, int Cin, int Cout,
                      int pad_lo, int act, void* stream)
{
    if (!x || !w || !y || !workspace) return rn_set_error(RN_E_INVALID, "%s: null pointer", who);
    if (B < 1 || H < 1 || W < 1) return rn_set_error(RN_E_INVALID, "%s: bad sizes", who);
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "%s: PReLU needs alpha", who);
    return rn_launch_conv_wino43(scheme, x, w, bias, alpha, residual, y, preact, workspace, B, H, W, Cin, Cout, pad_lo, act,
                                 (hipStream_t)stream);
}
extern "C" int rn_conv2d_wino43_fwd(const float* x, const float* w, const float* bias, const float* alpha, const float* residual,
                                    float* y, float* preact, float* workspace, int B, int H, int W, int Cin, int Cout, int act,
                                    void* stream)
{
    return wino4x_fwd(RN_WINO_F43, "rn_conv2d_wino43_fwd", x, w, bias, alpha, residual, y, preact, workspace, B, H, W, Cin, Cout, 1, act, stream);
}
extern "C" int rn_conv2d_wino63_fwd(const float* x, const float* w, const float* bias, const float* alpha, const float* residual,
                                    float* y, float* preact, float* workspace, int B, int H, int W, int Cin, int Cout, int act,
                                    void* stream)
{
    return wino4x_fwd(RN_WINO_F63, "rn_conv2d_wino63_fwd", x, w, bias, alpha, residual, y, preact, workspace, B, H, W, Cin, Cout, 1, act, stream);
}
extern "C" int rn_conv2d_wino44_fwd(const float* x, const float* w, const float* bias, const float* alpha, const float* residual,
                                    float* y, float* preact, float* workspace, int B, int H, int W, int Cin, int Cout,
                                    int transposed, int act, void* stream)
{
    return wino4x_fwd(RN_WINO_F44, "rn_conv2d_wino44_fwd", x, w, bias, alpha, residual, y, preact, workspace, B, H, W, Cin, Cout,
                      transposed ? 2 : 1, act, stream);
}

extern "C" int rn_winograd_input_transform(int scheme, const float* x, float* V, int B, int H, int W, int C, int pad_lo, void* stream)
{
    if (!x || !V) return rn_set_error(RN_E_INVALID, "rn_winograd_input_transform: null pointer");
    if (rn_wino_scheme_nxi(scheme) == 0 || B < 1 || H < 1 || W < 1 || C < 4 || C % 4 != 0 || pad_lo < 0 || pad_lo > 3)
        return rn_set_error(RN_E_INVALID, "rn_winograd_input_transform: bad arguments");
    return rn_launch_wino_input(scheme, x, V, B, H, W, C, pad_lo, (hipStream_t)stream);
}
extern "C" int rn_winograd_gemm(int scheme, const float* V, const float* w, float* M, long long T, int Cin, int Cout, void* stream)
{
    if (!V || !w || !M) return rn_set_error(RN_E_INVALID, "rn_winograd_gemm: null pointer");
    return rn_launch_wino_gemm(scheme, V, w, M, T, Cin, Cout, (hipStream_t)stream);
}
extern "C" int rn_winograd_output_transform(int scheme, const float* M, const float* bias, const float* alpha, const float* residual,
                                            float* y, float* preact, int B, int H, int W, int C, int act, void* stream)
{
    if (!M || !y) return rn_set_error(RN_E_INVALID, "rn_winograd_output_transform: null pointer");
    if (rn_wino_scheme_nxi(scheme) == 0 || B < 1 || H < 1 || W < 1 || C < 4 || C % 4 != 0)
        return rn_set_error(RN_E_INVALID, "rn_winograd_output_transform: bad arguments");
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "rn_winograd_output_transform: PReLU needs alpha");
    return rn_launch_wino_output(scheme, M, bias, alpha, residual, y, preact, B, H, W, C, act, (hipStream_t)stream);
}

// ---- the multiply stage on the bf16 matrix pipe at fp32 accuracy (conv_wino_bf3.hip)
extern "C" int rn_winograd_split_supported(int scheme, int Cin, int Cout) { return rn_wino_bf3_supported(scheme, Cin, Cout) ? 1 : 0; }
extern "C" size_t rn_winograd_split_packed_bytes(int scheme, int Cin, int Cout)
{
    return rn_wino_bf3_supported(scheme, Cin, Cout) ? rn_wino_bf3_packed_bytes(scheme, Cin, Cout) : 0;
}
extern "C" size_t rn_winograd_split_v_bytes(int scheme, long long T, int Cin)
{
    // the channel contract of the GEMM stage these planes feed (rn_winograd_split_supported): Cin >= 32, Cin % 32 == 0
    return (rn_split_scheme_nxi(scheme & 0xff) == 0 || (scheme >> 8) > 1 || T < 1 || Cin < 32 || Cin % 32 != 0) ? 0 : rn_wino_bf3_v_bytes(scheme, T, Cin);
}
extern "C" size_t rn_winograd_split_workspace_bytes(int scheme, int B, int H, int W, int Cin, int Cout)
{
    if (B < 1 || H < 1 || W < 1 || !rn_wino_bf3_supported(scheme, Cin, Cout)) return 0;
    return rn_wino_bf3_workspace_bytes(scheme, B, H, W, Cin, Cout);
}
extern "C" int rn_winograd_split_pack(int scheme, const float* w_tf, void* w_split, int Cin, int Cout, int transposed, void* stream)
{
    if (!w_tf || !w_split) return rn_set_error(RN_E_INVALID, "rn_winograd_split_pack: null pointer");
    return rn_launch_wino_pack_bf3(scheme, w_tf, w_split, Cin, Cout, transposed ? 1 : 0, (hipStream_t)stream);
}
extern "C" int rn_winograd_split_input_transform(int scheme, const float* x, void* Vs, int B, int H, int W, int C, int pad_lo, void* stream)
{
    if (!x || !Vs) return rn_set_error(RN_E_INVALID, "rn_winograd_split_input_transform: null pointer");
    if (rn_split_scheme_nxi(scheme & 0xff) == 0 || (scheme >> 8) > 1 || B < 1 || H < 1 || W < 1 || C < 32 || C % 32 != 0 || pad_lo < 0 || pad_lo > 3)
        return rn_set_error(RN_E_INVALID, "rn_winograd_split_input_transform: bad arguments (C must be a multiple of 32: the GEMM stage's K contract)");
    return rn_launch_wino_input_bf3(scheme, x, Vs, B, H, W, C, pad_lo, (hipStream_t)stream);
}
extern "C" int rn_winograd_split_gemm(int scheme, const void* Vs, const void* w_split, float* M, long long T, int Cin, int Cout, void* stream)
{
    if (!Vs || !w_split || !M) return rn_set_error(RN_E_INVALID, "rn_winograd_split_gemm: null pointer");
    return rn_launch_wino_gemm_bf3(scheme, Vs, w_split, M, T, Cin, Cout, (hipStream_t)stream);
}
extern "C" int rn_conv2d_winograd_split_fwd(int scheme, const float* x, const void* w_split, const float* bias, const float* alpha,
                                            const float* residual, float* y, float* preact, void* workspace, int B, int H, int W,
                                            int Cin, int Cout, int transposed, int act, void* stream)
{
    if (!x || !w_split || !y || !workspace) return rn_set_error(RN_E_INVALID, "rn_conv2d_winograd_split_fwd: null pointer");
    if (B < 1 || H < 1 || W < 1) return rn_set_error(RN_E_INVALID, "rn_conv2d_winograd_split_fwd: bad sizes");
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "rn_conv2d_winograd_split_fwd: PReLU needs alpha");
    const int pad_lo = (scheme & 0xff) == RN_WINO_F11 ? 0 : ((scheme & 0xff) == RN_WINO_F44 && transposed) ? 2 : 1;
    return rn_launch_conv_wino_bf3(scheme, x, w_split, bias, alpha, residual, y, preact, workspace, B, H, W, Cin, Cout, pad_lo, act,
                                   (hipStream_t)stream);
}

extern "C" int rn_conv2d_winograd_split_fwd_ex(int scheme, const float* x, const void* w_split, const float* bias, const float* alpha,
                                               const float* residual, float* y, float* preact, void* workspace, int B, int H, int W,
                                               int Cin, int Cout, int transposed, int act, const void* amax_x, void* amax_y, void* stream)
{
    if (!x || !w_split || !y || !workspace) return rn_set_error(RN_E_INVALID, "rn_conv2d_winograd_split_fwd_ex: null pointer");
    if (B < 1 || H < 1 || W < 1) return rn_set_error(RN_E_INVALID, "rn_conv2d_winograd_split_fwd_ex: bad sizes");
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "rn_conv2d_winograd_split_fwd_ex: PReLU needs alpha");
    const int pad_lo = (scheme & 0xff) == RN_WINO_F11 ? 0 : ((scheme & 0xff) == RN_WINO_F44 && transposed) ? 2 : 1;
    return rn_launch_conv_wino_bf3_ex(scheme, x, w_split, bias, alpha, residual, y, preact, workspace, B, H, W, Cin, Cout, pad_lo, act,
                                      static_cast<const unsigned*>(amax_x), static_cast<unsigned*>(amax_y), (hipStream_t)stream);
}
extern "C" int rn_winograd_split_input_transform_ex(int scheme, const float* x, void* Vs, int B, int H, int W, int C, int pad_lo,
                                                    const void* amax_x, void* stream)
{
    if (!x || !Vs) return rn_set_error(RN_E_INVALID, "rn_winograd_split_input_transform_ex: null pointer");
    if (rn_split_scheme_nxi(scheme & 0xff) == 0 || (scheme >> 8) > 1 || B < 1 || H < 1 || W < 1 || C < 32 || C % 32 != 0 || pad_lo < 0 || pad_lo > 3)
        return rn_set_error(RN_E_INVALID, "rn_winograd_split_input_transform_ex: bad arguments (C must be a multiple of 32: the GEMM stage's K contract)");
    return rn_launch_wino_input_bf3_ex(scheme, x, Vs, B, H, W, C, pad_lo, static_cast<const unsigned*>(amax_x), (hipStream_t)stream);
}
extern "C" int rn_winograd_output_transform_ex(int scheme, const float* M, const float* bias, const float* alpha, const float* residual,
                                               float* y, float* preact, int B, int H, int W, int C, int act, void* amax_y, void* stream)
{
    if (!M || !y) return rn_set_error(RN_E_INVALID, "rn_winograd_output_transform_ex: null pointer");
    if (B < 1 || H < 1 || W < 1 || C < 4 || C % 4 != 0) return rn_set_error(RN_E_INVALID, "rn_winograd_output_transform_ex: bad sizes");
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "rn_winograd_output_transform_ex: PReLU needs alpha");
    if (amax_y) { const int rc = rn_launch_word(static_cast<unsigned*>(amax_y), nullptr, (hipStream_t)stream); if (rc != RN_OK) return rc; }
    return rn_launch_wino_output_amax(scheme, M, bias, alpha, residual, y, preact, B, H, W, C, act, static_cast<unsigned*>(amax_y), (hipStream_t)stream);
}
extern "C" int rn_absmax(const float* x, long long n, void* amax, void* stream)
{
    if (!x || !amax || n < 4 || n % 4 != 0) return rn_set_error(RN_E_INVALID, "rn_absmax: bad arguments");
    return rn_launch_absmax(x, (size_t)n, static_cast<unsigned*>(amax), (hipStream_t)stream);
}

// ---- the 3x3x3 32 -> 32 convs of the 3-D encoder on the bf16 matrix pipe at fp32 accuracy (conv3d_wino_bf3.hip)
extern "C" int rn_winograd_split_wgrad_supported(int scheme, int Cin, int Cout) { return rn_wino_bf3_wgrad_supported(scheme, Cin, Cout) ? 1 : 0; }
extern "C" size_t rn_winograd_split_wgrad_workspace_bytes(int scheme, int B, int H, int W, int Cin, int Cout)
{
    return rn_wino_bf3_wgrad_supported(scheme, Cin, Cout) ? rn_wino_bf3_wgrad_workspace_bytes(scheme, B, H, W, Cin, Cout) : 0;
}
extern "C" int rn_conv2d_winograd_split_wgrad(int scheme, const float* x, const float* dz, float* dw, void* workspace, int B, int H, int W,
                                              int Cin, int Cout, void* stream)
{
    if (!x || !dz || !dw || !workspace) return rn_set_error(RN_E_INVALID, "rn_conv2d_winograd_split_wgrad: null pointer");
    if (B < 1 || H < 1 || W < 1) return rn_set_error(RN_E_INVALID, "rn_conv2d_winograd_split_wgrad: bad sizes");
    return rn_launch_conv_wino_bf3_wgrad(scheme, x, dz, dw, workspace, B, H, W, Cin, Cout, static_cast<hipStream_t>(stream));
}
// ..._ex: operand format as a parameter (0 | RN_SPLIT_FMT_H2 >> 8 = 1) and the max|x| hand-over of the 2-D entries
extern "C" size_t rn_conv3d_winograd_split_packed_bytes_ex(int fmt, int Cin, int Cout)
{
    return (Cin == 32 && Cout == 32 && (fmt == 0 || fmt == 1)) ? rn_conv3d_wino_split_packed_bytes(fmt) : 0;
}
extern "C" int rn_conv3d_winograd_split_pack_ex(int fmt, const float* w_tf, void* w_split, int Cin, int Cout, int transposed, void* stream)
{
    if (!w_tf || !w_split) return rn_set_error(RN_E_INVALID, "rn_conv3d_winograd_split_pack_ex: null pointer");
    if (Cin != 32 || Cout != 32 || fmt < 0 || fmt > 1) return rn_set_error(RN_E_UNSUPPORTED, "rn_conv3d_winograd_split_pack_ex: Cin=%d Cout=%d fmt=%d", Cin, Cout, fmt);
    return rn_launch_conv3d_wino_split_pack(fmt, w_tf, w_split, transposed ? 1 : 0, (hipStream_t)stream);
}
extern "C" int rn_conv3d_winograd_split_fwd_ex(int fmt, const float* x, const void* w_split, const float* bias, const float* alpha, const float* residual,
                                               float* y, float* preact, int B, int H, int W, int D, int Cin, int Cout, int act,
                                               const void* amax_x, void* amax_scratch, void* amax_y, void* stream)
{
    if (!x || !w_split || !y) return rn_set_error(RN_E_INVALID, "rn_conv3d_winograd_split_fwd_ex: null pointer");
    if (!rn_conv3d_wino_bf3_supported(Cin, Cout)) return rn_set_error(RN_E_UNSUPPORTED, "rn_conv3d_winograd_split_fwd_ex: Cin=%d Cout=%d", Cin, Cout);
    return rn_launch_conv3d_wino_split(fmt, x, w_split, bias, alpha, residual, y, preact, B, H, W, D, act, static_cast<const unsigned*>(amax_x),
                                       static_cast<unsigned*>(amax_scratch), static_cast<unsigned*>(amax_y), (hipStream_t)stream);
}
extern "C" int rn_conv3d_winograd_split_supported(int Cin, int Cout) { return rn_conv3d_wino_bf3_supported(Cin, Cout) ? 1 : 0; }
extern "C" size_t rn_conv3d_winograd_split_packed_bytes(int Cin, int Cout)
{
    return (Cin == 32 && Cout == 32) ? rn_conv3d_wino_bf3_packed_bytes() : 0;
}
extern "C" int rn_conv3d_winograd_split_pack(const float* w_tf, void* w_split, int Cin, int Cout, int transposed, void* stream)
{
    if (!w_tf || !w_split) return rn_set_error(RN_E_INVALID, "rn_conv3d_winograd_split_pack: null pointer");
    if (Cin != 32 || Cout != 32) return rn_set_error(RN_E_UNSUPPORTED, "rn_conv3d_winograd_split_pack: Cin=%d Cout=%d (32 -> 32 only)", Cin, Cout);
    return rn_launch_conv3d_wino_pack_bf3(w_tf, w_split, transposed ? 1 : 0, (hipStream_t)stream);
}
extern "C" int rn_conv3d_winograd_split_fwd(const float* x, const void* w_split, const float* bias, const float* alpha, const float* residual,
                                            float* y, float* preact, int B, int H, int W, int D, int Cin, int Cout, int act, void* stream)
{
    if (!x || !w_split || !y) return rn_set_error(RN_E_INVALID, "rn_conv3d_winograd_split_fwd: null pointer");
    if (!rn_conv3d_wino_bf3_supported(Cin, Cout)) return rn_set_error(RN_E_UNSUPPORTED, "rn_conv3d_winograd_split_fwd: Cin=%d Cout=%d", Cin, Cout);
    return rn_launch_conv3d_wino_bf3(x, w_split, bias, alpha, residual, y, preact, B, H, W, D, act, (hipStream_t)stream);
}

extern "C" int rn_winograd_output_input_supported(int scheme, int H, int W, int C, int act)
{
    static const bool off = getenv("RN_NO_WINO_OUTIN") != nullptr;
    const int m = rn_wino_scheme_m(scheme);
    if (off || m == 0 || scheme == RN_WINO_F44 || H < 1 || W < 1) return 0;
    const int tw = (W + m - 1) / m;
    return (C >= 16 && C % 16 == 0 && tw * 8 <= 256 && (size_t)3 * m * (tw * m + 2) * 16 * sizeof(float) <= (size_t)160 * 1024 &&
            (act & ~RN_ACT_PRELU) == 0) ? 1 : 0;
}
extern "C" int rn_winograd_output_input_transform(int scheme, const float* M, const float* bias, const float* alpha, const float* residual,
                                                  float* y, float* V_next, int B, int H, int W, int C, int act, void* stream)
{
    if (!M || !V_next) return rn_set_error(RN_E_INVALID, "rn_winograd_output_input_transform: null pointer");
    if (B < 1 || !rn_winograd_output_input_supported(scheme, H, W, C, act))
        return rn_set_error(RN_E_UNSUPPORTED, "rn_winograd_output_input_transform: scheme=%d H=%d W=%d C=%d act=%d does not fit the fused tiling",
                            scheme, H, W, C, act);
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "rn_winograd_output_input_transform: PReLU needs alpha");
    const int rc = rn_launch_wino_outin(scheme, M, bias, alpha, residual, y, V_next, B, H, W, C, act, (hipStream_t)stream);
    return rc == RN_E_UNSUPPORTED ? rn_set_error(RN_E_UNSUPPORTED, "rn_winograd_output_input_transform: not applicable") : rc;
}

extern "C" int rn_conv2d_wino43_wgrad_supported(int Cin, int Cout) { return rn_wino43_wgrad_supported(RN_WINO_F43, Cin, Cout) ? 1 : 0; }
extern "C" int rn_conv2d_wino44_wgrad_supported(int Cin, int Cout) { return rn_wino43_wgrad_supported(RN_WINO_F44, Cin, Cout) ? 1 : 0; }
static size_t wino4x_wgrad_ws(int scheme, int B, int H, int W, int Cin, int Cout)
{
    if (B < 1 || H < 1 || W < 1 || !rn_wino43_wgrad_supported(scheme, Cin, Cout)) return 0;
    return rn_wino43_wgrad_workspace_floats(scheme, B, H, W, Cin, Cout);
}
extern "C" size_t rn_conv2d_wino43_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout) { return wino4x_wgrad_ws(RN_WINO_F43, B, H, W, Cin, Cout); }
extern "C" size_t rn_conv2d_wino44_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout) { return wino4x_wgrad_ws(RN_WINO_F44, B, H, W, Cin, Cout); }
static int wino4x_wgrad(int scheme, const char* who, const float* x, const float* dz, float* dw, float* workspace, int B, int H, int W,
                        int Cin, int Cout, void* stream)
{
    if (!x || !dz || !dw || !workspace) return rn_set_error(RN_E_INVALID, "%s: null pointer", who);
    if (B < 1 || H < 1 || W < 1) return rn_set_error(RN_E_INVALID, "%s: bad sizes", who);
    return rn_launch_conv_wino43_wgrad(scheme, x, dz, dw, workspace, B, H, W, Cin, Cout, (hipStream_t)stream);
}
extern "C" int rn_conv2d_wino43_wgrad(const float* x, const float* dz, float* dw, float* workspace, int B, int H, int W, int Cin,
                                      int Cout, void* stream)
{
    return wino4x_wgrad(RN_WINO_F43, "rn_conv2d_wino43_wgrad", x, dz, dw, workspace, B, H, W, Cin, Cout, stream);
}
extern "C" int rn_conv2d_wino44_wgrad(const float* x, const float* dz, float* dw, float* workspace, int B, int H, int W, int Cin,
                                      int Cout, void* stream)
{
    return wino4x_wgrad(RN_WINO_F44, "rn_conv2d_wino44_wgrad", x, dz, dw, workspace, B, H, W, Cin, Cout, stream);
}

extern "C" int rn_conv2d_wino_wgrad_supported(int Cin, int Cout) { return rn_wino_wgrad_supported(Cin, Cout) ? 1 : 0; }

extern "C" int rn_conv2d_wino_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W, int Cin, int Cout, void* stream)
{
    if (!x || !dz || !dw) return rn_set_error(RN_E_INVALID, "rn_conv2d_wino_wgrad: null pointer");
    if (B < 1 || H < 1 || W < 1) return rn_set_error(RN_E_INVALID, "rn_conv2d_wino_wgrad: bad sizes");
    return rn_launch_conv_wino_wgrad(x, dz, dw, B, H, W, Cin, Cout, (hipStream_t)stream);
}

extern "C" int rn_conv2d_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W,
                               int Cin, int Cout, const int* ksize, const int* stride, void* stream)
{
    if (!ksize || !stride) return rn_set_error(RN_E_INVALID, "rn_conv2d_wgrad: null ksize/stride");
    const int I[3] = {H, W, 1}, k[3] = {ksize[0], ksize[1], 1}, s[3] = {stride[0], stride[1], 1};
    return conv_wgrad_nd(x, dz, dw, B, I, Cin, Cout, k, s, (hipStream_t)stream, "rn_conv2d_wgrad");
}

// ... and of a transposed conv (dw in TF layout [k..,Cout,Cin]): the roles swap -- the full-resolution
// dz [B,H*s,W*s,Cout] is the "input" of the equivalent forward conv and x [B,H,W,Cin] its output grid.
extern "C" int rn_conv2d_transpose_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W,
                                         int Cin, int Cout, int ksize, int stride, void* stream)
{
    const int I[3] = {H * stride, W * stride, 1}, k[3] = {ksize, ksize, 1}, s[3] = {stride, stride, 1};
    return conv_wgrad_nd(dz, x, dw, B, I, Cout, Cin, k, s, (hipStream_t)stream, "rn_conv2d_transpose_wgrad");
}

extern "C" int rn_conv3d_transpose_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W, int D,
                                         int Cin, int Cout, int ksize, int stride, void* stream)
{
    const int I[3] = {H * stride, W * stride, D * stride}, k[3] = {ksize, ksize, ksize}, s[3] = {stride, stride, stride};
    return conv_wgrad_nd(dz, x, dw, B, I, Cout, Cin, k, s, (hipStream_t)stream, "rn_conv3d_transpose_wgrad");
}
