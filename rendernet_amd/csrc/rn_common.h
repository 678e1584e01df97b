// Internal declarations shared by the HIP translation units of librendernet_hip.so (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/rendernet_hip.h"

// Generic "gridded" convolution problem: every conv / transposed-conv phase / 1x1 projection of
// the path is lowered to this one description (channels-last everywhere).
//   out[b, o0, o1, o2, n] = epi( sum_{t0,t1,t2,c} x[b, o0*S0-P0+t0, o1*S1-P1+t1, o2*S2-P2+t2, c]
//                                                * Wp[(t0*K1+t1)*K2+t2][c][n] )
// out-of-range input coordinates contribute zero (TF SAME).  The output element lives at
//   y + out_off + b*os_b + o0*os[0] + o1*os[1] + o2*os[2] + n          (elements)
// which lets a stride-2 transposed conv be written as 2^nd interleaved sub-pixel phases.
struct RnConvProblem {
    const float* x;
    const float* w;          // packed [K/4][Npad][4]
    const float* bias;       // [Cout] or null
    const float* alpha;      // [Cout] or null (PReLU)
    const float* residual;   // same addressing as y, or null
    float* y;
    float* preact;           // optional second output z = conv + bias (before PReLU/residual/sigmoid), or null
    int B, I[3], Cin;        // input  [B, I0, I1, I2, Cin]
    int O[3], Cout, Npad;    // output grid visited by this launch, channels, padded channels
    int K[3], S[3], P[3];    // taps, stride, pad_lo
    long long os_b, os[3], out_off;
    int act;
};

int rn_set_error(int code, const char* fmt, ...);
int rn_check_launch(const char* what);

// launchers implemented in the kernel translation units
int rn_launch_conv_igemm(const RnConvProblem& p, hipStream_t st);     // conv_igemm.hip  (MFMA)
int rn_launch_conv_direct(const RnConvProblem& p, hipStream_t st);    // conv_direct.hip (VALU)
int rn_launch_conv_tiled(const RnConvProblem& p, hipStream_t st);     // conv_tiled.hip  (VALU, LDS-tiled stem / tail shapes; RN_E_UNSUPPORTED = no match)
bool rn_igemm_supported(const RnConvProblem& p);
int rn_launch_conv3d_drun(const RnConvProblem& p, hipStream_t st);     // conv3d_drun.hip (MFMA, 3^3 s1, N=32)
bool rn_drun_supported(const RnConvProblem& p);

bool rn_wino_supported(int Cin, int Cout);                                                                 // conv_wino.hip
bool rn_wino3d_supported(int Cin, int Cout);
bool rn_wino4_supported(int Cin, int Cout);
int rn_wino_ntiles(int mode, int Cout);
int rn_launch_conv_wino(const float* x, const float* u, const float* bias, const float* alpha, const float* residual,
                        float* y, float* preact, int B, int H, int W, int D, int KD, int Cin, int Cout, int act,
                        int mode, int pad, hipStream_t st);

// The "scheme" of a 1x1 filter on the split multiply stage (conv_wino_bf3.hip): one plane, identity transforms -- the three launches are then
// split x into the GEMM's row format / one T x Cin x Cout GEMM on the 16-bit matrix pipe / the conv epilogue.  Exists only in the split path
// (RN_WINO_F11 in include/rendernet_hip.h): the exact-fp32 helpers below do not know it (rn_wino_scheme_nxi(RN_WINO_F11) == 0).
struct WinoF11 {
    static constexpr int M = 1, TA = 1, R = 1, NXI = 1;
    static constexpr float AT(int, int) { return 1.f; }
    static constexpr double G(int, int) { return 1.0; }
    static constexpr float BT(int, int) { return 1.f; }
};
int rn_split_scheme_nxi(int scheme);                  // conv_wino_bf3.hip: rn_wino_scheme_nxi / _m that also know RN_WINO_F11 (1 plane, 1 pixel)
int rn_split_scheme_m(int scheme);
bool rn_wino43_supported(int scheme, int Cin, int Cout);                                                  // conv_wino43.hip
int rn_wino_scheme_nxi(int scheme);
int rn_wino_scheme_r(int scheme);
int rn_wino_scheme_m(int scheme);                     // output pixels per tile side (4 | 4 | 6)
long long rn_wino43_plane_limit();                    // 2 GiB, or RN_WINO43_MAX_PLANE (tests)
size_t rn_wino43_workspace_floats(int scheme, int B, int H, int W, int Cin, int Cout);
int rn_launch_wino_pack(int scheme, const float* w_tf, float* u, int Cin, int Cout, int transposed, hipStream_t st);
int rn_launch_wino_input(int scheme, const float* x, float* V, int B, int H, int W, int C, int pad_lo, hipStream_t st);
int rn_launch_wino_gemm(int scheme, const float* V, const float* u, float* M, long long T, int Cin, int Cout, hipStream_t st);
int rn_launch_wino_outin(int scheme, const float* M, const float* bias, const float* alpha, const float* residual, float* y,
                         float* V, int B, int H, int W, int C, int act, hipStream_t st);   // RN_E_UNSUPPORTED (no message) = does not apply
int rn_launch_wino_output(int scheme, const float* M, const float* bias, const float* alpha, const float* residual, float* y,
                          float* preact, int B, int H, int W, int C, int act, hipStream_t st);
int rn_launch_wino_output_amax(int scheme, const float* M, const float* bias, const float* alpha, const float* residual, float* y,
                               float* preact, int B, int H, int W, int C, int act, unsigned* amax, hipStream_t st);
int rn_launch_conv_wino43(int scheme, const float* x, const float* u, const float* bias, const float* alpha, const float* residual,
                          float* y, float* preact, float* ws, int B, int H, int W, int Cin, int Cout, int pad_lo, int act, hipStream_t st);
bool rn_wino_bf3_supported(int scheme, int Cin, int Cout);                                                // conv_wino_bf3.hip
size_t rn_wino_bf3_packed_bytes(int scheme, int Cin, int Cout);
size_t rn_wino_bf3_v_bytes(int scheme, long long T, int Cin);
size_t rn_wino_bf3_workspace_bytes(int scheme, int B, int H, int W, int Cin, int Cout);
int rn_launch_wino_pack_bf3(int scheme, const float* w_tf, void* us, int Cin, int Cout, int transposed, hipStream_t st);
int rn_launch_wino_input_bf3(int scheme, const float* x, void* Vs, int B, int H, int W, int C, int pad_lo, hipStream_t st);
int rn_launch_word(unsigned* dst, const unsigned* src, hipStream_t st);                                                   // *dst = src ? *src : 0, as a kernel (graph-safe)
int rn_launch_absmax(const float* x, size_t n, unsigned* out, hipStream_t st);                                           // *out = bits of max|x| (n % 4 == 0)
int rn_launch_wino_input_bf3_ex(int scheme, const float* x, void* Vs, int B, int H, int W, int C, int pad_lo, const unsigned* amax_x, hipStream_t st);
int rn_launch_conv_wino_bf3_ex(int scheme, const float* x, const void* us, const float* bias, const float* alpha, const float* residual,
                               float* y, float* preact, void* ws, int B, int H, int W, int Cin, int Cout, int pad_lo, int act,
                               const unsigned* amax_x, unsigned* amax_y, hipStream_t st);
int rn_launch_wino_gemm_bf3(int scheme, const void* Vs, const void* us, float* M, long long T, int Cin, int Cout, hipStream_t st);
int rn_launch_gemm_bf3_planes(int nplanes, int tag, const void* Vs, const void* us, float* M, long long T, int Cin, int Cout, hipStream_t st);
bool rn_wino_bf3_wgrad_supported(int scheme, int Cin, int Cout);                                          // conv_wino_bf3_wgrad.hip
size_t rn_wino_bf3_wgrad_workspace_bytes(int scheme, int B, int H, int W, int Cin, int Cout);
int rn_launch_conv_wino_bf3_wgrad(int scheme, const float* x, const float* dz, float* dw, void* ws, int B, int H, int W, int Cin, int Cout, hipStream_t st);
int rn_launch_conv_wino_bf3(int scheme, const float* x, const void* us, const float* bias, const float* alpha, const float* residual,
                            float* y, float* preact, void* ws, int B, int H, int W, int Cin, int Cout, int pad_lo, int act, hipStream_t st);
bool rn_conv3d_wino_bf3_supported(int Cin, int Cout);                                                     // conv3d_wino_bf3.hip
size_t rn_conv3d_wino_bf3_packed_bytes();
int rn_launch_conv3d_wino_pack_bf3(const float* w_tf, void* us, int transposed, hipStream_t st);
int rn_launch_conv3d_wino_bf3(const float* x, const void* us, const float* bias, const float* alpha, const float* residual,
                              float* y, float* preact, int B, int H, int W, int D, int act, hipStream_t st);
size_t rn_conv3d_wino_split_packed_bytes(int fmt);
int rn_launch_conv3d_wino_split_pack(int fmt, const float* w_tf, void* us, int transposed, hipStream_t st);
int rn_launch_conv3d_wino_split(int fmt, const float* x, const void* us, const float* bias, const float* alpha, const float* residual,
                                float* y, float* preact, int B, int H, int W, int D, int act, const unsigned* amax_x, unsigned* scratch_amax,
                                unsigned* amax_y, hipStream_t st);
bool rn_wino43_wgrad_supported(int scheme, int Cin, int Cout);                                            // conv_wino43_wgrad.hip
size_t rn_wino43_wgrad_workspace_floats(int scheme, int B, int H, int W, int Cin, int Cout);
int rn_launch_conv_wino43_wgrad(int scheme, const float* x, const float* dz, float* dw, float* ws, int B, int H, int W, int Cin, int Cout, hipStream_t st);
bool rn_wino_wgrad_supported(int Cin, int Cout);                                                          // conv_wino_wgrad.hip
int rn_launch_conv_wino_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W, int Cin, int Cout, hipStream_t st);

bool rn_conv3d_wgrad_split_ok(int Cin, int Cout);
int rn_launch_conv3d_wgrad_split(const float* x, const float* dz, float* dw, int B, int H, int W, int D, hipStream_t st);
int rn_launch_conv_wgrad(const float* A, const float* G, float* dw, int B, const int* I, int Ca,
                         const int* O, int Cg, const int* K, const int* S, const int* P, hipStream_t st);   // conv_wgrad.hip
int rn_launch_conv_dgrad_direct(const float* dz, const float* w_fwd_packed, float* dx, int B, const int* I, int Cin,
                                const int* O, int Cout, const int* K, const int* S, const int* P, hipStream_t st); // train_kernels.hip

static inline int rn_round_up(int a, int b) { return (a + b - 1) / b * b; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: set it once per (kernel, device) -- a
// process may drive several GPUs -- and never again (the call costs tens of microseconds, which at batch 1 is a third of
// a short launch).  The cache is per thread: no locks, and a second thread merely repeats the idempotent call once.
int rn_ensure_dynamic_lds(const void* kernel, size_t bytes);     // capi.hip
