/*
 * librendernet_hip.so -- C ABI of the MI355X (gfx950) RenderNet forward render path.
 *
 * The reference (thunguyenphuoc/RenderNet) has no FFI layer: its boundary is the set of Python
 * callables in tools/ and the TF1 Session contract.  Each entry point below
 * replaces the TensorFlow op(s) one of those callables lowers to; the reference file:line is
 * cited per function.  The Python host (rendernet_amd/) binds these with ctypes and mirrors
 * the reference callables' names and arguments on top.
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers to float32, C-contiguous, channels-last
 *     (3-D features [B,H,W,D,C], 2-D features [B,H,W,C]) -- the layout TF uses;
 *   - caller owns every buffer; nothing is allocated, freed or synchronised inside;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default);
 *   - returns 0 on success, a negative RN_E_* code otherwise; never throws across the ABI;
 *     rn_last_error() returns a per-thread message for the last failing call;
 *   - re-entrant; no global mutable state besides that per-thread string.
 */
#ifndef RENDERNET_HIP_H
#define RENDERNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RN_VERSION 192            /* 0.1.92: + rn_epilogue_bwd_ws; 0.1.91: + the multiply stages on the 16-bit pipe at fp32-class accuracy (rn_winograd_split_*, rn_conv2d_winograd_split_fwd / _wgrad, rn_conv3d_winograd_split_*; RN_SPLIT_FMT_H2, the *_ex entries, rn_absmax) */

/* error codes */
#define RN_OK              0
#define RN_E_INVALID      -1      /* bad argument / unsupported shape */
#define RN_E_LAUNCH       -2      /* HIP launch error */
#define RN_E_UNSUPPORTED  -3

/* epilogue activation (applied as: v = acc + bias; act PReLU | ELU; v += residual; act sigmoid) */
#define RN_ACT_NONE     0
#define RN_ACT_PRELU    1         /* max(0,v) + alpha[c]*min(0,v)  -- tools/layer_util.py:27-45 */
#define RN_ACT_SIGMOID  2         /* RenderNet_Shader.py:127,130 */
#define RN_ACT_ELU      4         /* v > 0 ? v : exp(v)-1 -- tf.nn.elu of the shape decoder, Reconstruct_RenderNet_Face.py:49-68 */

/* weight-packing kinds for rn_pack_weights / rn_packed_weight_floats */
#define RN_PACK_CONV        0     /* TF conv filter  [k0,k1,k2,Cin,Cout]  (2-D: k2 = 1)            */
#define RN_PACK_CONVT_S1    1     /* TF conv_transpose filter [k0,k1,k2,Cout,Cin], stride 1:
                                     becomes a flipped forward conv                               */
#define RN_PACK_CONVT_S2    2     /* TF conv_transpose filter [4,4,(4,)Cout,Cin], stride 2:
                                     becomes 2^nd sub-pixel phase filters of 2 taps per dim       */

#define RN_PACK_CONV_WINO       3  /* TF conv filter [3,3,Cin,Cout] (ndim 2) or [3,3,3,Cin,Cout] (ndim 3) -> Winograd
                                     F(2x2,3x3) transformed U = G g G^T over the first two filter dims (16 planes,
                                     16*Cin*Cout floats, x3 for 3-D; Cin % 16 == 0, Cout % 16 == 0) for
                                     rn_conv2d_wino_fwd / rn_conv3d_wino_fwd                                     */
#define RN_PACK_CONVT_S1_WINO   4  /* TF conv_transpose filter [3,3,(3,)Cout,Cin], stride 1, taps flipped, same
                                     transform: the input gradient of a stride-1 3x3(x3) conv through
                                     rn_conv2d_wino_fwd / rn_conv3d_wino_fwd                                     */

#define RN_PACK_CONVT_S2_WINO   13 /* TF conv_transpose filter [4,4,Cout,Cin], STRIDE 2 (slim.conv2d_transpose, RenderNet_Shader.py:
                                    * 105-119: e_conv7 / e_conv8 / e_conv9): the four output phases as four 2x2 sub-filters, each in
                                    * Winograd F(2x2,2x2) form; 36*Cin*Cout floats; Cin, Cout multiples of 16 (rn_conv2d_transpose_s2_wino_fwd) */
#define RN_PACK_CONV_WINO43     7  /* TF conv filter [3,3,Cin,Cout] -> Winograd F(4x4,3x3) form U = G g G^T (36 planes,
                                     36*Cin*Cout floats, Cin % 32 == 0, Cout % 256 == 0): rn_conv2d_wino43_fwd      */
#define RN_PACK_CONVT_S1_WINO43 8  /* TF conv_transpose filter [3,3,Cout,Cin], stride 1, taps flipped, same transform
                                     (= the input gradient of a 3x3 conv when fed that conv's filter)               */
#define RN_PACK_CONV_WINO44     9  /* TF conv filter [4,4,Cin,Cout] -> Winograd F(4x4,4x4) form (49 planes, 49*Cin*Cout floats,
                                     Cin % 32 == 0, Cout % 256 == 0): rn_conv2d_wino44_fwd                            */
#define RN_PACK_CONVT_S1_WINO44 10 /* TF conv_transpose filter [4,4,Cout,Cin], stride 1, taps flipped, same transform:
                                     rn_conv2d_wino44_fwd with transposed = 1                                          */
#define RN_PACK_CONV_WINO63     11 /* TF conv filter [3,3,Cin,Cout] -> Winograd F(6x6,3x3) form (64 planes, 64*Cin*Cout floats,
                                     Cin % 32 == 0, Cout % 256 == 0): rn_conv2d_wino63_fwd                            */
#define RN_PACK_CONVT_S1_WINO63 12 /* TF conv_transpose filter [3,3,Cout,Cin], stride 1, taps flipped, same transform
                                     (= the input gradient of a 3x3 conv when fed that conv's filter)               */
#define RN_PACK_CONV_WINO4      5  /* TF conv filter [4,4,Cin,Cout] as four 2x2 sub-filters, each Winograd F(2x2,2x2)
                                     transformed (9 planes; 36*Cin*Cout floats; Cin % 16 == 0, Cout % 16 == 0) for
                                     rn_conv2d_wino4_fwd                                                          */
#define RN_PACK_CONVT_S1_WINO4  6  /* TF conv_transpose filter [4,4,Cout,Cin], stride 1, taps flipped, same transform:
                                     rn_conv2d_wino4_fwd with transposed = 1                                     */

int rn_version(void);
const char* rn_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Resampler.  Replaces tf_rotation_resampling (tools/resampling_voxel_grid.py:616-632 ->
 * :515-562, :564-614, :381-486) fused with tf_transform_voxel_to_match_image
 * (tools/model_util.py:41-49) and the spatial crop of tf_random_crop_voxel_image
 * (tools/model_util.py:95-98).
 *
 *   vox   [B,S,S,S,C]   source grid, addressed (dim1=z, dim2=y, dim3=x) as the reference does
 *   pose  [B,3]         (azimuth, elevation, scale) radians  -- "view_name:0"
 *   out   image_layout=1: [B,ph,pw,N,C] = X[b,h0+i,w0+j,k,c] with X = transform(resample(vox))
 *         image_layout=0: [B,N,N,N,C] raw tf_rotation_resampling output (h0=w0=0, ph=pw=N)
 * Semantics reproduced exactly: clamp-then-weight trilinear sampling, add_n order a..h.
 *
 * workspace: optional device scratch of rn_resample_workspace_bytes(B,S,C) bytes (per-item matrix +
 * occupancy bitmap).  With it (and C in {1,4}, S in {16,32,64,128}, N%32==0, ph%8==0, pw%8==0) the
 * tiled kernel runs: empty tiles are zero-filled at HBM speed and only occupied tiles are sampled,
 * from LDS.  Without it (NULL) a simple one-kernel path with identical results is used.
 * ---------------------------------------------------------------------------------------- */
size_t rn_resample_workspace_bytes(int B, int S, int C);

int rn_resample_fwd(const float* vox, const float* pose, float* out,
                    int B, int S, int N, int C,
                    int h0, int w0, int ph, int pw, int image_layout,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Same kernel, but the caller supplies the inverted 3x4 matrices M_inv [B,3,4]
 * (tools/resampling_voxel_grid.py:601-602) instead of the pose.  Source coordinates are
 * evaluated as ((m0*x + m1*y) + m2*z) + m3 with one rounding per operation, so the result is
 * bit-reproducible against oracle/resample.py mode="ordered". */
int rn_resample_affine_fwd(const float* vox, const float* m_inv, float* out,
                           int B, int S, int N, int C,
                           int h0, int w0, int ph, int pw, int image_layout,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Two sources resampled with ONE pose into ONE channel-concatenated output: the face renderer's
 * `tf.concat([tf_rotation_resampling(geometry), tf_rotation_resampling(texture)], axis=4)`
 * (RenderNet_Texture_Face_Normal.py:165-178; Reconstruct_RenderNet_Face.py:360-366).  vox_a [B,S,S,S,Ca],
 * vox_b [B,S,S,S,Cb] -> out [B,ph,pw,N,Ca+Cb] (image layout / window as above).  `affine` != 0: the second-to-last
 * pointer argument is M_inv [B,3,4] instead of the pose.  Bit-identical, channel by channel, to two rn_resample_* calls. */
int rn_resample_concat_fwd(const float* vox_a, int Ca, const float* vox_b, int Cb, const float* pose_or_m_inv,
                           int affine, float* out, int B, int S, int N, int h0, int w0, int ph, int pw,
                           int image_layout, void* stream);

/* pose [B,3] -> M_inv [B,3,4] (closed form of :526-602, evaluated in double). */
int rn_pose_to_affine(const float* pose, float* m_inv, int B, int S, int N, void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight packing (TF layout -> kernel layout), done once at load time on the device.
 * Packed layout: [phase][K/4][Npad][4] floats, K = taps*Cin, k = tap*Cin + c, Npad = Cout
 * rounded up to a multiple of 32.  rn_packed_weight_floats returns the element count.
 * kdims = {k0,k1,k2} (k2 = 1 for 2-D filters); ndim = 2 or 3.
 * ---------------------------------------------------------------------------------------- */
size_t rn_packed_weight_floats(int kind, int ndim, const int* kdims, int Cin, int Cout);
int rn_pack_weights(int kind, int ndim, const int* kdims, int Cin, int Cout,
                    const float* w_tf, float* w_packed, void* stream);

/* ------------------------------------------------------------------------------------------
 * Convolutions, TF "SAME" semantics, fused epilogue
 *     y = act( conv(x) + bias ) (+ residual)       (sigmoid, if requested, is applied last)
 * bias/alpha/residual may be NULL.  residual has the shape of y.
 *
 * rn_conv3d_fwd      replaces conv3d (tools/layer_util.py:228-265) + prelu (:27-45) +
 *                    the residual tf.add (:73, RenderNet_Shader.py:64).
 *                    x [B,H,W,D,Cin] -> y [B,ceil(H/s0),ceil(W/s1),ceil(D/s2),Cout]
 * rn_conv2d_fwd      replaces conv2d / slim.conv2d (tools/layer_util.py:147-183, :101-104;
 *                    RenderNet_Shader.py:83,87,98,102).  x [B,H,W,Cin] -> y [B,H/s,W/s,Cout]
 * rn_conv2d_transpose_fwd replaces conv2d_transpose / slim.conv2d_transpose
 *                    (tools/layer_util.py:186-226; RenderNet_Shader.py:106-129), k = 4, s in {1,2}.
 *                    x [B,H,W,Cin] -> y [B,H*s,W*s,Cout]
 * rn_conv3d_transpose_fwd replaces conv3d_transpose (tools/layer_util.py:269-309), k = 4.
 * rn_projection_fwd  replaces projection_unit (tools/layer_util.py:8-22): reads the 3-D tensor
 *                    x [B,H,W,D,C] directly as [B*H*W, D*C] (feature f = d*C + c; the reshape
 *                    is free in channels-last), 1x1 conv F->F, bias, PReLU, one kernel.
 * w_packed comes from rn_pack_weights with the matching kind.
 * ---------------------------------------------------------------------------------------- */
int rn_conv3d_fwd(const float* x, const float* w_packed, const float* bias, const float* alpha,
                  const float* residual, float* y,
                  int B, int H, int W, int D, int Cin, int Cout,
                  const int* ksize, const int* stride, int act, void* stream);

int rn_conv2d_fwd(const float* x, const float* w_packed, const float* bias, const float* alpha,
                  const float* residual, float* y,
                  int B, int H, int W, int Cin, int Cout,
                  const int* ksize, const int* stride, int act, void* stream);

int rn_conv2d_transpose_fwd(const float* x, const float* w_packed, const float* bias,
                            const float* alpha, const float* residual, float* y,
                            int B, int H, int W, int Cin, int Cout,
                            int ksize, int stride, int act, void* stream);

int rn_conv3d_transpose_fwd(const float* x, const float* w_packed, const float* bias,
                            const float* alpha, const float* residual, float* y,
                            int B, int H, int W, int D, int Cin, int Cout,
                            int ksize, int stride, int act, void* stream);

int rn_projection_fwd(const float* x, const float* w_packed, const float* bias, const float* alpha,
                      float* y, int B, int H, int W, int D, int C, void* stream);

/* Winograd F(2x2,3x3) form of rn_conv2d_fwd for the 3x3, stride-1, SAME convs of the 2-D trunk -- res_block_2d and
 * the *_skip convs (tools/layer_util.py:101-104; RenderNet_Shader.py:71-84, :91-99), 86.9 % of the path's FLOPs: same
 * arguments, epilogue and result (to fp32 rounding: the transforms reassociate the sum), 2.25x fewer multiplies, all
 * in fp32.  w_wino comes from rn_pack_weights(RN_PACK_CONV_WINO); packed with RN_PACK_CONVT_S1_WINO from the layer's
 * own TF filter it computes the layer's input gradient (dz [B,H,W,Cout_fwd] -> dx [B,H,W,Cin_fwd]).  preact may be
 * NULL (see rn_conv2d_fwd_train).  rn_conv2d_wino_supported: 1 when this library takes (Cin, Cout) on that path
 * (Cin % 16 == 0, Cout % 16 == 0, and the environment does not set RN_NO_WINOGRAD), else 0 -- use rn_conv2d_fwd.
 * Every pointer must be 16-byte aligned. */
/* rn_conv3d_wino_fwd: the same for the 3x3x3, stride-1 convs of the 3-D encoder -- conv3d in res_block_3d and
 * res1_skip (tools/layer_util.py:60-73; RenderNet_Shader.py:44-64): Winograd F(2x2,3x3) over (H,W), direct over the
 * three depth taps (in channels-last [B,H,W,D,C] the three depth neighbours of a voxel are 3*C contiguous floats, so
 * every output depth slice is a 2-D conv with 3*C input channels): 27 -> 12 multiplies per output and channel pair. */
int rn_conv2d_wino_supported(int Cin, int Cout);
/* rn_conv2d_wino43_fwd: the same layers through Winograd F(4x4,3x3) -- 36 multiplies per 4x4 outputs and channel pair
 * instead of 144 (F(2x2,3x3): 64) -- in three launches: input transform V = B^T d B of every 6x6 patch, 36 exact-fp32 MFMA
 * GEMMs M[xi] = V[xi] . U[xi], output transform A^T m A fused with the bias / PReLU / residual epilogue.  V and M live in
 * `workspace` (rn_conv2d_wino43_workspace_floats(B,H,W,Cin,Cout) floats, device memory, contents undefined afterwards).
 * Same epilogue contract as rn_conv2d_fwd_train.  fp32 rounding: about 5e-6 of max|y| at Cin = 1024 (interpolation points
 * 0, 1, -1, 2, -1/2, inf), against 1e-6 for the F(2x2) path; the end-to-end tolerance of the path is 1e-3.
 * Needs Cin % 32 == 0 and Cout % 256 == 0 (rn_conv2d_wino43_supported).
 *
 * rn_conv2d_wino44_fwd: the 4x4, stride-1 layers (e_conv5, e_conv6: slim.conv2d [4,4], RenderNet_Shader.py:86-88, :101-103)
 * through F(4x4,4x4): 49 multiplies per 4x4 outputs and channel pair instead of 256 (rn_conv2d_wino4_fwd: 144); same three
 * launches on 7x7 patches.  transposed = 0: SAME conv (one row/column of padding before), filter packed with
 * RN_PACK_CONV_WINO44; transposed = 1: stride-1 conv2d_transpose (two before), RN_PACK_CONVT_S1_WINO44 -- the input
 * gradient of the conv when fed the conv's own filter.  fp32 rounding about 1e-5 of max|y|.
 *
 * rn_conv2d_wino63_fwd: the 3x3 layers again, through F(6x6,3x3): 64 multiplies per 6x6 outputs and channel pair (1.78 per
 * output against 2.25) on 8x8 patches, interpolation points 0, +-1, +-2, +-1/2, inf; same three launches, same contract as
 * rn_conv2d_wino43_fwd.  Pays where the 6-pixel tile grid wastes little (64x64 maps: 11x11 tiles, 0.84 of F(4x4,3x3)'s
 * multiplies and transform traffic; 32x32 maps: no gain).  fp32 rounding about 2.7e-5 of max|y| at Cin = 1024 -- three times
 * F(4x4,3x3)'s, still 40 times inside the path's 1e-3 tolerance; RN_NO_WINOGRAD63=1 switches it off. */
int rn_conv2d_wino43_supported(int Cin, int Cout);
size_t rn_conv2d_wino43_workspace_floats(int B, int H, int W, int Cin, int Cout);
int rn_conv2d_wino43_fwd(const float* x, const float* w_wino43, const float* bias, const float* alpha,
                         const float* residual, float* y, float* preact, float* workspace,
                         int B, int H, int W, int Cin, int Cout, int act, void* stream);
int rn_conv2d_wino63_supported(int Cin, int Cout);
size_t rn_conv2d_wino63_workspace_floats(int B, int H, int W, int Cin, int Cout);
int rn_conv2d_wino63_fwd(const float* x, const float* w_wino63, const float* bias, const float* alpha,
                         const float* residual, float* y, float* preact, float* workspace,
                         int B, int H, int W, int Cin, int Cout, int act, void* stream);
int rn_conv2d_wino44_supported(int Cin, int Cout);
size_t rn_conv2d_wino44_workspace_floats(int B, int H, int W, int Cin, int Cout);
int rn_conv2d_wino44_fwd(const float* x, const float* w_wino44, const float* bias, const float* alpha,
                         const float* residual, float* y, float* preact, float* workspace,
                         int B, int H, int W, int Cin, int Cout, int transposed, int act, void* stream);
/* The three stages on their own; scheme = RN_WINO_F43 (6x6 tiles, 36 planes) | RN_WINO_F44 (7x7 tiles, 49 planes) |
 * RN_WINO_F63 (8x8 tiles, 64 planes).  T = B*ceil(H/m)*ceil(W/m) tiles, m = 4 (F43, F44) or 6 (F63); V [nxi][T][Cin], M [nxi][T][Cout]; every plane of V and M must stay below 2 GiB --
 * the rn_conv2d_wino4x_fwd entries split the batch themselves, these do not.  pad_lo: rows / columns of zero padding
 * before the first pixel (SAME conv: 1; stride-1 transposed conv: filter size - 2). */
#define RN_WINO_F43 0
#define RN_WINO_F44 1
#define RN_WINO_F63 2
#define RN_WINO_F11 3   /* split entries only (rn_winograd_split_* / rn_conv2d_winograd_split_fwd*): a 1x1 filter as ONE plane with identity
                         * transforms, i.e. a plain T x Cin x Cout GEMM on the split multiply stage, T = B*H*W pixels -- the projection unit's
                         * 1x1 conv (tools/layer_util.py:8-22) and its input gradient; w_tf [1,1,Cin,Cout] (transposed: [1,1,Cout,Cin]);
                         * Cin % 32 == 0 (Cin >= 32), Cout % 256 == 0 -- the contract of every split scheme (rn_winograd_split_supported);
                         * rn_winograd_split_v_bytes / _input_transform check the same Cin.  The exact-fp32 stage entries reject it. */
int rn_winograd_input_transform(int scheme, const float* x, float* V, int B, int H, int W, int C, int pad_lo, void* stream);
int rn_winograd_gemm(int scheme, const float* V, const float* w_packed, float* M, long long T, int Cin, int Cout, void* stream);
int rn_winograd_output_transform(int scheme, const float* M, const float* bias, const float* alpha, const float* residual,
                                 float* y, float* preact, int B, int H, int W, int C, int act, void* stream);
/* Output transform of one conv FUSED with the input transform of the next (stride-1 3x3, same channel count C on both sides --
 * the convs of a res_block_2d stack, tools/layer_util.py:91-105, RenderNet_Shader.py:71-84,91-99): M [nxi][T][C] of conv a ->
 * epilogue (bias, PReLU, residual) -> V [nxi][T][C] of conv b, the activation staying in LDS.  y (may be NULL) additionally
 * receives the activation [B,H,W,C] -- pass it when something else reads it later (a block's output is the next block's
 * residual).  V and y are bit-identical to rn_winograd_output_transform followed by rn_winograd_input_transform(pad_lo = 1).
 * scheme RN_WINO_F43 | RN_WINO_F63; act: 0 | RN_ACT_PRELU.  rn_winograd_output_input_supported says whether the tiling applies
 * (C % 16 == 0, at most 32 tiles per row, ring of 3*m rows within the CU's LDS); otherwise run the two launches. */
int rn_winograd_output_input_supported(int scheme, int H, int W, int C, int act);
int rn_winograd_output_input_transform(int scheme, const float* M, const float* bias, const float* alpha, const float* residual,
                                       float* y, float* V_next, int B, int H, int W, int C, int act, void* stream);
/* The same three launches with the multiply stage on the bf16 matrix pipe AT FP32 ACCURACY ("split" route; replaces the
 * slim.conv2d / tf.nn.conv2d of the wide stride-1 2-D layers -- tools/layer_util.py:91-105, :171, RenderNet_Shader.py:71-103 --
 * exactly like the entries above).  gfx950 runs bf16-input MFMA at 16x the rate of f32-input MFMA, both accumulating in
 * fp32.  Every fp32 value of V (input transform) and U (filter transform) is stored as the EXACT sum of three bf16 pieces
 * (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1), round to nearest even) and a product is the six piece products
 * with i + j <= 2; what is dropped is <= 3 * 2^-25 |x||y| per product, below the rounding of an fp32 FMA.  The result is
 * NOT bit-identical to the exact-fp32 route; against a float64 conv it measures the same or smaller error
 * (tests/test_gpu_wino_robust.py, same bars).
 *   rn_winograd_split_pack             w_tf (TF layout; transposed = 1: a conv_transpose filter, taps flipped) -> w_split,
 *                                      rn_winograd_split_packed_bytes(scheme,Cin,Cout) = nxi*Cin*Cout*6 bytes
 *   rn_winograd_split_input_transform  x [B,H,W,C] -> Vs [nxi][C/16][T][3][16] bf16 (rn_winograd_split_v_bytes(scheme,T,C))
 *   rn_winograd_split_gemm             Vs, w_split -> M [nxi][T][Cout] fp32 (then rn_winograd_output_transform)
 *   rn_conv2d_winograd_split_fwd       the three launches; workspace: rn_winograd_split_workspace_bytes(...) BYTES of device
 *                                      memory; scheme RN_WINO_F43 | RN_WINO_F63 (3x3) | RN_WINO_F44 (4x4; transposed = 1 pads
 *                                      two before); epilogue arguments as rn_conv2d_fwd_train.
 * Needs Cin % 32 == 0, Cout % 256 == 0 (rn_winograd_split_supported); planes below 2 GiB as above. */
/* Operand format of the split entries, OR-ed into `scheme`: 0 = three bf16 pieces per fp32 value, six piece products (above);
 * RN_SPLIT_FMT_H2 = the value divided by a power-of-two scale of its tensor (from max|x| of the tensor, found by the transform's
 * launcher itself, and the growth bound of the transform) as TWO fp16 pieces -- 22 mantissa bits for everything within 2^-18 of the
 * scaled maximum -- and three piece products, fp32 accumulation, the scales multiplied back on the way out: half the matrix work,
 * 4 instead of 6 bytes per operand element.  The buffers carry a 256-byte tail with the tensor's max|x|; sizes from the *_bytes entries. */
#define RN_SPLIT_FMT_H2 0x100
int rn_winograd_split_supported(int scheme, int Cin, int Cout);
size_t rn_winograd_split_packed_bytes(int scheme, int Cin, int Cout);
size_t rn_winograd_split_v_bytes(int scheme, long long T, int Cin);
size_t rn_winograd_split_workspace_bytes(int scheme, int B, int H, int W, int Cin, int Cout);
int rn_winograd_split_pack(int scheme, const float* w_tf, void* w_split, int Cin, int Cout, int transposed, void* stream);
int rn_winograd_split_input_transform(int scheme, const float* x, void* Vs, int B, int H, int W, int C, int pad_lo, void* stream);
int rn_winograd_split_gemm(int scheme, const void* Vs, const void* w_split, float* M, long long T, int Cin, int Cout, void* stream);
int rn_conv2d_winograd_split_fwd(int scheme, const float* x, const void* w_split, const float* bias, const float* alpha,
                                 const float* residual, float* y, float* preact, void* workspace, int B, int H, int W,
                                 int Cin, int Cout, int transposed, int act, void* stream);
/* ..._ex: format RN_SPLIT_FMT_H2 needs max|x| of every layer input.  A layer's launcher finds it with one pass over x -- or takes it
 * from `amax_x`, a device word holding the bit pattern of max|x| (or of an upper bound), which the launch that PRODUCED x wrote as its
 * `amax_y` (max over the tensor after the whole epilogue).  Both may be NULL; format 0 ignores amax_x.  rn_absmax: the stand-alone pass
 * (n % 4 == 0 floats, 16-byte aligned).
 * CONTRACT: a caller-supplied amax_x MUST be >= max|x| of the tensor the launch reads.  It is trusted, not re-checked: an understated
 * word gives a scale that is too small and the fp16 pieces of the larger values overflow to inf -- results are then undefined (inf / NaN
 * in the affected outputs, no error code).  Pass NULL when in doubt.  A non-finite max|x| (an activation that already overflowed fp32)
 * selects the largest power-of-two scale: the output then carries inf / NaN like the exact-fp32 route's, deterministically. */
int rn_conv2d_winograd_split_fwd_ex(int scheme, const float* x, const void* w_split, const float* bias, const float* alpha,
                                    const float* residual, float* y, float* preact, void* workspace, int B, int H, int W,
                                    int Cin, int Cout, int transposed, int act, const void* amax_x, void* amax_y, void* stream);
int rn_winograd_split_input_transform_ex(int scheme, const float* x, void* Vs, int B, int H, int W, int C, int pad_lo,
                                         const void* amax_x, void* stream);
int rn_winograd_output_transform_ex(int scheme, const float* M, const float* bias, const float* alpha, const float* residual,
                                    float* y, float* preact, int B, int H, int W, int C, int act, void* amax_y, void* stream);
int rn_absmax(const float* x, long long n, void* amax, void* stream);
/* Filter gradient of the same layers (tf.nn.conv2d_backprop_filter of slim.conv2d [3,3] / [4,4], stride 1: tools/layer_util.py:101-104,
 * RenderNet_Shader.py:71-103) with the reduction over the tiles on the bf16 pipe, same arithmetic: dw [R,R,Cin,Cout] += ...;
 * scheme RN_WINO_F43 (3x3) or RN_WINO_F44 (4x4); x [B,H,W,Cin], dz [B,H,W,Cout]; workspace of
 * rn_winograd_split_wgrad_workspace_bytes bytes.  The exact-fp32 counterpart is rn_conv2d_wino43_wgrad / rn_conv2d_wino44_wgrad. */
int rn_winograd_split_wgrad_supported(int scheme, int Cin, int Cout);
size_t rn_winograd_split_wgrad_workspace_bytes(int scheme, int B, int H, int W, int Cin, int Cout);
int rn_conv2d_winograd_split_wgrad(int scheme, const float* x, const float* dz, float* dw, void* workspace, int B, int H, int W,
                                   int Cin, int Cout, void* stream);
/* The 3x3x3, stride-1, 32 -> 32 channel convs of the 3-D encoder (res_block_3d, res1_skip: tools/layer_util.py:60-73 ->
 * tf.nn.conv3d :253; RenderNet_Shader.py:51-64) on the bf16 matrix pipe at fp32 accuracy: Winograd F(2x2,3x3) over (H, W),
 * direct over depth like rn_conv3d_wino_fwd, every product taken as six bf16 piece products with fp32 accumulation (see the
 * split entries above), fused in one launch (filter fragments register-resident, transforms in registers, 16-byte pixel
 * epilogue).  x, y [B,H,W,D,32]; w_split from rn_conv3d_winograd_split_pack (rn_conv3d_winograd_split_packed_bytes bytes;
 * w_tf = the TF filter [3,3,3,32,32]; transposed = 1: the input-gradient form -- taps flipped, channel roles swapped -- of
 * the same tensor); epilogue arguments as rn_conv3d_fwd_train.  Measured (MI355X, B = 24, 64x64x32 layer): 0.50 ms against
 * rn_conv3d_wino_fwd's 0.82 ms, error 2.4e-7 .. 3.8e-7 of max|y| against 3.2e-7 .. 4.9e-7.  The Python surface uses it
 * whenever the split multiply stage is selected (RN_WINO_GEMM=split; RN_CONV3D_SPLIT=0 | 1 overrides). */
int rn_conv3d_winograd_split_supported(int Cin, int Cout);
size_t rn_conv3d_winograd_split_packed_bytes(int Cin, int Cout);
int rn_conv3d_winograd_split_pack(const float* w_tf, void* w_split, int Cin, int Cout, int transposed, void* stream);
int rn_conv3d_winograd_split_fwd(const float* x, const void* w_split, const float* bias, const float* alpha, const float* residual,
                                 float* y, float* preact, int B, int H, int W, int D, int Cin, int Cout, int act, void* stream);
/* ..._ex: the operand format as a parameter (fmt 0 = three bf16 pieces; 1 = two fp16 pieces of value / tensor scale, see RN_SPLIT_FMT_H2)
 * and the max|x| hand-over: amax_x = device word holding the bit pattern of max|x| (NULL: the launcher makes a pass over x into
 * amax_scratch, a device word of the caller's); amax_y (may be NULL) receives max|y|. */
size_t rn_conv3d_winograd_split_packed_bytes_ex(int fmt, int Cin, int Cout);
int rn_conv3d_winograd_split_pack_ex(int fmt, const float* w_tf, void* w_split, int Cin, int Cout, int transposed, void* stream);
int rn_conv3d_winograd_split_fwd_ex(int fmt, const float* x, const void* w_split, const float* bias, const float* alpha, const float* residual,
                                    float* y, float* preact, int B, int H, int W, int D, int Cin, int Cout, int act,
                                    const void* amax_x, void* amax_scratch, void* amax_y, void* stream);
int rn_conv3d_wino_supported(int Cin, int Cout);
/* rn_conv2d_wino4_fwd: the 4x4, stride-1 layers -- e_conv5, e_conv6 (slim.conv2d [4,4], RenderNet_Shader.py:86-88, :101-103;
 * transposed = 0, SAME padding (1,2)) and e_conv7_1 (slim.conv2d_transpose [4,4] stride 1, :109-111; transposed = 1: the
 * flipped conv with padding (2,1)), and their input gradients (a conv's is the transposed form of the same filter and vice
 * versa).  The 4x4 filter is the sum of four 2x2 sub-filters applied to the input shifted by (0|2, 0|2) pixels; each is a
 * Winograd F(2x2,2x2): 36 multiplies per 2x2 outputs and channel pair instead of 64, in fp32.  Cin % 16 == 0, Cout % 16 == 0. */
int rn_conv2d_wino4_supported(int Cin, int Cout);
int rn_conv2d_wino4_fwd(const float* x, const float* w_wino4, const float* bias, const float* alpha,
                        const float* residual, float* y, float* preact,
                        int B, int H, int W, int Cin, int Cout, int transposed, int act, void* stream);

/* 4x4 STRIDE-2 SAME transposed conv (slim.conv2d_transpose, RenderNet_Shader.py:105-119; tools/layer_util.py:186-226) through
 * Winograd F(2x2,2x2) per output phase: y[2m+pa, 2n+pb] is a 2x2 conv of the input, 9 multiplies per 4 outputs instead of 16,
 * exact fp32 MFMA; the four phases are items of ONE launch.  x [B,H,W,Cin] -> y [B,2H,2W,Cout]; w_wino_s2 from
 * rn_pack_weights(RN_PACK_CONVT_S2_WINO); epilogue as rn_conv2d_fwd_train (bias, preact, PReLU / ELU, residual, sigmoid). */
int rn_conv2d_transpose_s2_wino_supported(int Cin, int Cout);
int rn_conv2d_transpose_s2_wino_fwd(const float* x, const float* w_wino_s2, const float* bias, const float* alpha,
                                    const float* residual, float* y, float* preact,
                                    int B, int H, int W, int Cin, int Cout, int act, void* stream);
int rn_conv3d_wino_fwd(const float* x, const float* w_wino, const float* bias, const float* alpha,
                       const float* residual, float* y, float* preact,
                       int B, int H, int W, int D, int Cin, int Cout, int act, void* stream);
int rn_conv2d_wino_fwd(const float* x, const float* w_wino, const float* bias, const float* alpha,
                       const float* residual, float* y, float* preact,
                       int B, int H, int W, int Cin, int Cout, int act, void* stream);

/* fully_connected (tools/layer_util.py:311-343): y[B,out] = act(x[B,in] @ w[in,out] + bias).
 * w is the TF matrix unpacked ([in,out] row-major). */
int rn_fully_connected_fwd(const float* x, const float* w, const float* bias, const float* alpha,
                           float* y, int B, int in_features, int out_features, int act, void* stream);

/* Stand-alone PReLU (tools/layer_util.py:27-45) for callers that do not use the fused
 * epilogues: y = max(0,x) + alpha[c]*min(0,x), c = last (channel) dim of size C; n = #elements. */
int rn_prelu_fwd(const float* x, const float* alpha, float* y, size_t n, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Phong composite of the demo (tools/Phong_shading.py:202-228 black-background branch with
 * mask, :162-200, :138-148).  normals [B,H,W,3] in [0,1]; light_dir [B,3] (not normalised),
 * light_col [B,3]; out [B,H,W,3].
 * ---------------------------------------------------------------------------------------- */
int rn_phong_composite_fwd(const float* normals, const float* light_dir, const float* light_col,
                           float ambient, float k_diffuse, float* out,
                           int B, int H, int W, void* stream);

/* Phong composite, all flavours of tools/Phong_shading.py, and its gradient (the differentiable tf_phong_composite of
 * the inverse-rendering graph, Reconstruct_RenderNet_Face.py:377-378):
 *   n = (img-0.5)/|img-0.5|;  D = clip(k_diffuse * max(n . l/|l|, 0) * light_col, 0, 1);  mask = sigmoid(255*s - thr)
 *   shading = clip(mask*(ambient + D) + (1-mask), 0, 1);  out = shading (* albedo when albedo != NULL)
 * mask_mode: NP_BLACK s=|img| thr 150 (np_mask :138-148) | NP_WHITE s=|1-img| thr 80 (np_mask_white :150-160) |
 *            TF_BLACK s=|img| thr 80 (tf_mask :23-32)    | TF_WHITE s=sqrt(3)-|img| thr 80 (tf_mask_white :34-44) |
 *            NO_MASK  shading = clip(ambient + D)        (with_mask=False, :104-105 / :222-223)
 * normals, albedo, out, dout, dnormals, dalbedo: [B,H,W,3]; light_dir, light_col: [B,3].
 * bwd: dnormals / dalbedo are WRITTEN, dlight_dir [B,3] is ACCUMULATED (atomics; zero it first); each may be NULL. */
#define RN_PHONG_NP_BLACK 0
#define RN_PHONG_NP_WHITE 1
#define RN_PHONG_TF_BLACK 2
#define RN_PHONG_TF_WHITE 3
#define RN_PHONG_NO_MASK  4
int rn_phong_composite_ex_fwd(const float* normals, const float* light_dir, const float* light_col,
                              const float* albedo, float ambient, float k_diffuse, float* out,
                              int B, int H, int W, int mask_mode, void* stream);
int rn_phong_composite_bwd(const float* normals, const float* light_dir, const float* light_col,
                           const float* albedo, float ambient, float k_diffuse, const float* dout,
                           float* dnormals, float* dlight_dir, float* dalbedo,
                           int B, int H, int W, int mask_mode, void* stream);

/* ==========================================================================================
 * Training step (BASELINE config 4).  Replaces what TensorFlow's autodiff derives from
 * `tf.train.AdamOptimizer(...).minimize(recon_loss)` (RenderNet_Shader.py:159-167) for the ops of
 * tools/layer_util.py: per conv flavour a forward that also emits the pre-activation, the input
 * gradient (dgrad), the filter gradient (wgrad); the backward of the fused epilogue; the loss;
 * the Adam update.  Same conventions as above (device pointers, stream-ordered, caller-owned).
 * ========================================================================================== */

/* Forward entry points of the section above with one extra output: `preact` (shape of y, may be
 * NULL) receives z = conv(x) + bias, the value BEFORE PReLU / residual / sigmoid, which the PReLU
 * backward needs (tools/layer_util.py:27-45: d/dalpha = sum dy*min(z,0) is lost once z is clipped). */
int rn_conv3d_fwd_train(const float* x, const float* w_packed, const float* bias, const float* alpha,
                        const float* residual, float* y, float* preact,
                        int B, int H, int W, int D, int Cin, int Cout,
                        const int* ksize, const int* stride, int act, void* stream);
int rn_conv2d_fwd_train(const float* x, const float* w_packed, const float* bias, const float* alpha,
                        const float* residual, float* y, float* preact,
                        int B, int H, int W, int Cin, int Cout,
                        const int* ksize, const int* stride, int act, void* stream);
int rn_conv2d_transpose_fwd_train(const float* x, const float* w_packed, const float* bias,
                                  const float* alpha, const float* residual, float* y, float* preact,
                                  int B, int H, int W, int Cin, int Cout,
                                  int ksize, int stride, int act, void* stream);
int rn_conv3d_transpose_fwd_train(const float* x, const float* w_packed, const float* bias,
                                  const float* alpha, const float* residual, float* y, float* preact,
                                  int B, int H, int W, int D, int Cin, int Cout,
                                  int ksize, int stride, int act, void* stream);

/* fully_connected (tools/layer_util.py:311-343; the texture decoder, RenderNet_Texture_Face_Normal.py:34-46) with
 * the pre-activation saved, and its backward: dw [in,out] += x^T dz (ACCUMULATED; zero it first), dx [B,in] = dz w^T.
 * dx or dw may be NULL.  Bias / PReLU gradients come from rn_epilogue_bwd on rows [B,out]. */
int rn_fully_connected_fwd_train(const float* x, const float* w, const float* bias, const float* alpha,
                                 float* y, float* preact, int B, int in_features, int out_features, int act, void* stream);
int rn_fully_connected_bwd(const float* x, const float* w, const float* dz, float* dx, float* dw,
                           int B, int in_features, int out_features, void* stream);

/* Backward of the fused epilogue  y = sigmoid?( prelu?(z) + residual ),  z = conv + bias, rows [M,C]:
 *   dt = dy * y*(1-y) if act has RN_ACT_SIGMOID (needs y);  the residual's gradient is dt;
 *   dz = dt * (z > 0 ? 1 : alpha[c]) and dalpha[c] += sum_rows dt*min(z,0) if RN_ACT_PRELU (needs z);
 *   dz = dy * (y < 0 ? y+1 : 1) if RN_ACT_ELU (needs y, TF's EluGrad);
 *   dbias[c] += sum_rows dz.
 * dz may alias dy or be NULL; dbias / dalpha are ACCUMULATED (atomics) and may be NULL. */
int rn_epilogue_bwd(const float* dy, const float* z, const float* y, const float* alpha,
                    float* dz, float* dbias, float* dalpha, size_t M, int C, int act, void* stream);
/* The same with a caller-owned workspace of rn_epilogue_bwd_workspace_floats(M, C) floats (contents need not survive the call; one per
 * stream): on large tensors the per-channel sums then go through per-row-block partials and a second small launch instead of
 * ~2 C x 512 same-address atomics -- a serialised tail of ~20 us per call on the 1024-channel layers (the backward of
 * tools/layer_util.py:27-45 under RenderNet_Shader.py:165-167; the training step of the shader net 85.4 -> 81.0 ms).  ws = NULL: rn_epilogue_bwd. */
size_t rn_epilogue_bwd_workspace_floats(size_t M, int C);
int rn_epilogue_bwd_ws(const float* dy, const float* z, const float* y, const float* alpha,
                       float* dz, float* dbias, float* dalpha, size_t M, int C, int act,
                       float* ws, size_t ws_floats, void* stream);

/* Input gradients (tf.nn.conv*_backprop_input).  H,W(,D) are the FORWARD INPUT sizes of the layer.
 *   rn_conv{2,3}d_dgrad            dz [B,ceil(H/s)..,Cout] -> dx [B,H,W(,D),Cin].  stride 1: pack the
 *                                  layer's TF filter with RN_PACK_CONVT_S1 (a conv filter [k..,Cin,Cout]
 *                                  read as a transposed-conv filter); strided (Cin <= 16): pack it with
 *                                  RN_PACK_CONV.
 *   rn_conv{2,3}d_transpose_dgrad  dz [B,H*s,W*s(,D*s),Cout] -> dx [B,H,W(,D),Cin]; pack the layer's TF
 *                                  filter [k..,Cout,Cin] with RN_PACK_CONV (read as a conv filter
 *                                  [k.., in=Cout, out=Cin]). */
int rn_conv3d_dgrad(const float* dz, const float* w_packed, float* dx, int B, int H, int W, int D,
                    int Cin, int Cout, const int* ksize, const int* stride, void* stream);
int rn_conv2d_dgrad(const float* dz, const float* w_packed, float* dx, int B, int H, int W,
                    int Cin, int Cout, const int* ksize, const int* stride, void* stream);
int rn_conv2d_transpose_dgrad(const float* dz, const float* w_packed, float* dx, int B, int H, int W,
                              int Cin, int Cout, int ksize, int stride, void* stream);
int rn_conv3d_transpose_dgrad(const float* dz, const float* w_packed, float* dx, int B, int H, int W, int D,
                              int Cin, int Cout, int ksize, int stride, void* stream);

/* Filter gradients (tf.nn.conv*_backprop_filter), written in the TF layout of the layer's filter
 * ([k..,Cin,Cout] for convs, [k..,Cout,Cin] for transposed convs) and ACCUMULATED into dw with fp32
 * atomics: zero dw (or keep a running gradient in it) before the call.  x is the layer's forward
 * input, dz the gradient w.r.t. its pre-activation. */
int rn_conv3d_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W, int D,
                    int Cin, int Cout, const int* ksize, const int* stride, void* stream);
int rn_conv2d_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W,
                    int Cin, int Cout, const int* ksize, const int* stride, void* stream);
/* rn_conv3d_wgrad for the 3x3x3, stride-1, 32 -> 32 convs of the 3-D encoder (res_block_3d / res1_skip: tools/layer_util.py:60-73,
 * RenderNet_Shader.py:51-64; tf.nn.conv3d_backprop_filter_v2 under AdamOptimizer.minimize, :165-167) with the reduction over the positions
 * on the bf16 matrix pipe at fp32 accuracy: every fp32 value of x and dz as three bf16 pieces (exact sum, made on the fly), six piece
 * products, fp32 accumulation -- the arithmetic of the split forward entries.  Same contract: dw [3,3,3,32,32] ACCUMULATED with fp32 atomics. */
int rn_conv3d_wgrad_split_supported(int Cin, int Cout);
int rn_conv3d_wgrad_split(const float* x, const float* dz, float* dw, int B, int H, int W, int D, int Cin, int Cout, void* stream);
int rn_conv2d_transpose_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W,
                              int Cin, int Cout, int ksize, int stride, void* stream);
/* rn_conv2d_wgrad for the 3x3, stride-1 convs (res_block_2d, *_skip: tools/layer_util.py:101-104) through the Winograd
 * identity dg = G^T [ sum_tiles (B^T d B) .* (A dY A^T) ] G: 16 multiplies per 2x2-output tile and channel pair instead
 * of 36, fp32; same contract (dw [3,3,Cin,Cout] ACCUMULATED with fp32 atomics).  Cin % 64 == 0 and Cout % 64 == 0
 * (rn_conv2d_wino_wgrad_supported). */
int rn_conv2d_wino_wgrad_supported(int Cin, int Cout);
/* ... and through F(4x4,3x3) on the three-launch path's GEMM structure (conv_wino43_wgrad.hip): V = B^T d B of the input,
 * dM = A dY A^T of the output gradient, 36 exact-fp32 MFMA GEMMs dU[xi] = V[xi]^T . dM[xi] over the tiles (K-split into
 * planes), dw += G^T (sum dU) G.  36 multiplies per 4x4 outputs and channel pair.  Cin % 256 == 0 and Cout % 256 == 0;
 * `workspace`: rn_conv2d_wino43_wgrad_workspace_floats(B,H,W,Cin,Cout) floats of device memory. */
int rn_conv2d_wino43_wgrad_supported(int Cin, int Cout);
size_t rn_conv2d_wino43_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout);
int rn_conv2d_wino43_wgrad(const float* x, const float* dz, float* dw, float* workspace, int B, int H, int W, int Cin, int Cout,
                           void* stream);
/* the same for the 4x4, stride-1 convs (e_conv5, e_conv6) through F(4x4,4x4): dw [4,4,Cin,Cout], 49 GEMMs */
int rn_conv2d_wino44_wgrad_supported(int Cin, int Cout);
size_t rn_conv2d_wino44_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout);
int rn_conv2d_wino44_wgrad(const float* x, const float* dz, float* dw, float* workspace, int B, int H, int W, int Cin, int Cout,
                           void* stream);
int rn_conv2d_wino_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W, int Cin, int Cout, void* stream);
int rn_conv3d_transpose_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W, int D,
                              int Cin, int Cout, int ksize, int stride, void* stream);

/* Backward of the resampler (what TensorFlow's autodiff derives from tools/resampling_voxel_grid.py:381-486 and
 * :515-602; used by the reference's inverse rendering, Reconstruct_RenderNet_Face.py:360-364, :402).
 *   rn_resample_affine_bwd  dout has the shape of the forward output (image_layout / window as in the forward call).
 *                           dvox [B,S,S,S,C] += scatter of weight*dout (may be NULL); dm [B,3,4] += d(loss)/d(M_inv)
 *                           (may be NULL; needs vox).  Both are ACCUMULATED with atomics: zero them first.
 *   rn_pose_to_affine_bwd   dpose [B,3] += J^T dm with J = d(M_inv)/d(azimuth, elevation, scale) of rn_pose_to_affine.
 * For a pose-driven forward call: m = rn_pose_to_affine(pose); rn_resample_affine_bwd(..., m, ...); rn_pose_to_affine_bwd. */
int rn_resample_affine_bwd(const float* vox, const float* m_inv, const float* dout, float* dvox, float* dm,
                           int B, int S, int N, int C, int h0, int w0, int ph, int pw, int image_layout, void* stream);
int rn_pose_to_affine_bwd(const float* pose, const float* dm, float* dpose, int B, int S, int N, void* stream);
/* rn_resample_affine_bwd for one source of rn_resample_concat_fwd: dout holds dout_channels per sample and this source's C
 * channels start at dout_offset.  One call per source; dm accumulates over them. */
int rn_resample_affine_bwd_strided(const float* vox, const float* m_inv, const float* dout, int dout_channels,
                                   int dout_offset, float* dvox, float* dm, int B, int S, int N, int C,
                                   int h0, int w0, int ph, int pw, int image_layout, void* stream);

/* tf.nn.dropout(x, keep_prob) = x / keep_prob * floor(keep_prob + U[0,1))  (RenderNet_Shader.py:39,43,47,88,103,107-123
 * with tools/layer_util.py:124-131; README default keep_prob 0.75 for training).  The uniforms come from a counter-based
 * generator: element e uses word e%4 of Philox4x32-10(counter = (e/4, stream_id), key = seed), u = (word >> 8) * 2^-24.
 * No mask is stored: calling it again with the same (seed, stream_id) on the output gradient IS the backward pass.
 * y may alias x.  keep_prob in (0, 1].  Any float-aligned pointers (16-byte aligned ones take the vector path); n = 0 is a
 * no-op.  The mask of element e is the same whatever the alignment. */
int rn_dropout(const float* x, float* y, size_t n, float keep_prob, unsigned long long seed,
               unsigned long long stream_id, void* stream);

/* Reconstruction loss and d(loss)/d(pred)  (RenderNet_Shader.py:159-163).
 *   mode 0: binary cross-entropy  sum_elems -(t*log(1e-6+p) + (1-t)*log(1e-6+1-p)) / divisor   (divisor = batch)
 *   mode 1: mean squared error    sum_elems (t-p)^2 / divisor                                  (divisor = #elements)
 * *loss_sum (double, device) is ACCUMULATED; dpred (may be NULL) gets the gradient.  With the batch
 * sharded over ranks pass the GLOBAL divisor: per-rank values and gradients then simply add up. */
int rn_loss_fwd_bwd(const float* pred, const float* target, float* dpred, double* loss_sum,
                    size_t n, double divisor, int mode, void* stream);

/* tf.train.AdamOptimizer update (RenderNet_Shader.py:166) on a flat buffer of n floats:
 *   g' = grad_scale*g;  m = b1*m + (1-b1)*g';  v = b2*v + (1-b2)*g'^2;  p -= lr_t * m / (sqrt(v) + eps)
 * lr_t = lr * sqrt(1-b2^t)/(1-b1^t) is computed by the caller (TF's formulation).  16-byte aligned. */
int rn_adam_step(float* param, const float* grad, float* m, float* v, size_t n,
                 float lr_t, float beta1, float beta2, float eps, float grad_scale, void* stream);

/* tf.train.GradientDescentOptimizer update, p -= lr*grad (the latent-variable optimisers of the inverse-rendering
 * loop, Reconstruct_RenderNet_Face.py:397-413). */
int rn_sgd_step(float* param, const float* grad, size_t n, float lr, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RENDERNET_HIP_H */
