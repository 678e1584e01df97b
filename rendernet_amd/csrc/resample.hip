// Pose-conditioned trilinear voxel resampler, fused with the voxel->image axis transform and
// the spatial crop.  Replaces tf_rotation_resampling (tools/resampling_voxel_grid.py:616-632:
// pose -> matrices :515-562, inverse warp :564-614, 8-gather interpolation :381-486),
// tf_transform_voxel_to_match_image (tools/model_util.py:41-49) and the voxel half of
// tf_random_crop_voxel_image (tools/model_util.py:95-98) -- one kernel, one pass over HBM:
// reads the S^3 source (L2-resident: 1 MiB per item at S=64), writes the N^3 target once,
// directly in the network's input layout.  None of the reference's temporaries (homogeneous
// meshgrid [B,4,N^3], transformed coords [B,3,N^3], 8 gathered tensors, 8 weight tensors)
// is materialised.
//
// Semantics reproduced exactly: x0=floor(x), x1=x0+1, BOTH clamped to
// [0,S-1], weights computed from the CLAMPED indices, products ((wx*wy)*wz)*I, sum order
// a,b,c,d,e,f,g,h; every operation individually rounded (no FMA contraction), so the affine
// entry point is bit-exact against the NumPy oracle evaluated in the same order.
#include "rn_common.h"
#include <math.h>
#include <stdlib.h>

// One rounding per multiply and per add, as in the reference's op-by-op graph: no FMA contraction
// anywhere in this file (also enforced with -ffp-contract=off in rendernet_amd/build.py).
#pragma clang fp contract(off)

struct ResampleArgs {
    const float* vox;     // [B,S,S,S,C]
    const float* mat;     // FROM_POSE ? pose [B,3] : m_inv [B,12]
    float* out;
    int B, S, N, C;
    int h0, w0, ph, pw;
    int image_layout;
};

__device__ __forceinline__ void pose_to_affine_dev(const float* pose, int S, int N, float* m /*12*/)
{
    // closed form of inverse(T_new_inv * Sc * (Rot_Z*Rot_Y) * T) rows 0:3
    // (tools/resampling_voxel_grid.py:526-602): p_src = (1/s) R^T (p_out - N/2) + S/2
    const double az = (double)pose[0] - 1.5707963267948966;
    const double el = (double)pose[1];
    const double is = 1.0 / (double)pose[2];
    const double ca = cos(az), sa = sin(az), ce = cos(el), se = sin(el);
    const double rt[3][3] = {{ce * ca, -se * ca, sa}, {se, ce, 0.0}, {-ce * sa, se * sa, ca}};
    const double hn = 0.5 * N, hs = 0.5 * S;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double a0 = rt[r][0] * is, a1 = rt[r][1] * is, a2 = rt[r][2] * is;
        m[r * 4 + 0] = (float)a0; m[r * 4 + 1] = (float)a1; m[r * 4 + 2] = (float)a2;
        m[r * 4 + 3] = (float)(hs - (a0 + a1 + a2) * hn);
    }
}

__global__ void pose_to_affine_kernel(const float* pose, float* m_inv, int B, int S, int N)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) pose_to_affine_dev(pose + 3 * b, S, N, m_inv + 12 * b);
}

// one source coordinate: ((m0*x + m1*y) + m2*z) + m3, one rounding per op
__device__ __forceinline__ float coord(float m0, float m1, float m2, float m3, float x, float y, float z)
{
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m0, x), __fmul_rn(m1, y)), __fmul_rn(m2, z)), m3);
}

template <int CT>   // CT = compile-time channel count (1, 4) or 0 for a runtime loop
__device__ __forceinline__ void sample_point(const float* __restrict__ vb, int S, int C,
                                             float x, float y, float z, float* __restrict__ o)
{
    const int mx = S - 1;
    int x0 = (int)floorf(x), y0 = (int)floorf(y), z0 = (int)floorf(z);
    int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    x0 = min(max(x0, 0), mx); x1 = min(max(x1, 0), mx);
    y0 = min(max(y0, 0), mx); y1 = min(max(y1, 0), mx);
    z0 = min(max(z0, 0), mx); z1 = min(max(z1, 0), mx);
    const float ax = __fsub_rn((float)x1, x), bx = __fsub_rn(x, (float)x0);
    const float ay = __fsub_rn((float)y1, y), by = __fsub_rn(y, (float)y0);
    const float az = __fsub_rn((float)z1, z), bz = __fsub_rn(z, (float)z0);
    const float wa = __fmul_rn(__fmul_rn(ax, ay), az);
    const float wb = __fmul_rn(__fmul_rn(ax, by), az);
    const float wc = __fmul_rn(__fmul_rn(bx, ay), az);
    const float wd = __fmul_rn(__fmul_rn(bx, by), az);
    const float we = __fmul_rn(__fmul_rn(ax, ay), bz);
    const float wf = __fmul_rn(__fmul_rn(ax, by), bz);
    const float wg = __fmul_rn(__fmul_rn(bx, ay), bz);
    const float wh = __fmul_rn(__fmul_rn(bx, by), bz);
    const int S2 = S * S;
    const int ia = z0 * S2 + y0 * S + x0, ib = z0 * S2 + y1 * S + x0;
    const int ic = z0 * S2 + y0 * S + x1, id = z0 * S2 + y1 * S + x1;
    const int ie = z1 * S2 + y0 * S + x0, iff = z1 * S2 + y1 * S + x0;
    const int ig = z1 * S2 + y0 * S + x1, ih = z1 * S2 + y1 * S + x1;
    const int nc = CT ? CT : C;
#pragma unroll
    for (int c = 0; c < nc; ++c) {
        float v = __fmul_rn(wa, vb[(size_t)ia * nc + c]);
        v = __fadd_rn(v, __fmul_rn(wb, vb[(size_t)ib * nc + c]));
        v = __fadd_rn(v, __fmul_rn(wc, vb[(size_t)ic * nc + c]));
        v = __fadd_rn(v, __fmul_rn(wd, vb[(size_t)id * nc + c]));
        v = __fadd_rn(v, __fmul_rn(we, vb[(size_t)ie * nc + c]));
        v = __fadd_rn(v, __fmul_rn(wf, vb[(size_t)iff * nc + c]));
        v = __fadd_rn(v, __fmul_rn(wg, vb[(size_t)ig * nc + c]));
        v = __fadd_rn(v, __fmul_rn(wh, vb[(size_t)ih * nc + c]));
        o[c] = v;
    }
}

// Grid: blockIdx.x walks (b, i, j-group); each thread owns 4 consecutive depth samples k of one
// (i, j) line, so a wave writes 1 KiB contiguous (C=1) and the 8 gathers of neighbouring lanes
// fall into neighbouring source cells.
template <int CT, bool FROM_POSE>
__global__ __launch_bounds__(256)
void resample_kernel(const ResampleArgs a)
{
    __shared__ float msh[12];
    const int N = a.N;
    const int kthreads = N / 4;                      // threads per depth line
    const int lines_per_block = 256 / kthreads;      // (i,j) lines per block
    const long long nlines = (long long)a.B * a.ph * a.pw;
    const long long line0 = (long long)blockIdx.x * lines_per_block;
    // all lines of a block belong to one batch item (pw*ph multiple of lines_per_block is
    // enforced by the launcher)
    const int b = (int)(line0 / ((long long)a.ph * a.pw));
    if (threadIdx.x == 0) {
        if (FROM_POSE) pose_to_affine_dev(a.mat + 3 * b, a.S, N, msh);
        else for (int q = 0; q < 12; ++q) msh[q] = a.mat[12 * b + q];
    }
    __syncthreads();
    const long long line = line0 + threadIdx.x / kthreads;
    if (line >= nlines) return;
    const int k0 = (threadIdx.x % kthreads) * 4;
    const int ij = (int)(line - (long long)b * a.ph * a.pw);
    const int i = ij / a.pw + a.h0, j = ij % a.pw + a.w0;

    const float m00 = msh[0], m01 = msh[1], m02 = msh[2], m03 = msh[3];
    const float m10 = msh[4], m11 = msh[5], m12 = msh[6], m13 = msh[7];
    const float m20 = msh[8], m21 = msh[9], m22 = msh[10], m23 = msh[11];

    // output grid point of element (i,j,k):  image layout X[b,i,j,k] = R[b, z=j, y=N-1-i, x=k]
    // (tools/model_util.py:47-48); raw layout R[b, z=i, y=j, x=k].
    const float gy = a.image_layout ? (float)(N - 1 - i) : (float)j;
    const float gz = a.image_layout ? (float)j : (float)i;
    const float* vb = a.vox + (size_t)b * a.S * a.S * a.S * (CT ? CT : a.C);
    const int nc = CT ? CT : a.C;
    float* op = a.out + ((size_t)line * N + k0) * nc;

    if (CT == 1) {
        float r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float gx = (float)(k0 + q);
            sample_point<1>(vb, a.S, 1, coord(m00, m01, m02, m03, gx, gy, gz),
                            coord(m10, m11, m12, m13, gx, gy, gz), coord(m20, m21, m22, m23, gx, gy, gz), &r[q]);
        }
        *reinterpret_cast<float4*>(op) = make_float4(r[0], r[1], r[2], r[3]);
    } else if (CT == 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float gx = (float)(k0 + q);
            float r[4];
            sample_point<4>(vb, a.S, 4, coord(m00, m01, m02, m03, gx, gy, gz),
                            coord(m10, m11, m12, m13, gx, gy, gz), coord(m20, m21, m22, m23, gx, gy, gz), r);
            *reinterpret_cast<float4*>(op + q * 4) = make_float4(r[0], r[1], r[2], r[3]);
        }
    } else {
        for (int q = 0; q < 4; ++q) {
            const float gx = (float)(k0 + q);
            const float xs = coord(m00, m01, m02, m03, gx, gy, gz);
            const float ys = coord(m10, m11, m12, m13, gx, gy, gz);
            const float zs = coord(m20, m21, m22, m23, gx, gy, gz);
            for (int c0 = 0; c0 < nc; c0 += 8) {   // runtime channel count, chunks of <= 8
                float r[8];
                const int cc = min(8, nc - c0);
                // generic path: per-channel gathers
                const int mx = a.S - 1;
                int x0 = (int)floorf(xs), y0 = (int)floorf(ys), z0 = (int)floorf(zs);
                int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
                x0 = min(max(x0, 0), mx); x1 = min(max(x1, 0), mx);
                y0 = min(max(y0, 0), mx); y1 = min(max(y1, 0), mx);
                z0 = min(max(z0, 0), mx); z1 = min(max(z1, 0), mx);
                const float ax = __fsub_rn((float)x1, xs), bx = __fsub_rn(xs, (float)x0);
                const float ay = __fsub_rn((float)y1, ys), by = __fsub_rn(ys, (float)y0);
                const float az = __fsub_rn((float)z1, zs), bz = __fsub_rn(zs, (float)z0);
                const float w8[8] = {__fmul_rn(__fmul_rn(ax, ay), az), __fmul_rn(__fmul_rn(ax, by), az),
                                     __fmul_rn(__fmul_rn(bx, ay), az), __fmul_rn(__fmul_rn(bx, by), az),
                                     __fmul_rn(__fmul_rn(ax, ay), bz), __fmul_rn(__fmul_rn(ax, by), bz),
                                     __fmul_rn(__fmul_rn(bx, ay), bz), __fmul_rn(__fmul_rn(bx, by), bz)};
                const int S2 = a.S * a.S;
                const int i8[8] = {z0 * S2 + y0 * a.S + x0, z0 * S2 + y1 * a.S + x0, z0 * S2 + y0 * a.S + x1,
                                   z0 * S2 + y1 * a.S + x1, z1 * S2 + y0 * a.S + x0, z1 * S2 + y1 * a.S + x0,
                                   z1 * S2 + y0 * a.S + x1, z1 * S2 + y1 * a.S + x1};
                for (int c = 0; c < cc; ++c) {
                    float v = __fmul_rn(w8[0], vb[(size_t)i8[0] * nc + c0 + c]);
                    for (int e = 1; e < 8; ++e) v = __fadd_rn(v, __fmul_rn(w8[e], vb[(size_t)i8[e] * nc + c0 + c]));
                    r[c] = v;
                }
                for (int c = 0; c < cc; ++c) op[(size_t)q * nc + c0 + c] = r[c];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Two sources, one pose, one output: the face renderer resamples the geometry grid and the decoded texture volume with
// the SAME view parameters and concatenates them along the channel axis (RenderNet_Texture_Face_Normal.py:165-178;
// Reconstruct_RenderNet_Face.py:360-366: tf_rotation_resampling x2 + tf.concat).  Here the coordinates, clamped taps and
// weights of a sample are computed once and both volumes are gathered with them; the Ca + Cb channels of a sample are
// written contiguously -- the concatenated tensor is produced directly, no [B,N^3,1] / [B,N^3,4] intermediates and no
// concat copy (1 GB at batch 24).  Per channel the arithmetic is that of sample_point: bit-identical to the separate calls.
// ------------------------------------------------------------------------------------------------------------------
struct ConcatArgs {
    const float* va; const float* vb_; const float* mat; float* out;
    int B, S, N, Ca, Cb;
    int h0, w0, ph, pw, image_layout;
};

// FAST14: Ca = 1, Cb = 4 (geometry + 4-channel texture volume, the face renderer): the texture taps are 16-byte loads and
// a thread's 4 samples x 5 channels leave as five 16-byte stores.
template <bool FROM_POSE, bool FAST14>
__global__ __launch_bounds__(256)
void resample_concat_kernel(const ConcatArgs a)
{
    __shared__ float msh[12];
    const int N = a.N, Ct = a.Ca + a.Cb;
    const int kthreads = N / 4, lines_per_block = 256 / kthreads;
    const long long nlines = (long long)a.B * a.ph * a.pw;
    const long long line0 = (long long)blockIdx.x * lines_per_block;
    const int b = (int)(line0 / ((long long)a.ph * a.pw));
    if (threadIdx.x == 0) {
        if (FROM_POSE) pose_to_affine_dev(a.mat + 3 * b, a.S, N, msh);
        else for (int q = 0; q < 12; ++q) msh[q] = a.mat[12 * b + q];
    }
    __syncthreads();
    const long long line = line0 + threadIdx.x / kthreads;
    if (line >= nlines) return;
    const int k0 = (threadIdx.x % kthreads) * 4;
    const int ij = (int)(line - (long long)b * a.ph * a.pw);
    const int i = ij / a.pw + a.h0, j = ij % a.pw + a.w0;
    const float gy = a.image_layout ? (float)(N - 1 - i) : (float)j;
    const float gz = a.image_layout ? (float)j : (float)i;
    const size_t S3 = (size_t)a.S * a.S * a.S;
    const float* pa = a.va + (size_t)b * S3 * a.Ca;
    const float* pb = a.vb_ + (size_t)b * S3 * a.Cb;
    float* op = a.out + ((size_t)line * N + k0) * Ct;
    float r20[20];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float gx = (float)(k0 + q);
        const float x = coord(msh[0], msh[1], msh[2], msh[3], gx, gy, gz);
        const float y = coord(msh[4], msh[5], msh[6], msh[7], gx, gy, gz);
        const float z = coord(msh[8], msh[9], msh[10], msh[11], gx, gy, gz);
        const int mx = a.S - 1;
        int x0 = (int)floorf(x), y0 = (int)floorf(y), z0 = (int)floorf(z);
        int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
        x0 = min(max(x0, 0), mx); x1 = min(max(x1, 0), mx);
        y0 = min(max(y0, 0), mx); y1 = min(max(y1, 0), mx);
        z0 = min(max(z0, 0), mx); z1 = min(max(z1, 0), mx);
        const float ax = __fsub_rn((float)x1, x), bx = __fsub_rn(x, (float)x0);
        const float ay = __fsub_rn((float)y1, y), by = __fsub_rn(y, (float)y0);
        const float az = __fsub_rn((float)z1, z), bz = __fsub_rn(z, (float)z0);
        const float w8[8] = {__fmul_rn(__fmul_rn(ax, ay), az), __fmul_rn(__fmul_rn(ax, by), az),
                             __fmul_rn(__fmul_rn(bx, ay), az), __fmul_rn(__fmul_rn(bx, by), az),
                             __fmul_rn(__fmul_rn(ax, ay), bz), __fmul_rn(__fmul_rn(ax, by), bz),
                             __fmul_rn(__fmul_rn(bx, ay), bz), __fmul_rn(__fmul_rn(bx, by), bz)};
        const int S2 = a.S * a.S;
        const int i8[8] = {z0 * S2 + y0 * a.S + x0, z0 * S2 + y1 * a.S + x0, z0 * S2 + y0 * a.S + x1, z0 * S2 + y1 * a.S + x1,
                           z1 * S2 + y0 * a.S + x0, z1 * S2 + y1 * a.S + x0, z1 * S2 + y0 * a.S + x1, z1 * S2 + y1 * a.S + x1};
        // a sample outside the source along an axis has coinciding clamped taps and cancelling weights, but the add_n order
        // leaves rounding residues: the arithmetic is done as written in every case
        if (FAST14) {
            float ta[8];
            float4 tb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ta[e] = pa[i8[e]];
                tb[e] = *reinterpret_cast<const float4*>(pb + (size_t)i8[e] * 4);
            }
            float v0 = __fmul_rn(w8[0], ta[0]), v1 = __fmul_rn(w8[0], tb[0].x), v2 = __fmul_rn(w8[0], tb[0].y);
            float v3 = __fmul_rn(w8[0], tb[0].z), v4 = __fmul_rn(w8[0], tb[0].w);
#pragma unroll
            for (int e = 1; e < 8; ++e) {
                v0 = __fadd_rn(v0, __fmul_rn(w8[e], ta[e]));
                v1 = __fadd_rn(v1, __fmul_rn(w8[e], tb[e].x));
                v2 = __fadd_rn(v2, __fmul_rn(w8[e], tb[e].y));
                v3 = __fadd_rn(v3, __fmul_rn(w8[e], tb[e].z));
                v4 = __fadd_rn(v4, __fmul_rn(w8[e], tb[e].w));
            }
            r20[q * 5 + 0] = v0; r20[q * 5 + 1] = v1; r20[q * 5 + 2] = v2; r20[q * 5 + 3] = v3; r20[q * 5 + 4] = v4;
            continue;
        }
        for (int c = 0; c < Ct; ++c) {
            const float* src = c < a.Ca ? pa + c : pb + (c - a.Ca);
            const int cs = c < a.Ca ? a.Ca : a.Cb;
            float v = __fmul_rn(w8[0], src[(size_t)i8[0] * cs]);
#pragma unroll
            for (int e = 1; e < 8; ++e) v = __fadd_rn(v, __fmul_rn(w8[e], src[(size_t)i8[e] * cs]));
            op[(size_t)q * Ct + c] = v;
        }
    }
    if (FAST14) {
#pragma unroll
        for (int q = 0; q < 5; ++q)
            reinterpret_cast<float4*>(op)[q] = make_float4(r20[4 * q], r20[4 * q + 1], r20[4 * q + 2], r20[4 * q + 3]);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Brick form of the Ca = 1, Cb = 4 case for DENSE volumes (the decoded texture volume is non-zero everywhere, so nothing can
// be culled the way resample_tiled.hip culls empty space): a workgroup owns an 8 x 8 x 8 brick of output samples.  Its
// source footprint is a rotated box of about 4 x 4 x 4 voxels (the 128^3 grid samples the 64^3 volume at half-voxel steps);
// a box containing all its taps is staged in LDS once (<= 10^3 voxels x 20 B) with row-contiguous
// loads, the 8 x 5 taps of every sample are then LDS reads instead of 16 divergent global gathers, and the results leave
// through LDS as whole 160-byte line segments in 16-byte stores.  Same arithmetic per sample as resample_concat_kernel:
// bit-identical.  A brick whose box does not fit (an extreme scale through the affine entry) gathers from global memory.
// Measured (B=24, 64^3 -> 128^3, 1+4 channels): see DESIGN.md section 4.
// ------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int BRK = 8, BRK_MAXVOX = 1000;          // 10^3 voxels: 30 KB of LDS with the output stage -> five workgroups per CU
}

template <bool FROM_POSE>
__global__ __launch_bounds__(256)
void resample_concat_brick_kernel(const ConcatArgs a)
{
    __shared__ float msh_[12];
    __shared__ float ga[BRK_MAXVOX];
    __shared__ __attribute__((aligned(16))) float gb[BRK_MAXVOX * 4];
    __shared__ __attribute__((aligned(16))) float ob[BRK * BRK * BRK * 5];
    const int N = a.N, tid = threadIdx.x;
    const int nbk = N / BRK, nbj = a.pw / BRK, nbi = a.ph / BRK;
    int blk = blockIdx.x;
    const int bk = blk % nbk; blk /= nbk;
    const int bj = blk % nbj; blk /= nbj;
    const int bi = blk % nbi;
    const int b = blk / nbi;
    // the matrix: with the affine entry 12 wave-uniform loads (no barrier); from a pose one lane's double-precision closed form
    // (rendernet_amd.ops converts poses with rn_pose_to_affine first: one small launch instead of a ~2 us chain per brick)
    float msh[12];
    if (FROM_POSE) {
        if (tid == 0) pose_to_affine_dev(a.mat + 3 * b, a.S, N, msh_);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 12; ++q) msh[q] = msh_[q];
    } else {
#pragma unroll
        for (int q = 0; q < 12; ++q) msh[q] = a.mat[12 * b + q];
    }
    const int mx = a.S - 1;
    // the brick's sample (li, lj, lk) has grid coordinates gx = k, gy / gz from (i, j) as in resample_concat_kernel
    auto grid_of = [&](int li, int lj, int lk, float& gx, float& gy, float& gz) {
        const int i = bi * BRK + li + a.h0, j = bj * BRK + lj + a.w0;
        gx = (float)(bk * BRK + lk);
        gy = a.image_layout ? (float)(N - 1 - i) : (float)j;
        gz = a.image_layout ? (float)j : (float)i;
    };
    // A box that CONTAINS all taps (a superset only stages a few more voxels): the coordinate is affine in the brick's local
    // (li, lj, lk), so from the exact coordinate of sample (0, 0, 0) the extremes over the brick are at most the sums of the
    // negative / positive axis contributions 7 * |m| away; 0.02 of slack covers the rounding of the per-sample chains (their
    // error is ~1e-5 at these magnitudes).  ~40 instructions per thread instead of eight exact corner evaluations.
    int blo[3], bhi[3];
    {
        float gx, gy, gz, hx, hy, hz;
        grid_of(0, 0, 0, gx, gy, gz);
        grid_of(BRK - 1, BRK - 1, BRK - 1, hx, hy, hz);
        const float dx = hx - gx, dy = hy - gy, dz = hz - gz;          // +-7 along each grid axis
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float c0 = coord(msh[4 * r], msh[4 * r + 1], msh[4 * r + 2], msh[4 * r + 3], gx, gy, gz);
            const float ex = msh[4 * r] * dx, ey = msh[4 * r + 1] * dy, ez = msh[4 * r + 2] * dz;
            const float lo = c0 + fminf(ex, 0.f) + fminf(ey, 0.f) + fminf(ez, 0.f) - 0.02f;
            const float hi = c0 + fmaxf(ex, 0.f) + fmaxf(ey, 0.f) + fmaxf(ez, 0.f) + 0.02f;
            blo[r] = min(max((int)floorf(fminf(fmaxf(lo, -4.0f), (float)a.S + 4.0f)), 0), mx);
            bhi[r] = min(max((int)floorf(fminf(fmaxf(hi, -4.0f), (float)a.S + 4.0f)) + 1, 0), mx);
        }
    }
    const int bx0 = blo[0], by0 = blo[1], bz0 = blo[2], nx = bhi[0] - blo[0] + 1, ny = bhi[1] - blo[1] + 1, nz = bhi[2] - blo[2] + 1;
    const bool fits = nx * ny * nz <= BRK_MAXVOX;
    const size_t S3 = (size_t)a.S * a.S * a.S;
    const float* pa = a.va + (size_t)b * S3;
    const float* pb = a.vb_ + (size_t)b * S3 * 4;
    if (fits) {
        const int nvox = nx * ny * nz;
        for (int v = tid; v < nvox; v += 256) {
            const int xx = v % nx, r = v / nx;
            const int yy = r % ny, zz = r / ny;
            const size_t si = ((size_t)(bz0 + zz) * a.S + (by0 + yy)) * a.S + (bx0 + xx);
            ga[v] = pa[si];
            *reinterpret_cast<float4*>(gb + 4 * v) = *reinterpret_cast<const float4*>(pb + si * 4);
        }
    }
    __syncthreads();
    const int line = tid >> 2, li = line >> 3, lj = line & 7;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int lk = (tid & 3) * 2 + q;
        float gx, gy, gz;
        grid_of(li, lj, lk, gx, gy, gz);
        const float x = coord(msh[0], msh[1], msh[2], msh[3], gx, gy, gz);
        const float y = coord(msh[4], msh[5], msh[6], msh[7], gx, gy, gz);
        const float z = coord(msh[8], msh[9], msh[10], msh[11], gx, gy, gz);
        int x0 = (int)floorf(x), y0 = (int)floorf(y), z0 = (int)floorf(z);
        int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
        x0 = min(max(x0, 0), mx); x1 = min(max(x1, 0), mx);
        y0 = min(max(y0, 0), mx); y1 = min(max(y1, 0), mx);
        z0 = min(max(z0, 0), mx); z1 = min(max(z1, 0), mx);
        const float ax = __fsub_rn((float)x1, x), bx = __fsub_rn(x, (float)x0);
        const float ay = __fsub_rn((float)y1, y), by = __fsub_rn(y, (float)y0);
        const float az = __fsub_rn((float)z1, z), bz = __fsub_rn(z, (float)z0);
        const float w8[8] = {__fmul_rn(__fmul_rn(ax, ay), az), __fmul_rn(__fmul_rn(ax, by), az),
                             __fmul_rn(__fmul_rn(bx, ay), az), __fmul_rn(__fmul_rn(bx, by), az),
                             __fmul_rn(__fmul_rn(ax, ay), bz), __fmul_rn(__fmul_rn(ax, by), bz),
                             __fmul_rn(__fmul_rn(bx, ay), bz), __fmul_rn(__fmul_rn(bx, by), bz)};
        float ta[8];
        float4 tb[8];
        if (fits) {
            const int sy = nx, sz = nx * ny;
            const int X0 = x0 - bx0, X1 = x1 - bx0, Y0 = (y0 - by0) * sy, Y1 = (y1 - by0) * sy, Z0 = (z0 - bz0) * sz, Z1 = (z1 - bz0) * sz;
            const int i8[8] = {Z0 + Y0 + X0, Z0 + Y1 + X0, Z0 + Y0 + X1, Z0 + Y1 + X1, Z1 + Y0 + X0, Z1 + Y1 + X0, Z1 + Y0 + X1, Z1 + Y1 + X1};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ta[e] = ga[i8[e]];
                tb[e] = *reinterpret_cast<const float4*>(gb + 4 * i8[e]);
            }
        } else {
            const int S2 = a.S * a.S;
            const int i8[8] = {z0 * S2 + y0 * a.S + x0, z0 * S2 + y1 * a.S + x0, z0 * S2 + y0 * a.S + x1, z0 * S2 + y1 * a.S + x1,
                               z1 * S2 + y0 * a.S + x0, z1 * S2 + y1 * a.S + x0, z1 * S2 + y0 * a.S + x1, z1 * S2 + y1 * a.S + x1};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ta[e] = pa[i8[e]];
                tb[e] = *reinterpret_cast<const float4*>(pb + (size_t)i8[e] * 4);
            }
        }
        float v0 = __fmul_rn(w8[0], ta[0]), v1 = __fmul_rn(w8[0], tb[0].x), v2 = __fmul_rn(w8[0], tb[0].y);
        float v3 = __fmul_rn(w8[0], tb[0].z), v4 = __fmul_rn(w8[0], tb[0].w);
#pragma unroll
        for (int e = 1; e < 8; ++e) {
            v0 = __fadd_rn(v0, __fmul_rn(w8[e], ta[e]));
            v1 = __fadd_rn(v1, __fmul_rn(w8[e], tb[e].x));
            v2 = __fadd_rn(v2, __fmul_rn(w8[e], tb[e].y));
            v3 = __fadd_rn(v3, __fmul_rn(w8[e], tb[e].z));
            v4 = __fadd_rn(v4, __fmul_rn(w8[e], tb[e].w));
        }
        float* o = ob + (line * BRK + lk) * 5;
        o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3; o[4] = v4;
    }
    __syncthreads();
    // 64 lines x 40 floats: whole 160-byte segments, 16-byte stores
    for (int t = tid; t < BRK * BRK * 10; t += 256) {
        const int ln = t / 10, f = t - ln * 10;
        const int i = bi * BRK + (ln >> 3), j = bj * BRK + (ln & 7);
        float* op = a.out + ((((size_t)b * a.ph + i) * a.pw + j) * N + (size_t)bk * BRK) * 5;
        reinterpret_cast<float4*>(op)[f] = *reinterpret_cast<const float4*>(ob + ln * 40 + f * 4);
    }
}

extern "C" int rn_resample_concat_fwd(const float* vox_a, int Ca, const float* vox_b, int Cb, const float* pose_or_m_inv,
                                      int affine, float* out, int B, int S, int N, int h0, int w0, int ph, int pw,
                                      int image_layout, void* stream)
{
    if (!vox_a || !vox_b || !pose_or_m_inv || !out) return rn_set_error(RN_E_INVALID, "rn_resample_concat_fwd: null pointer");
    if (B <= 0 || S < 2 || N < 4 || Ca < 1 || Cb < 1) return rn_set_error(RN_E_INVALID, "rn_resample_concat_fwd: bad dims");
    if (N % 4 != 0 || N / 4 > 256 || 256 % (N / 4) != 0)
        return rn_set_error(RN_E_INVALID, "rn_resample_concat_fwd: N=%d must be a power of two in [4,1024]", N);
    if (h0 < 0 || w0 < 0 || ph < 1 || pw < 1 || h0 + ph > N || w0 + pw > N)
        return rn_set_error(RN_E_INVALID, "rn_resample_concat_fwd: crop window out of range");
    if (!image_layout && (h0 || w0 || ph != N || pw != N))
        return rn_set_error(RN_E_INVALID, "rn_resample_concat_fwd: crop needs image_layout=1");
    const int lpb = 256 / (N / 4);
    if (((long long)ph * pw) % lpb != 0)
        return rn_set_error(RN_E_INVALID, "rn_resample_concat_fwd: ph*pw=%d must be a multiple of %d", ph * pw, lpb);
    const long long nb = (long long)B * ph * pw / lpb;
    if (nb > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "rn_resample_concat_fwd: grid too large");
    ConcatArgs a{vox_a, vox_b, pose_or_m_inv, out, B, S, N, Ca, Cb, h0, w0, ph, pw, image_layout};
    const bool fast = Ca == 1 && Cb == 4 && (((uintptr_t)vox_b | (uintptr_t)out) & 15) == 0;
    static const bool no_brick = getenv("RN_RESAMPLE_NO_BRICK") != nullptr;
    if (fast && !no_brick && N % BRK == 0 && ph % BRK == 0 && pw % BRK == 0) {
        const long long nbr = (long long)B * (ph / BRK) * (pw / BRK) * (N / BRK);
        if (nbr <= 0x7fffffffLL) {
            if (affine) hipLaunchKernelGGL(resample_concat_brick_kernel<false>, dim3((unsigned)nbr), dim3(256), 0, (hipStream_t)stream, a);
            else hipLaunchKernelGGL(resample_concat_brick_kernel<true>, dim3((unsigned)nbr), dim3(256), 0, (hipStream_t)stream, a);
            return rn_check_launch("rn_resample_concat_fwd (brick)");
        }
    }
    if (affine && fast) hipLaunchKernelGGL((resample_concat_kernel<false, true>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a);
    else if (affine) hipLaunchKernelGGL((resample_concat_kernel<false, false>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a);
    else if (fast) hipLaunchKernelGGL((resample_concat_kernel<true, true>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((resample_concat_kernel<true, false>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a);
    return rn_check_launch("rn_resample_concat_fwd");
}

template <bool FROM_POSE>
static int launch_resample(const ResampleArgs& a, hipStream_t st)
{
    if (a.B <= 0 || a.S < 2 || a.N < 4 || a.C < 1) return rn_set_error(RN_E_INVALID, "resample: bad dims");
    if (a.N % 4 != 0 || a.N / 4 > 256 || 256 % (a.N / 4) != 0)
        return rn_set_error(RN_E_INVALID, "resample: N=%d must be a power of two in [4,1024]", a.N);
    if (a.h0 < 0 || a.w0 < 0 || a.ph < 1 || a.pw < 1 || a.h0 + a.ph > a.N || a.w0 + a.pw > a.N)
        return rn_set_error(RN_E_INVALID, "resample: crop window out of range");
    if (!a.image_layout && (a.h0 || a.w0 || a.ph != a.N || a.pw != a.N))
        return rn_set_error(RN_E_INVALID, "resample: crop needs image_layout=1");
    const int lpb = 256 / (a.N / 4);
    if (((long long)a.ph * a.pw) % lpb != 0)
        return rn_set_error(RN_E_INVALID, "resample: ph*pw=%d must be a multiple of %d", a.ph * a.pw, lpb);
    const long long nb = (long long)a.B * a.ph * a.pw / lpb;
    if (nb > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "resample: grid too large");
    if (a.C == 1) hipLaunchKernelGGL((resample_kernel<1, FROM_POSE>), dim3((unsigned)nb), dim3(256), 0, st, a);
    else if (a.C == 4) hipLaunchKernelGGL((resample_kernel<4, FROM_POSE>), dim3((unsigned)nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((resample_kernel<0, FROM_POSE>), dim3((unsigned)nb), dim3(256), 0, st, a);
    return rn_check_launch("resample");
}

// tiled production path (resample_tiled.hip)
bool rn_resample_tiled_supported(int B, int S, int N, int C, int ph, int pw);
size_t rn_resample_tiled_workspace(int B, int S);
int rn_launch_resample_tiled(const float* vox, const float* mat_or_pose, bool from_pose, float* out,
                             int B, int S, int N, int C, int h0, int w0, int ph, int pw, int image_layout,
                             void* workspace, hipStream_t st);

template <bool FROM_POSE>
static int resample_entry(const float* vox, const float* mat, float* out, int B, int S, int N, int C,
                          int h0, int w0, int ph, int pw, int image_layout, void* workspace,
                          size_t workspace_bytes, void* stream, const char* who)
{
    if (!vox || !mat || !out) return rn_set_error(RN_E_INVALID, "%s: null pointer", who);
    ResampleArgs a{vox, mat, out, B, S, N, C, h0, w0, ph, pw, image_layout};
    static const bool force_simple = getenv("RN_RESAMPLE_SIMPLE") != nullptr;
    if (!force_simple && workspace && B > 0 && rn_resample_tiled_supported(B, S, N, C, ph, pw) &&
        workspace_bytes >= rn_resample_tiled_workspace(B, S)) {
        // same argument checks as the simple path
        if (h0 < 0 || w0 < 0 || ph < 1 || pw < 1 || h0 + ph > N || w0 + pw > N)
            return rn_set_error(RN_E_INVALID, "resample: crop window out of range");
        if (!image_layout && (h0 || w0 || ph != N || pw != N))
            return rn_set_error(RN_E_INVALID, "resample: crop needs image_layout=1");
        return rn_launch_resample_tiled(vox, mat, FROM_POSE, out, B, S, N, C, h0, w0, ph, pw, image_layout,
                                        workspace, (hipStream_t)stream);
    }
    return launch_resample<FROM_POSE>(a, (hipStream_t)stream);
}

extern "C" size_t rn_resample_workspace_bytes(int B, int S, int C)
{
    (void)C;
    if (B < 1 || S < 4 || S % 4) return 0;
    return rn_resample_tiled_workspace(B, S);
}

extern "C" int rn_resample_fwd(const float* vox, const float* pose, float* out, int B, int S, int N, int C,
                               int h0, int w0, int ph, int pw, int image_layout,
                               void* workspace, size_t workspace_bytes, void* stream)
{
    return resample_entry<true>(vox, pose, out, B, S, N, C, h0, w0, ph, pw, image_layout, workspace,
                                workspace_bytes, stream, "rn_resample_fwd");
}

extern "C" int rn_resample_affine_fwd(const float* vox, const float* m_inv, float* out, int B, int S, int N, int C,
                                      int h0, int w0, int ph, int pw, int image_layout,
                                      void* workspace, size_t workspace_bytes, void* stream)
{
    return resample_entry<false>(vox, m_inv, out, B, S, N, C, h0, w0, ph, pw, image_layout, workspace,
                                 workspace_bytes, stream, "rn_resample_affine_fwd");
}

extern "C" int rn_pose_to_affine(const float* pose, float* m_inv, int B, int S, int N, void* stream)
{
    if (!pose || !m_inv || B <= 0) return rn_set_error(RN_E_INVALID, "rn_pose_to_affine: bad argument");
    hipLaunchKernelGGL(pose_to_affine_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, pose, m_inv, B, S, N);
    return rn_check_launch("pose_to_affine");
}
