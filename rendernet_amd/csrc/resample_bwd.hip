// Backward of the resampler -- what makes the renderer differentiable w.r.t. the voxel grid and the pose
// (the inverse-rendering use of the reference: Reconstruct_RenderNet_Face.py:360-364, :402 optimises shape and pose
// through tf_rotation_resampling).  These are the gradients TensorFlow's autodiff derives from
// tools/resampling_voxel_grid.py:381-486: floor / clip carry no gradient, the eight tf.gather's scatter
// `weight * dout` back into the voxel grid, and the weights (x1c - x), (x - x0c), ... are linear in the coordinates.
//
//   d out / d vox :  dvox[tap] += w_tap * dout            (8 atomics per sample and channel)
//   d out / d x   :  sum_c dout_c * [ ay*az*(c-a) + by*az*(d-b) + ay*bz*(g-e) + by*bz*(h-f) ]   (likewise y, z), and
//   d L / d M_inv[r][:] = sum_samples (dL/dcoord_r) * (gx, gy, gz, 1)   since coord = M_inv * (gx, gy, gz, 1)
//   d L / d pose  =  J^T dL/dM_inv with J the Jacobian of the closed-form M_inv(azimuth, elevation, scale).
//
// A sample with an axis whose two clamped indices coincide (x0c == x1c: it lies outside the volume along that axis)
// contributes +w and -w to the SAME voxels (ax = -bx exactly), i.e. nothing but rounding noise, and its coordinate
// derivatives vanish identically; those samples -- 7/8 of the grid at scale 1 -- are skipped, which also removes the
// atomic pile-up on the border voxels.
#include "rn_common.h"
#include <math.h>

#pragma clang fp contract(off)

struct ResampleBwdArgs {
    const float* vox;      // [B,S,S,S,C] (needed for dm only)
    const float* m_inv;    // [B,12]
    const float* dout;     // image_layout=1: [B,ph,pw,N,C] window; else [B,N,N,N,C]
    float* dvox;           // [B,S,S,S,C] accumulated, or null
    float* dm;             // [B,12] accumulated, or null
    int B, S, N, C;
    int h0, w0, ph, pw, image_layout;
    int dcs, dco;          // dout holds dcs channels per sample; this source's C channels start at channel dco
};

__device__ __forceinline__ float coord_b(float m0, float m1, float m2, float m3, float x, float y, float z)
{
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m0, x), __fmul_rn(m1, y)), __fmul_rn(m2, z)), m3);
}

// one thread per output sample (all channels); a block never straddles two batch items
__global__ __launch_bounds__(256)
void resample_bwd_kernel(const ResampleBwdArgs a)
{
    __shared__ float red[4][12];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long per_item = (long long)a.ph * a.pw * a.N;
    const long long blocks_per_item = (per_item + 255) / 256;
    const int b = (int)(blockIdx.x / blocks_per_item);
    const long long e = (blockIdx.x % blocks_per_item) * 256 + tid;
    float acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.f;
    if (e < per_item) {
        const int k = (int)(e % a.N), jl = (int)((e / a.N) % a.pw), il = (int)(e / ((long long)a.N * a.pw));
        const int i = a.h0 + il, j = a.w0 + jl;
        const float gx = (float)k;
        const float gy = a.image_layout ? (float)(a.N - 1 - i) : (float)j;
        const float gz = a.image_layout ? (float)j : (float)i;
        const float* m = a.m_inv + 12 * b;
        const float x = coord_b(m[0], m[1], m[2], m[3], gx, gy, gz);
        const float y = coord_b(m[4], m[5], m[6], m[7], gx, gy, gz);
        const float z = coord_b(m[8], m[9], m[10], m[11], gx, gy, gz);
        const int mx = a.S - 1;
        int x0 = (int)floorf(x), y0 = (int)floorf(y), z0 = (int)floorf(z);
        int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
        x0 = min(max(x0, 0), mx); x1 = min(max(x1, 0), mx);
        y0 = min(max(y0, 0), mx); y1 = min(max(y1, 0), mx);
        z0 = min(max(z0, 0), mx); z1 = min(max(z1, 0), mx);
        if (x0 != x1 && y0 != y1 && z0 != z1) {
            const float ax = (float)x1 - x, bx = x - (float)x0;
            const float ay = (float)y1 - y, by = y - (float)y0;
            const float az = (float)z1 - z, bz = z - (float)z0;
            const float* dp = a.dout + ((size_t)b * per_item + e) * a.dcs + a.dco;
            const size_t S = a.S;
            const size_t base = (size_t)b * S * S * S;
            const size_t ia = base + ((size_t)z0 * S + y0) * S + x0, ib = base + ((size_t)z0 * S + y1) * S + x0;
            const size_t ic = base + ((size_t)z0 * S + y0) * S + x1, id = base + ((size_t)z0 * S + y1) * S + x1;
            const size_t ie = base + ((size_t)z1 * S + y0) * S + x0, if_ = base + ((size_t)z1 * S + y1) * S + x0;
            const size_t ig = base + ((size_t)z1 * S + y0) * S + x1, ih = base + ((size_t)z1 * S + y1) * S + x1;
            float gxs = 0.f, gys = 0.f, gzs = 0.f;
            for (int c = 0; c < a.C; ++c) {
                const float d = dp[c];
                if (d == 0.f) continue;
                if (a.dvox) {
                    unsafeAtomicAdd(a.dvox + ia * a.C + c, ax * ay * az * d);
                    unsafeAtomicAdd(a.dvox + ib * a.C + c, ax * by * az * d);
                    unsafeAtomicAdd(a.dvox + ic * a.C + c, bx * ay * az * d);
                    unsafeAtomicAdd(a.dvox + id * a.C + c, bx * by * az * d);
                    unsafeAtomicAdd(a.dvox + ie * a.C + c, ax * ay * bz * d);
                    unsafeAtomicAdd(a.dvox + if_ * a.C + c, ax * by * bz * d);
                    unsafeAtomicAdd(a.dvox + ig * a.C + c, bx * ay * bz * d);
                    unsafeAtomicAdd(a.dvox + ih * a.C + c, bx * by * bz * d);
                }
                if (a.dm) {
                    const float va = a.vox[ia * a.C + c], vb = a.vox[ib * a.C + c], vc = a.vox[ic * a.C + c], vd = a.vox[id * a.C + c];
                    const float ve = a.vox[ie * a.C + c], vf = a.vox[if_ * a.C + c], vg = a.vox[ig * a.C + c], vh = a.vox[ih * a.C + c];
                    gxs += d * (ay * az * (vc - va) + by * az * (vd - vb) + ay * bz * (vg - ve) + by * bz * (vh - vf));
                    gys += d * (ax * az * (vb - va) + bx * az * (vd - vc) + ax * bz * (vf - ve) + bx * bz * (vh - vg));
                    gzs += d * (ax * ay * (ve - va) + ax * by * (vf - vb) + bx * ay * (vg - vc) + bx * by * (vh - vd));
                }
            }
            const float gg[3] = {gxs, gys, gzs};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                acc[4 * r + 0] = gg[r] * gx; acc[4 * r + 1] = gg[r] * gy; acc[4 * r + 2] = gg[r] * gz; acc[4 * r + 3] = gg[r];
            }
        }
    }
    if (a.dm) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            float v = acc[q];
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
            if (lane == 0) red[wave][q] = v;
        }
        __syncthreads();
        if (tid < 12) {
            const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
            if (v != 0.f) unsafeAtomicAdd(a.dm + 12 * b + tid, v);
        }
    }
}

// dpose[b][0..2] += J^T dm[b], J = d(M_inv)/d(azimuth, elevation, scale) of the closed form in resample.hip /
// resample_tiled.hip (pose_to_affine): a_rc = rt[r][c]/s, t_r = S/2 - (a_r0 + a_r1 + a_r2) * N/2
__global__ void pose_to_affine_bwd_kernel(const float* __restrict__ pose, const float* __restrict__ dm,
                                          float* __restrict__ dpose, int B, int S, int N)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double az = (double)pose[3 * b] - 1.5707963267948966, el = (double)pose[3 * b + 1], s = (double)pose[3 * b + 2];
    const double is = 1.0 / s, hn = 0.5 * N;
    const double ca = cos(az), sa = sin(az), ce = cos(el), se = sin(el);
    const double rt[3][3] = {{ce * ca, -se * ca, sa}, {se, ce, 0.0}, {-ce * sa, se * sa, ca}};
    const double d_az[3][3] = {{-ce * sa, se * sa, ca}, {0.0, 0.0, 0.0}, {-ce * ca, se * ca, -sa}};
    const double d_el[3][3] = {{-se * ca, -ce * ca, 0.0}, {ce, -se, 0.0}, {se * sa, ce * sa, 0.0}};
    double g[3] = {0.0, 0.0, 0.0};
    (void)S;
    for (int r = 0; r < 3; ++r) {
        const double dt = (double)dm[12 * b + 4 * r + 3];
        for (int c = 0; c < 3; ++c) {
            // d(loss)/d(a_rc) including the path through t_r
            const double da = (double)dm[12 * b + 4 * r + c] - hn * dt;
            g[0] += da * d_az[r][c] * is;
            g[1] += da * d_el[r][c] * is;
            g[2] += da * (-rt[r][c] * is * is);
        }
    }
    dpose[3 * b + 0] += (float)g[0];
    dpose[3 * b + 1] += (float)g[1];
    dpose[3 * b + 2] += (float)g[2];
}

static int resample_bwd_impl(const float* vox, const float* m_inv, const float* dout, int dcs, int dco, float* dvox, float* dm,
                             int B, int S, int N, int C, int h0, int w0, int ph, int pw, int image_layout, void* stream);

extern "C" int rn_resample_affine_bwd(const float* vox, const float* m_inv, const float* dout, float* dvox, float* dm,
                                      int B, int S, int N, int C, int h0, int w0, int ph, int pw, int image_layout,
                                      void* stream)
{
    return resample_bwd_impl(vox, m_inv, dout, C, 0, dvox, dm, B, S, N, C, h0, w0, ph, pw, image_layout, stream);
}

// The same for ONE source of rn_resample_concat_fwd: dout holds dout_channels per sample, this source's C channels start
// at channel dout_offset (call once per source; dm accumulates over the calls).
extern "C" int rn_resample_affine_bwd_strided(const float* vox, const float* m_inv, const float* dout, int dout_channels,
                                              int dout_offset, float* dvox, float* dm, int B, int S, int N, int C,
                                              int h0, int w0, int ph, int pw, int image_layout, void* stream)
{
    if (dout_channels < 1 || dout_offset < 0 || dout_offset + C > dout_channels)
        return rn_set_error(RN_E_INVALID, "rn_resample_affine_bwd_strided: channels [%d, %d) outside %d", dout_offset, dout_offset + C, dout_channels);
    return resample_bwd_impl(vox, m_inv, dout, dout_channels, dout_offset, dvox, dm, B, S, N, C, h0, w0, ph, pw, image_layout, stream);
}

static int resample_bwd_impl(const float* vox, const float* m_inv, const float* dout, int dcs, int dco, float* dvox, float* dm,
                             int B, int S, int N, int C, int h0, int w0, int ph, int pw, int image_layout, void* stream)
{
    if (!m_inv || !dout || (!dvox && !dm)) return rn_set_error(RN_E_INVALID, "rn_resample_affine_bwd: null pointer");
    if (dm && !vox) return rn_set_error(RN_E_INVALID, "rn_resample_affine_bwd: the matrix gradient needs vox");
    if (B < 1 || S < 2 || N < 1 || C < 1) return rn_set_error(RN_E_INVALID, "rn_resample_affine_bwd: bad dims");
    if (h0 < 0 || w0 < 0 || ph < 1 || pw < 1 || h0 + ph > N || w0 + pw > N)
        return rn_set_error(RN_E_INVALID, "rn_resample_affine_bwd: crop window out of range");
    if (!image_layout && (h0 || w0 || ph != N || pw != N))
        return rn_set_error(RN_E_INVALID, "rn_resample_affine_bwd: crop needs image_layout=1");
    ResampleBwdArgs a{vox, m_inv, dout, dvox, dm, B, S, N, C, h0, w0, ph, pw, image_layout, dcs, dco};
    const long long per_item = (long long)ph * pw * N;
    const long long nb = (per_item + 255) / 256 * B;
    if (nb > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "rn_resample_affine_bwd: grid too large");
    hipLaunchKernelGGL(resample_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a);
    return rn_check_launch("rn_resample_affine_bwd");
}

extern "C" int rn_pose_to_affine_bwd(const float* pose, const float* dm, float* dpose, int B, int S, int N, void* stream)
{
    if (!pose || !dm || !dpose || B < 1) return rn_set_error(RN_E_INVALID, "rn_pose_to_affine_bwd: bad argument");
    hipLaunchKernelGGL(pose_to_affine_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, pose, dm, dpose, B, S, N);
    return rn_check_launch("rn_pose_to_affine_bwd");
}
