// Memory-bound kernels of the training step (RenderNet_Shader.py:159-167): backward of the fused conv
// epilogue (bias / PReLU / sigmoid), the reconstruction loss and its gradient, TF's Adam update, and
// the input gradient of the channel-starved strided stem conv (e_conv2) that has no MFMA shape.
#include "rn_common.h"
#include <math.h>
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
// Backward of  y = sigmoid?( prelu?(z) + residual ),  z = conv + bias   (tools/layer_util.py:27-45,
// :133-144; the residual's gradient is dy itself -- after the sigmoid factor -- and needs no kernel):
//     dt = dy * y*(1-y)            if sigmoid   (needs y)
//     dz = dt * (z > 0 ? 1 : alpha[c]),  dalpha[c] += sum dt * min(z, 0)      if PReLU  (needs z)
//     dz = dy * (y < 0 ? y + 1 : 1)                                            if ELU    (needs y; TF's EluGrad)
//     dbias[c] += sum dz
// Rows are [M, C] channels-last.  dz may alias dy.  dbias/dalpha are accumulated with atomics.
// ---------------------------------------------------------------------------------------------
struct EpiBwdArgs {
    const float* dy; const float* z; const float* y; const float* alpha;
    float* dz; float* dbias; float* dalpha;
    long long M; int C; int act; int rows_per_block;
    float* ws;                                     // null: atomics onto dbias / dalpha; else [row block][2][C] partial sums (rn_epilogue_bwd_ws)
};

// fast path: C % 4 == 0 and (256 % (C/4) == 0 or (C/4) % 256 == 0): a thread owns one float4 channel
// group for its whole life, so its partial sums stay in registers
// NT threads per workgroup (256 | 1024): the row-block count is capped (see the launcher), so on large tensors the bytes in flight come from
// wider workgroups -- 1024 threads = four row lanes per channel group at C = 1024
template <int NT>
__global__ __launch_bounds__(NT)
void epilogue_bwd_vec_kernel(const EpiBwdArgs a)
{
    __shared__ float red[2][NT * 4];
    const int G = a.C >> 2;                         // float4 groups per row
    const int gper = G < 256 ? G : 256;             // groups handled by one block column
    const int g = blockIdx.y * 256 + (threadIdx.x % gper);
    const int rsub = threadIdx.x / gper, rstep = NT / gper;
    const long long r0 = (long long)blockIdx.x * a.rows_per_block;
    const long long r1 = r0 + a.rows_per_block < a.M ? r0 + a.rows_per_block : a.M;
    float4 al = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.act & RN_ACT_PRELU) al = reinterpret_cast<const float4*>(a.alpha)[g];
    float sb[4] = {0.f, 0.f, 0.f, 0.f}, sa[4] = {0.f, 0.f, 0.f, 0.f};
    const bool has_s = (a.act & RN_ACT_SIGMOID) != 0, has_p = (a.act & RN_ACT_PRELU) != 0;
    const bool has_e = (a.act & RN_ACT_ELU) != 0;
    const float aa[4] = {al.x, al.y, al.z, al.w};
    // UNR rows in flight per thread: the loop is pure streaming (12-16 B in, 16 B out per lane), so the
    // loads of all UNR rows are issued before the first use
    constexpr int UNR = 4;
    for (long long rb = r0 + rsub; rb < r1; rb += (long long)rstep * UNR) {
        float4 d[UNR], zv[UNR], yv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long r = rb + (long long)u * rstep;
            if (r < r1) {
                const size_t e = (size_t)r * G + g;
                d[u] = reinterpret_cast<const float4*>(a.dy)[e];
                if (has_p) zv[u] = reinterpret_cast<const float4*>(a.z)[e];
                if (has_s || has_e) yv[u] = reinterpret_cast<const float4*>(a.y)[e];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long r = rb + (long long)u * rstep;
            if (r < r1) {
                const size_t e = (size_t)r * G + g;
                float dv[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
                if (has_s) {
                    const float yy[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) dv[q] *= yy[q] * (1.f - yy[q]);
                }
                if (has_e) {
                    const float yy[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) dv[q] = yy[q] < 0.f ? dv[q] * (yy[q] + 1.f) : dv[q];
                }
                if (has_p) {
                    const float zz[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        sa[q] += dv[q] * fminf(zz[q], 0.f);
                        dv[q] = zz[q] > 0.f ? dv[q] : dv[q] * aa[q];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) sb[q] += dv[q];
                if (a.dz) reinterpret_cast<float4*>(a.dz)[e] = make_float4(dv[0], dv[1], dv[2], dv[3]);
            }
        }
    }
    // reduce the rstep row-lanes of each channel group through LDS, then one atomic per channel
#pragma unroll
    for (int q = 0; q < 4; ++q) { red[0][threadIdx.x * 4 + q] = sb[q]; red[1][threadIdx.x * 4 + q] = sa[q]; }
    __syncthreads();
    if (threadIdx.x < gper) {
        float tb[4] = {0.f, 0.f, 0.f, 0.f}, ta[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < rstep; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tb[q] += red[0][(s * gper + threadIdx.x) * 4 + q];
                ta[q] += red[1][(s * gper + threadIdx.x) * 4 + q];
            }
        if (a.ws) {
            // the row block's partial sums, summed over the blocks by epilogue_bwd_reduce_kernel: 512 row blocks x 2 C atomics onto 2 C addresses
            // cost ~20 us of serialised tail per call (a third of a 100 MB reduce-only call)
            float* wp = a.ws + ((size_t)blockIdx.x * 2) * a.C + g * 4;
            *reinterpret_cast<float4*>(wp) = make_float4(tb[0], tb[1], tb[2], tb[3]);
            *reinterpret_cast<float4*>(wp + a.C) = make_float4(ta[0], ta[1], ta[2], ta[3]);
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (a.dbias) unsafeAtomicAdd(a.dbias + g * 4 + q, tb[q]);
            if ((a.act & RN_ACT_PRELU) && a.dalpha) unsafeAtomicAdd(a.dalpha + g * 4 + q, ta[q]);
        }
    }
}

// generic path (any C <= 1024, e.g. the 1|3-channel image head): LDS accumulators
__global__ __launch_bounds__(256)
void epilogue_bwd_gen_kernel(const EpiBwdArgs a)
{
    __shared__ float sb[1024], sa[1024];
    for (int c = threadIdx.x; c < a.C; c += 256) { sb[c] = 0.f; sa[c] = 0.f; }
    __syncthreads();
    const long long e0 = (long long)blockIdx.x * a.rows_per_block * a.C;
    long long e1 = e0 + (long long)a.rows_per_block * a.C;
    if (e1 > a.M * a.C) e1 = a.M * a.C;
    // C == 1: plain block reduction instead of same-address LDS atomics
    float loc_b = 0.f;
    for (long long e = e0 + threadIdx.x; e < e1; e += 256) {
        const int c = (int)(e % a.C);
        float d = a.dy[e];
        if (a.act & RN_ACT_SIGMOID) { const float yv = a.y[e]; d *= yv * (1.f - yv); }
        if (a.act & RN_ACT_ELU) { const float yv = a.y[e]; d = yv < 0.f ? d * (yv + 1.f) : d; }
        if (a.act & RN_ACT_PRELU) {
            const float zv = a.z[e];
            atomicAdd(&sa[c], d * fminf(zv, 0.f));
            d = zv > 0.f ? d : d * a.alpha[c];
        }
        if (a.C == 1) loc_b += d; else atomicAdd(&sb[c], d);
        if (a.dz) a.dz[e] = d;
    }
    if (a.C == 1) {
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) loc_b += __shfl_xor(loc_b, s);
        if ((threadIdx.x & 63) == 0) atomicAdd(&sb[0], loc_b);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < a.C; c += 256) {
        if (a.dbias) unsafeAtomicAdd(a.dbias + c, sb[c]);
        if ((a.act & RN_ACT_PRELU) && a.dalpha) unsafeAtomicAdd(a.dalpha + c, sa[c]);
    }
}

// sums the row-block partials [nb][2][C] of epilogue_bwd_vec_kernel: thread = one channel of one of the two sums, blockIdx.y = a slice of the
// row blocks; nslice atomics per address
__global__ __launch_bounds__(256)
void epilogue_bwd_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dbias, float* __restrict__ dalpha, int C, int nb, int per_slice)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * C) return;
    float* dst = i < C ? dbias : dalpha;
    if (!dst) return;
    const int b0 = blockIdx.y * per_slice, b1 = b0 + per_slice < nb ? b0 + per_slice : nb;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
        s0 += ws[(size_t)b * 2 * C + i];
        s1 += ws[(size_t)(b + 1) * 2 * C + i];
        s2 += ws[(size_t)(b + 2) * 2 * C + i];
        s3 += ws[(size_t)(b + 3) * 2 * C + i];
    }
    for (; b < b1; ++b) s0 += ws[(size_t)b * 2 * C + i];
    if (b0 < b1) unsafeAtomicAdd(dst + (i < C ? i : i - C), (s0 + s1) + (s2 + s3));
}

static int epilogue_bwd_impl(const float* dy, const float* z, const float* y, const float* alpha,
                             float* dz, float* dbias, float* dalpha, size_t M, int C, int act, float* ws, size_t ws_floats, void* stream)
{
    if (!dy || M < 1 || C < 1) return rn_set_error(RN_E_INVALID, "rn_epilogue_bwd: bad arguments");
    if ((act & RN_ACT_PRELU) && (!z || !alpha)) return rn_set_error(RN_E_INVALID, "rn_epilogue_bwd: PReLU needs z and alpha");
    if ((act & (RN_ACT_SIGMOID | RN_ACT_ELU)) && !y) return rn_set_error(RN_E_INVALID, "rn_epilogue_bwd: sigmoid / ELU need y");
    EpiBwdArgs a{dy, z, y, alpha, dz, dbias, dalpha, (long long)M, C, act, 0, nullptr};
    hipStream_t st = (hipStream_t)stream;
    const int G = C / 4;
    if (C % 4 == 0 && ((G <= 256 && 256 % G == 0) || G % 256 == 0)) {
        const int gy = G <= 256 ? 1 : G / 256;
        // large tensors (>= 8 MiB per operand): 1024-thread workgroups -- with the row-block count capped at ~512 a 256-thread block keeps
        // only 8 waves per CU in flight (1.8 TB/s on the res2 layers); RN_EPI_NT=256 | 1024 forces either (measurement)
        static const int nt_env = getenv("RN_EPI_NT") ? atoi(getenv("RN_EPI_NT")) : 0;
        const int NT = nt_env == 256 || nt_env == 1024 ? nt_env : ((long long)M * C >= (2ll << 20) ? 1024 : 256);
        const int gper = G < 256 ? G : 256;
        const int rstep = NT / gper;
        // ~512 row blocks: every block ends with 2*C same-address atomics, which the L2 serialises per
        // cache line (~40 ns each) -- thousands of blocks made THAT the critical path (0.31 ms for a
        // 300 MB stream); 2 blocks per CU with 4 rows in flight per thread still cover the HBM latency
        long long rpb = ((long long)M + 511) / 512;
        rpb = (rpb + 4 * rstep - 1) / (4 * rstep) * (4 * rstep);       // whole 4-row unrolls
        if (rpb < 4 * rstep) rpb = 4 * rstep;
        a.rows_per_block = (int)rpb;
        const long long nb = ((long long)M + rpb - 1) / rpb;
        const bool sums = dbias || ((act & RN_ACT_PRELU) && dalpha);
        const bool two_stage = ws && sums && nb >= 32 && (size_t)nb * 2 * C <= ws_floats;      // (few row blocks: the atomics are no tail worth a launch)
        if (two_stage) a.ws = ws;
        if (NT == 1024) hipLaunchKernelGGL(epilogue_bwd_vec_kernel<1024>, dim3((unsigned)nb, gy), dim3(1024), 0, st, a);
        else hipLaunchKernelGGL(epilogue_bwd_vec_kernel<256>, dim3((unsigned)nb, gy), dim3(256), 0, st, a);
        if (two_stage) {
            const int nslice = 16, per = (int)((nb + nslice - 1) / nslice);
            hipLaunchKernelGGL(epilogue_bwd_reduce_kernel, dim3((unsigned)((2 * C + 255) / 256), nslice), dim3(256), 0, st,
                               ws, dbias, (act & RN_ACT_PRELU) ? dalpha : nullptr, C, (int)nb, per);
        }
    } else {
        if (C > 1024) return rn_set_error(RN_E_UNSUPPORTED, "rn_epilogue_bwd: C=%d", C);
        long long rpb = ((long long)M + 4095) / 4096;
        if (rpb < 256) rpb = 256;
        a.rows_per_block = (int)rpb;
        const long long nb = ((long long)M + rpb - 1) / rpb;
        hipLaunchKernelGGL(epilogue_bwd_gen_kernel, dim3((unsigned)nb), dim3(256), 0, st, a);
    }
    return rn_check_launch("rn_epilogue_bwd");
}

extern "C" int rn_epilogue_bwd(const float* dy, const float* z, const float* y, const float* alpha,
                               float* dz, float* dbias, float* dalpha, size_t M, int C, int act, void* stream)
{
    return epilogue_bwd_impl(dy, z, y, alpha, dz, dbias, dalpha, M, C, act, nullptr, 0, stream);
}

extern "C" size_t rn_epilogue_bwd_workspace_floats(size_t M, int C)
{
    (void)M;
    return C >= 1 ? (size_t)512 * 2 * (size_t)C : 0;          // at most 512 row blocks x [2][C]
}

extern "C" int rn_epilogue_bwd_ws(const float* dy, const float* z, const float* y, const float* alpha,
                                  float* dz, float* dbias, float* dalpha, size_t M, int C, int act,
                                  float* ws, size_t ws_floats, void* stream)
{
    return epilogue_bwd_impl(dy, z, y, alpha, dz, dbias, dalpha, M, C, act, ws, ws_floats, stream);
}

// ---------------------------------------------------------------------------------------------
// Reconstruction loss and its gradient w.r.t. the prediction (RenderNet_Shader.py:159-163).
//   mode 0 (greyscale): loss = mean_b( -sum_{hwc} t*log(1e-6+p) + (1-t)*log(1e-6+1-p) )
//   mode 1 (RGB):       loss = mean over all elements (t-p)^2   (tf.losses.mean_squared_error)
// loss_sum (double, device) is accumulated: the caller zeroes it and reads loss = *loss_sum.
// `divisor` is the denominator of the mean: the batch size (mode 0) or the element count (mode 1) --
// the GLOBAL one when ranks each hold a shard, so that the sum of the per-rank gradients (and of the
// per-rank loss_sum values) is the gradient (value) of the global mean.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void loss_kernel(const float* __restrict__ p, const float* __restrict__ t, float* __restrict__ dp,
                 double* __restrict__ loss_sum, size_t n, int mode, float inv_div)
{
    double loc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float pv = p[i], tv = t[i];
        if (mode == 0) {
            const float a0 = 1e-6f + pv, a1 = 1e-6f + 1.f - pv;
            loc += (double)(-(tv * logf(a0) + (1.f - tv) * logf(a1)));
            if (dp) dp[i] = -(tv / a0 - (1.f - tv) / a1) * inv_div;
        } else {
            const float d = pv - tv;
            loc += (double)(d * d);
            if (dp) dp[i] = 2.f * d * inv_div;
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) loc += __shfl_xor(loc, s);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = loc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, (part[0] + part[1] + part[2] + part[3]) * (double)inv_div);
}

extern "C" int rn_loss_fwd_bwd(const float* pred, const float* target, float* dpred, double* loss_sum,
                               size_t n, double divisor, int mode, void* stream)
{
    if (!pred || !target || !loss_sum || n < 1 || !(divisor > 0.0) || (mode != 0 && mode != 1))
        return rn_set_error(RN_E_INVALID, "rn_loss_fwd_bwd: bad arguments");
    const float inv_div = (float)(1.0 / divisor);
    size_t nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(loss_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pred, target, dpred,
                       loss_sum, n, mode, inv_div);
    return rn_check_launch("rn_loss_fwd_bwd");
}

// ---------------------------------------------------------------------------------------------
// tf.train.AdamOptimizer (RenderNet_Shader.py:166) over one flat parameter buffer:
//   m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= lr_t * m / (sqrt(v) + eps)
// with lr_t = lr * sqrt(1-b2^t)/(1-b1^t) computed by the caller (TF's formulation: eps is NOT scaled).
// grad_scale multiplies g first (1 for summed shard gradients of a global-mean loss).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                 size_t n4, size_t n, float lr_t, float b1, float b2, float eps, float gs)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* pp = &pv.x; const float* gg = &gv.x; float* mm = &mv.x; float* vq = &vv.x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float gq = gg[q] * gs;
            mm[q] = b1 * mm[q] + (1.f - b1) * gq;
            vq[q] = b2 * vq[q] + (1.f - b2) * gq * gq;
            pp[q] -= lr_t * mm[q] / (sqrtf(vq[q]) + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0) {
        const size_t i = n4 * 4 + threadIdx.x;
        if (i < n) {
            const float gq = g[i] * gs;
            const float mq = b1 * m[i] + (1.f - b1) * gq;
            const float vq = b2 * v[i] + (1.f - b2) * gq * gq;
            m[i] = mq; v[i] = vq;
            p[i] -= lr_t * mq / (sqrtf(vq) + eps);
        }
    }
}

extern "C" int rn_adam_step(float* param, const float* grad, float* m, float* v, size_t n,
                            float lr_t, float beta1, float beta2, float eps, float grad_scale, void* stream)
{
    if (!param || !grad || !m || !v || n < 1) return rn_set_error(RN_E_INVALID, "rn_adam_step: bad arguments");
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15)
        return rn_set_error(RN_E_INVALID, "rn_adam_step: buffers must be 16-byte aligned");
    const size_t n4 = n / 4;
    size_t nb = (n4 + 255) / 256;
    // one float4 per thread (no grid-stride loop): 237 M parameters 1.22 ms = 5.4 TB/s over the seven streams; capped at 4096 workgroups 1.42 ms
    // (scripts/adam_bench.py; RN_ADAM_WGS = cap, measurement)
    static const size_t cap = getenv("RN_ADAM_WGS") ? (size_t)atoi(getenv("RN_ADAM_WGS")) : (size_t)0x7fffffff;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, param, grad, m, v,
                       n4, n, lr_t, beta1, beta2, eps, grad_scale);
    return rn_check_launch("rn_adam_step");
}

// tf.train.GradientDescentOptimizer (the four latent-variable optimisers of the inverse-rendering loop,
// Reconstruct_RenderNet_Face.py:397-413): p -= lr * g.  Latents are a few hundred floats: one small launch.
__global__ void sgd_kernel(float* __restrict__ param, const float* __restrict__ grad, size_t n, float lr)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        param[i] -= lr * grad[i];
}

extern "C" int rn_sgd_step(float* param, const float* grad, size_t n, float lr, void* stream)
{
    if (!param || !grad || n < 1) return rn_set_error(RN_E_INVALID, "rn_sgd_step: bad arguments");
    size_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, param, grad, n, lr);
    return rn_check_launch("rn_sgd_step");
}

// ---------------------------------------------------------------------------------------------
// tf.nn.dropout (RenderNet_Shader.py:39,43,47,88,103,107-123; tools/layer_util.py:124-131):
//     y = x / keep_prob * floor(keep_prob + u),   u ~ U[0,1)
// The uniform comes from a counter-based generator (Philox4x32-10, Salmon et al. 2011): element e uses word e%4 of
// philox(counter = (e/4, stream), key = seed), u = (word >> 8) * 2^-24.  Nothing is stored: the backward pass applies
// the SAME call (same seed and stream) to the gradient and so regenerates the forward mask bit for bit.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c.x, p1 = 0xCD9E8D57ull * c.z;
        c = make_uint4((unsigned)(p1 >> 32) ^ c.y ^ k.x, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k.y, (unsigned)p0);
        k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
    }
    return c;
}

// keep = floor(keep_prob + u) as TF writes it, clamped to {0, 1}: for fp32 keep_prob < 1 and u a multiple of 2^-24 below 1 the
// sum cannot round up to 2 (it is at most 2 - 2^-23, which is representable), the clamp makes that independent of the argument.
__device__ __forceinline__ float dropout_keep(float keep_prob, unsigned word)
{
    return fminf(floorf(keep_prob + (float)(word >> 8) * 5.9604644775390625e-08f), 1.0f);
}

// VEC: x and y are 16-byte aligned (whole float4 groups by vector loads/stores); otherwise every element goes through scalar
// accesses (an offset view of a larger gradient buffer).  The mask is the same function of the ELEMENT index either way.
template <bool VEC>
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float keep_prob, float inv_keep,
                               unsigned long long seed, unsigned long long stream)
{
    const size_t nq = (n + 3) / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (size_t)gridDim.x * blockDim.x) {
        const uint4 r = philox4x32_10(make_uint4((unsigned)q, (unsigned)(q >> 32), (unsigned)stream, (unsigned)(stream >> 32)),
                                      make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
        if (VEC && q * 4 + 3 < n) {
            float4 v = *reinterpret_cast<const float4*>(x + q * 4);
            float* e = reinterpret_cast<float*>(&v);
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i] = e[i] * inv_keep * dropout_keep(keep_prob, w[i]);
            *reinterpret_cast<float4*>(y + q * 4) = v;
        } else {
            for (int i = 0; i < 4 && q * 4 + i < n; ++i)
                y[q * 4 + i] = x[q * 4 + i] * inv_keep * dropout_keep(keep_prob, w[i]);
        }
    }
}

extern "C" int rn_dropout(const float* x, float* y, size_t n, float keep_prob, unsigned long long seed,
                          unsigned long long stream_id, void* stream)
{
    if (n == 0) return RN_OK;                                   // an empty tensor: nothing to do
    if (!x || !y) return rn_set_error(RN_E_INVALID, "rn_dropout: null pointer");
    if (!(keep_prob > 0.f) || keep_prob > 1.f) return rn_set_error(RN_E_INVALID, "rn_dropout: keep_prob %g not in (0, 1]", keep_prob);
    if ((((uintptr_t)x | (uintptr_t)y) & 3) != 0) return rn_set_error(RN_E_INVALID, "rn_dropout: pointers must be float-aligned");
    const bool vec = (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    size_t nb = ((n + 3) / 4 + 255) / 256;
    if (nb > 16384) nb = 16384;
    if (vec)
        hipLaunchKernelGGL(dropout_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, y, n, keep_prob,
                           1.0f / keep_prob, seed, stream_id);
    else
        hipLaunchKernelGGL(dropout_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, y, n, keep_prob,
                           1.0f / keep_prob, seed, stream_id);
    return rn_check_launch("rn_dropout");
}

// ---------------------------------------------------------------------------------------------
// Input gradient of a forward conv by direct gather, for the strided / channel-starved stem
// (e_conv2: 3^3, stride (1,1,2), 8 -> 16; RenderNet_Shader.py:40-43):
//     dx[b,i,c] = sum_{t, n : (i + P - t) % S == 0, o = (i+P-t)/S in range} dz[b,o,n] * w[t][c][n]
// One thread owns one input position and all (<= CI) input channels.  The filter comes in the
// forward-packed layout ([K/4][Npad][4], k = tap*Cin + c) and is staged in LDS as [tap][n][CI].
// ---------------------------------------------------------------------------------------------
struct DgradDirectArgs {
    const float* dz; const float* w; float* dx;
    long long Mi;
    int I0, I1, I2, Cin, O0, O1, O2, Cout, Npad;
    int K0, K1, K2, S0, S1, S2, P0, P1, P2;
};

template <int CI>
__global__ __launch_bounds__(256)
void conv_dgrad_direct_kernel(const DgradDirectArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);          // [taps][Cout][CI]
    const int taps = a.K0 * a.K1 * a.K2;
    for (int i = threadIdx.x; i < taps * a.Cout * CI; i += blockDim.x) {
        const int c = i % CI, n = (i / CI) % a.Cout, tap = i / (CI * a.Cout);
        const int k = tap * a.Cin + c;
        wl[i] = (c < a.Cin) ? a.w[((size_t)(k >> 2) * a.Npad + n) * 4 + (k & 3)] : 0.f;
    }
    __syncthreads();
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.Mi) return;
    long long t = m;
    const int i2 = (int)(t % a.I2); t /= a.I2;
    const int i1 = (int)(t % a.I1); t /= a.I1;
    const int i0 = (int)(t % a.I0); const int b = (int)(t / a.I0);
    float acc[CI];
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[c] = 0.f;
    for (int t0 = 0; t0 < a.K0; ++t0) {
        const int u0 = i0 + a.P0 - t0;
        if (u0 < 0 || u0 % a.S0 != 0 || u0 / a.S0 >= a.O0) continue;
        for (int t1 = 0; t1 < a.K1; ++t1) {
            const int u1 = i1 + a.P1 - t1;
            if (u1 < 0 || u1 % a.S1 != 0 || u1 / a.S1 >= a.O1) continue;
            for (int t2 = 0; t2 < a.K2; ++t2) {
                const int u2 = i2 + a.P2 - t2;
                if (u2 < 0 || u2 % a.S2 != 0 || u2 / a.S2 >= a.O2) continue;
                const float* gp = a.dz + ((((long long)b * a.O0 + u0 / a.S0) * a.O1 + u1 / a.S1) * a.O2 + u2 / a.S2) * a.Cout;
                const float* wp = wl + (size_t)((t0 * a.K1 + t1) * a.K2 + t2) * a.Cout * CI;
                for (int n = 0; n < a.Cout; ++n) {
                    const float gv = gp[n];
#pragma unroll
                    for (int c = 0; c < CI; ++c) acc[c] = fmaf(gv, wp[n * CI + c], acc[c]);
                }
            }
        }
    }
    float* op = a.dx + m * a.Cin;
#pragma unroll
    for (int c = 0; c < CI; ++c)
        if (c < a.Cin) op[c] = acc[c];
}

int rn_launch_conv_dgrad_direct(const float* dz, const float* w_fwd_packed, float* dx, int B, const int* I, int Cin,
                                const int* O, int Cout, const int* K, const int* S, const int* P, hipStream_t st)
{
    DgradDirectArgs a;
    a.dz = dz; a.w = w_fwd_packed; a.dx = dx;
    a.Mi = (long long)B * I[0] * I[1] * I[2];
    a.I0 = I[0]; a.I1 = I[1]; a.I2 = I[2]; a.Cin = Cin;
    a.O0 = O[0]; a.O1 = O[1]; a.O2 = O[2]; a.Cout = Cout; a.Npad = rn_round_up(Cout, 32);
    a.K0 = K[0]; a.K1 = K[1]; a.K2 = K[2]; a.S0 = S[0]; a.S1 = S[1]; a.S2 = S[2];
    a.P0 = P[0]; a.P1 = P[1]; a.P2 = P[2];
    if (Cin > 16) return rn_set_error(RN_E_UNSUPPORTED, "conv_dgrad_direct: Cin=%d > 16", Cin);
    const int CI = Cin <= 4 ? 4 : (Cin <= 8 ? 8 : 16);
    const size_t lds = (size_t)K[0] * K[1] * K[2] * Cout * CI * 4;
    if (lds > 64 * 1024) return rn_set_error(RN_E_UNSUPPORTED, "conv_dgrad_direct: filter %zu B exceeds LDS budget", lds);
    const long long nb = (a.Mi + 255) / 256;
    if (nb > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv_dgrad_direct: grid too large");
    if (CI == 4) hipLaunchKernelGGL(conv_dgrad_direct_kernel<4>, dim3((unsigned)nb), dim3(256), lds, st, a);
    else if (CI == 8) hipLaunchKernelGGL(conv_dgrad_direct_kernel<8>, dim3((unsigned)nb), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(conv_dgrad_direct_kernel<16>, dim3((unsigned)nb), dim3(256), lds, st, a);
    return rn_check_launch("conv_dgrad_direct");
}
