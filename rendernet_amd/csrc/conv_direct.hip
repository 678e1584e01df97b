// Direct (VALU) convolution for the channel-starved stem / tail layers of the path, which are
// HBM/L2-bound and hopeless MFMA shapes: e_conv1 (Cin=1|5 -> 8, 5^3 s2), e_conv2 (8 -> 16,
// 3^3 s(1,1,2)), e_conv11 (16 -> 1|3), and the texture decoder's 4/8-channel 3-D convs.
// (RenderNet_Shader.py:36-43, :125-131; RenderNet_Texture_Face_Normal.py:34-46.)
//
// One thread owns one output position and all (<= CO) output channels; the filter is staged
// once per workgroup into LDS as [k][CO] and read back with wave-uniform (broadcast) addresses;
// activations are read channels-last (Cin contiguous) straight from L1/L2.
#include "rn_common.h"

struct DirectArgs {
    const float* x; const float* w; const float* bias; const float* alpha; const float* res; float* y; float* z;
    long long M;
    int I0, I1, I2, Cin;
    int O0, O1, O2, Cout, Npad;
    int K0, K1, K2, S0, S1, S2, P0, P1, P2;
    long long os_b, os0, os1, os2, out_off;
    int act, Ktot;
};

template <int CO>
__global__ __launch_bounds__(256)
void conv_direct_kernel(const DirectArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);          // [Ktot][CO]
    for (int i = threadIdx.x; i < a.Ktot * CO; i += blockDim.x) {
        const int k = i / CO, n = i % CO;
        wl[i] = (n < a.Cout) ? a.w[((size_t)(k >> 2) * a.Npad + n) * 4 + (k & 3)] : 0.f;
    }
    __syncthreads();

    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.M) return;
    long long t = m;
    const int o2 = (int)(t % a.O2); t /= a.O2;
    const int o1 = (int)(t % a.O1); t /= a.O1;
    const int o0 = (int)(t % a.O0); const int b = (int)(t / a.O0);
    const int in0 = o0 * a.S0 - a.P0, in1 = o1 * a.S1 - a.P1, in2 = o2 * a.S2 - a.P2;

    float acc[CO];
#pragma unroll
    for (int n = 0; n < CO; ++n) acc[n] = 0.f;

    int k = 0;
    for (int t0 = 0; t0 < a.K0; ++t0) {
        const int i0 = in0 + t0;
        for (int t1 = 0; t1 < a.K1; ++t1) {
            const int i1 = in1 + t1;
            for (int t2 = 0; t2 < a.K2; ++t2, k += a.Cin) {
                const int i2 = in2 + t2;
                if ((unsigned)i0 >= (unsigned)a.I0 || (unsigned)i1 >= (unsigned)a.I1 ||
                    (unsigned)i2 >= (unsigned)a.I2) continue;
                const float* xp = a.x + ((((long long)b * a.I0 + i0) * a.I1 + i1) * a.I2 + i2) * a.Cin;
                const float* wp = wl + (size_t)k * CO;
                if ((a.Cin & 3) == 0) {
                    for (int c = 0; c < a.Cin; c += 4) {
                        const float4 xv = *reinterpret_cast<const float4*>(xp + c);
#pragma unroll
                        for (int n = 0; n < CO; ++n) {
                            acc[n] = fmaf(xv.x, wp[(c + 0) * CO + n], acc[n]);
                            acc[n] = fmaf(xv.y, wp[(c + 1) * CO + n], acc[n]);
                            acc[n] = fmaf(xv.z, wp[(c + 2) * CO + n], acc[n]);
                            acc[n] = fmaf(xv.w, wp[(c + 3) * CO + n], acc[n]);
                        }
                    }
                } else {
                    for (int c = 0; c < a.Cin; ++c) {
                        const float xv = xp[c];
#pragma unroll
                        for (int n = 0; n < CO; ++n) acc[n] = fmaf(xv, wp[c * CO + n], acc[n]);
                    }
                }
            }
        }
    }

    const long long oo = a.out_off + b * a.os_b + o0 * a.os0 + o1 * a.os1 + o2 * a.os2;
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

template <int CO>
static int launch_direct(const DirectArgs& a, hipStream_t st)
{
    const size_t lds = (size_t)a.Ktot * CO * sizeof(float);
    if (lds > 160 * 1024) return rn_set_error(RN_E_UNSUPPORTED, "conv_direct: filter %zu B exceeds LDS", lds);
    auto kern = conv_direct_kernel<CO>;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(160 * 1024)); if (rc_ != RN_OK) return rc_; }
    const long long nb = (a.M + 255) / 256;
    if (nb > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv_direct: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(256), lds, st, a);
    return rn_check_launch("conv_direct");
}

int rn_launch_conv_direct(const RnConvProblem& p, hipStream_t st)
{
    DirectArgs a;
    a.x = p.x; a.w = p.w; a.bias = p.bias; a.alpha = p.alpha; a.res = p.residual; a.y = p.y; a.z = p.preact;
    a.M = (long long)p.B * p.O[0] * p.O[1] * p.O[2];
    if (a.M <= 0) return rn_set_error(RN_E_INVALID, "conv_direct: empty problem");
    a.I0 = p.I[0]; a.I1 = p.I[1]; a.I2 = p.I[2]; a.Cin = p.Cin;
    a.O0 = p.O[0]; a.O1 = p.O[1]; a.O2 = p.O[2]; a.Cout = p.Cout; a.Npad = p.Npad;
    a.K0 = p.K[0]; a.K1 = p.K[1]; a.K2 = p.K[2];
    a.S0 = p.S[0]; a.S1 = p.S[1]; a.S2 = p.S[2];
    a.P0 = p.P[0]; a.P1 = p.P[1]; a.P2 = p.P[2];
    a.os_b = p.os_b; a.os0 = p.os[0]; a.os1 = p.os[1]; a.os2 = p.os[2]; a.out_off = p.out_off;
    a.act = p.act;
    a.Ktot = p.K[0] * p.K[1] * p.K[2] * p.Cin;
    if (p.Cout <= 1) return launch_direct<1>(a, st);
    if (p.Cout <= 4) return launch_direct<4>(a, st);
    if (p.Cout <= 8) return launch_direct<8>(a, st);
    if (p.Cout <= 16) return launch_direct<16>(a, st);
    return rn_set_error(RN_E_UNSUPPORTED, "conv_direct: Cout=%d > 16 with Cin=%d not a multiple of 16", p.Cout, p.Cin);
}
