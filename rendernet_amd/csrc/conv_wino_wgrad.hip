// Filter gradient of the stride-1 3x3 2-D convs through the Winograd F(2x2,3x3) identities, exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32) -- tf.nn.conv2d_backprop_filter of the res_block_2d / *_skip convs
// (tools/layer_util.py:101-104, RenderNet_Shader.py:71-84, :91-99), 38 % of the training step with the direct kernel.
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A      =>      dg = G^T [ sum_tiles (B^T d B) .* (A dY A^T) ] G
//
// i.e. per xi = (i, j) a GEMM  dU[xi][ci][co] = sum_t V[xi][t][ci] * dM[xi][t][co]  over the 2x2-output tiles t, with
// V = B^T d B from the layer input (4x4 patch of the tile) and dM = A dY A^T from the 2x2 tile of the output gradient:
// 16 multiplies per tile and channel pair instead of 36.  Both transforms happen at fragment-read time; the 16 partial
// sums of a (ci, co) pair sit in one lane, so the inverse transform G^T dU G (16 -> 9 taps) is a per-lane sum before the
// result is added into dw [3,3,Cin,Cout] (TF layout, fp32 atomics: the call accumulates, like the direct kernel).
//
// Workgroup: 512 threads = 8 waves (two per SIMD); output block 64 ci x 64 co x 16 xi; wave (wi, wo) owns 16 ci x 32 co
// (two 16x16 MFMA tiles, 128 accumulator registers).  K loop over groups of 4x4 tiles (8x8 output pixels): the 10x10-pixel
// input patch x 64 ci (25 KiB) and the 8x8-pixel gradient patch x 64 co (16 KiB) go global -> LDS by DMA, two stages.
// One MFMA sums over k = 4 tiles: lane group kq supplies tile row kq, MFMA s of a step tile column s.
#include "rn_common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WinoWgradArgs {
    const float* x; const float* dz; float* dw;
    unsigned x_bytes, dz_bytes;
    int B, H, W, Cin, Cout;
    int gh, gw;                 // 4x4-tile groups per image along H and W (8 output pixels each)
    int ngroups;                // B*gh*gw
    int nci, nco;               // Cin/64, Cout/64
    int ksplit;                 // workgroups sharing one output block (split over the tile groups)
};

namespace {
constexpr int XPIX = 100, XPIECES = 25;          // 10x10 patch pixels x 256 B = 25 DMA pieces of 4 pixels
constexpr int ZPIX = 64, ZPIECES = 16;           // 8x8 gradient pixels x 256 B
constexpr int WG_XB = XPIECES * 1024, WG_ZB = ZPIECES * 1024, WG_STAGE = WG_XB + WG_ZB;   // 41 984 B per stage
constexpr unsigned WGOOB = 0x80000000u;
}

__global__ __launch_bounds__(512, 1)
void conv_wino_wgrad_kernel(const WinoWgradArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [stage][x 25 KiB | dz 16 KiB]
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    const int wi = wave >> 1, wo = wave & 1;                          // 16-ci group (0..3), 32-co half (0..1)

    // block -> (ci block, co block, split)
    int blk = blockIdx.x;
    const int sp = blk % a.ksplit; blk /= a.ksplit;
    const int cob = blk % a.nco, cib = blk / a.nco;
    const int per = (a.ngroups + a.ksplit - 1) / a.ksplit;
    const int g_begin = sp * per, g_end = min(a.ngroups, g_begin + per);
    if (g_begin >= g_end) return;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz), 0, a.dz_bytes, 0x00020000);

    // DMA pieces of a stage: 0..24 = input patch (4 pixels x 256 B each), 25..40 = gradient patch; wave w moves pieces
    // w, w + 8, ... (6 slots, the last ones beyond 40 are idle).  Per-lane offsets depend on the group: recomputed per step.
    f32x4 acc[16][2];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][nt][r] = 0.f;

    auto issue = [&](int g, int stage) {
        const int gx = g % a.gw, gy = (g / a.gw) % a.gh, b = g / (a.gw * a.gh);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int p = wave + 8 * i;
            if (p >= XPIECES + ZPIECES) break;                         // uniform
            const int px4 = lane >> 4, c16 = lane & 15;                // pixel within the piece, 16-B chunk (4 channels)
            if (p < XPIECES) {
                const int q = p * 4 + px4;                             // patch pixel 0..99
                const int py = q / 10, pxx = q - py * 10;
                const int iy = gy * 8 - 1 + py, ix = gx * 8 - 1 + pxx;
                const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const int ch = c16 ^ (((py >> 1) & 3) << 2);           // bank swizzle: slot c16 of pixel row py holds chunk ch
                const unsigned off = ok ? (unsigned)(((b * a.H + iy) * a.W + ix) * a.Cin + cib * 64 + ch * 4) * 4u : WGOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)(smem + stage * WG_STAGE + p * 1024), 16, off, 0, 0, 0);
            } else {
                const int q = (p - XPIECES) * 4 + px4;                 // gradient pixel 0..63
                const int py = q >> 3, pxx = q & 7;
                const int iy = gy * 8 + py, ix = gx * 8 + pxx;
                const bool ok = iy < a.H && ix < a.W;
                const int ch = c16 ^ (((py >> 1) & 3) << 2);
                const unsigned off = ok ? (unsigned)(((b * a.H + iy) * a.W + ix) * a.Cout + cob * 64 + ch * 4) * 4u : WGOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(zrsrc, (lds_void*)(smem + stage * WG_STAGE + WG_XB + (p - XPIECES) * 1024), 16, off, 0, 0, 0);
            }
        }
    };

    issue(g_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int stage = 0;
    // fragment-read bases (bytes): input pixel (2kq + ai, 2s + bi) channel wi*16 + l16; gradient pixel (2kq + p, 2s + q)
    // channel wo*32 + nt*16 + l16
    // The 16-B chunk slots of a pixel are XOR-swizzled with (pixel row >> 1) & 3 in bits 2-3 (see issue()): the four kq
    // groups of a fragment read would otherwise hit the same 16 banks (their rows are 5 KiB / 4 KiB apart).
    const unsigned xlow = (unsigned)((l16 >> 2) * 16 + (l16 & 3) * 4);
    const unsigned xbase01 = (unsigned)((2 * kq * 10) * 256 + ((wi ^ kq) << 6)) + xlow;                 // patch rows 2kq, 2kq+1
    const unsigned xbase23 = (unsigned)((2 * kq * 10) * 256 + ((wi ^ ((kq + 1) & 3)) << 6)) + xlow;     // patch rows 2kq+2, 2kq+3
    const unsigned zbase0 = (unsigned)(WG_XB + (2 * kq * 8) * 256 + (((wo * 2 + 0) ^ kq) << 6)) + xlow;
    const unsigned zbase1 = (unsigned)(WG_XB + (2 * kq * 8) * 256 + (((wo * 2 + 1) ^ kq) << 6)) + xlow;
    for (int g = g_begin; g < g_end; ++g) {
        if (g + 1 < g_end) issue(g + 1, stage ^ 1);
        const char* xs01 = smem + stage * WG_STAGE + xbase01;
        const char* xs23 = smem + stage * WG_STAGE + xbase23;
        const char* zs0 = smem + stage * WG_STAGE + zbase0;
        const char* zs1 = smem + stage * WG_STAGE + zbase1;
        // row transform of the 10 patch columns: t[i][col] = (B^T d)[i], B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
        float t_[4][10];
#pragma unroll
        for (int col = 0; col < 10; ++col) {
            const float d0 = *reinterpret_cast<const float*>(xs01 + (0 * 10 + col) * 256);
            const float d1 = *reinterpret_cast<const float*>(xs01 + (1 * 10 + col) * 256);
            const float d2 = *reinterpret_cast<const float*>(xs23 + (2 * 10 + col) * 256);
            const float d3 = *reinterpret_cast<const float*>(xs23 + (3 * 10 + col) * 256);
            t_[0][col] = d0 - d2; t_[1][col] = d1 + d2; t_[2][col] = d2 - d1; t_[3][col] = d1 - d3;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {                                  // tile column s of tile row kq
            float v_[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float c0 = t_[i][2 * s], c1 = t_[i][2 * s + 1], c2 = t_[i][2 * s + 2], c3 = t_[i][2 * s + 3];
                v_[i * 4 + 0] = c0 - c2; v_[i * 4 + 1] = c1 + c2; v_[i * 4 + 2] = c2 - c1; v_[i * 4 + 3] = c1 - c3;
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                // dM' = A' dY A'^T with A' = [[1,0],[1,1],[1,-1],[0,1]]: row / column 3 carry the opposite sign of A dY A^T
                // (A's last row is [0,-1]); the inverse transform below puts the sign back
                const char* zs = nt ? zs1 : zs0;
                const float y00 = *reinterpret_cast<const float*>(zs + (0 * 8 + 2 * s) * 256);
                const float y01 = *reinterpret_cast<const float*>(zs + (0 * 8 + 2 * s + 1) * 256);
                const float y10 = *reinterpret_cast<const float*>(zs + (1 * 8 + 2 * s) * 256);
                const float y11 = *reinterpret_cast<const float*>(zs + (1 * 8 + 2 * s + 1) * 256);
                const float u_[4][2] = {{y00, y01}, {y00 + y10, y01 + y11}, {y00 - y10, y01 - y11}, {y10, y11}};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float m0 = u_[i][0], m1 = u_[i][0] + u_[i][1], m2 = u_[i][0] - u_[i][1], m3 = u_[i][1];
                    acc[i * 4 + 0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v_[i * 4 + 0], m0, acc[i * 4 + 0][nt], 0, 0, 0);
                    acc[i * 4 + 1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v_[i * 4 + 1], m1, acc[i * 4 + 1][nt], 0, 0, 0);
                    acc[i * 4 + 2][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v_[i * 4 + 2], m2, acc[i * 4 + 2][nt], 0, 0, 0);
                    acc[i * 4 + 3][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v_[i * 4 + 3], m3, acc[i * 4 + 3][nt], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stage ^= 1;
    }

    // epilogue: dg = G^T dU G with dU[i][j] = sgn_i sgn_j acc[4i+j] (sgn_3 = -1, see above),
    // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].  C/D layout: lane (l16, kq), register r: row 4kq + r = ci, col l16 = co.
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int co = cob * 64 + wo * 32 + nt * 16 + l16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = cib * 64 + wi * 16 + 4 * kq + r;
            float e_[3][4];                                             // (G^T dU)[p][j]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sj = j == 3 ? -1.f : 1.f;
                const float u0 = acc[0 + j][nt][r] * sj, u1 = acc[4 + j][nt][r] * sj, u2 = acc[8 + j][nt][r] * sj;
                const float u3 = -acc[12 + j][nt][r] * sj;
                e_[0][j] = u0 + 0.5f * (u1 + u2);
                e_[1][j] = 0.5f * (u1 - u2);
                e_[2][j] = 0.5f * (u1 + u2) + u3;
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const float w0 = e_[p][0] + 0.5f * (e_[p][1] + e_[p][2]);
                const float w1 = 0.5f * (e_[p][1] - e_[p][2]);
                const float w2 = 0.5f * (e_[p][1] + e_[p][2]) + e_[p][3];
                float* d = a.dw + ((size_t)(p * 3) * a.Cin + ci) * a.Cout + co;
                unsafeAtomicAdd(d, w0);
                unsafeAtomicAdd(d + (size_t)a.Cin * a.Cout, w1);
                unsafeAtomicAdd(d + (size_t)2 * a.Cin * a.Cout, w2);
            }
        }
    }
}

bool rn_wino_wgrad_supported(int Cin, int Cout)
{
    static const bool off = getenv("RN_NO_WINOGRAD_WGRAD") != nullptr || getenv("RN_NO_WINOGRAD") != nullptr;
    return !off && Cin % 64 == 0 && Cout % 64 == 0;
}

// x [B,H,W,Cin], dz [B,H,W,Cout] -> dw [3,3,Cin,Cout] += conv2d_backprop_filter (3x3, stride 1, SAME)
int rn_launch_conv_wino_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W, int Cin, int Cout, hipStream_t st)
{
    if (Cin % 64 != 0 || Cout % 64 != 0) return rn_set_error(RN_E_UNSUPPORTED, "conv_wino_wgrad: Cin=%d Cout=%d (need %%64)", Cin, Cout);
    const long long xi = (long long)H * W * Cin * 4, zi = (long long)H * W * Cout * 4;
    if (xi >= 0x80000000LL || zi >= 0x80000000LL)
        return rn_set_error(RN_E_UNSUPPORTED, "conv_wino_wgrad: one batch item exceeds the 2 GiB buffer window");
    if (xi * B >= 0x80000000LL || zi * B >= 0x80000000LL) {
        const long long big = xi > zi ? xi : zi;
        const int chunk = (int)(0x7fffffffLL / big);
        for (int b0 = 0; b0 < B; b0 += chunk) {
            const int nb = B - b0 < chunk ? B - b0 : chunk;
            const int rc = rn_launch_conv_wino_wgrad(x + (size_t)b0 * (xi / 4), dz + (size_t)b0 * (zi / 4), dw, nb, H, W, Cin, Cout, st);
            if (rc != RN_OK) return rc;
        }
        return RN_OK;
    }
    WinoWgradArgs a;
    a.x = x; a.dz = dz; a.dw = dw;
    a.x_bytes = (unsigned)(xi * B); a.dz_bytes = (unsigned)(zi * B);
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.gh = (H + 7) / 8; a.gw = (W + 7) / 8;
    a.ngroups = B * a.gh * a.gw;
    a.nci = Cin / 64; a.nco = Cout / 64;
    int ks = 256 / (a.nci * a.nco);
    if (ks < 1) ks = 1;
    if (ks > a.ngroups) ks = a.ngroups;
    a.ksplit = ks;
    const size_t lds = (size_t)2 * WG_STAGE;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_wino_wgrad_kernel), lds); if (rc_ != RN_OK) return rc_; }
    hipLaunchKernelGGL(conv_wino_wgrad_kernel, dim3((unsigned)(a.nci * a.nco * ks)), dim3(512), lds, st, a);
    return rn_check_launch("conv_wino_wgrad");
}
