// Filter gradient of the wide stride-1 3x3 2-D convs through Winograd F(4x4,3x3), exact-fp32 MFMA for the multiply stage --
// tf.nn.conv2d_backprop_filter of the res_block_2d / *_skip convs (tools/layer_util.py:101-104, RenderNet_Shader.py:71-84,
// :91-99) in the training step.
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A      =>      dg = G^T [ sum_tiles (B^T d B) .* (A dY A^T) ] G
//
//   1. wino_input_kernel (conv_wino43.hip)   x  [B,H,W,Cin]   -> V  [36][T tiles][Cin]        (the forward's input transform)
//   2. wino_dout_kernel                      dz [B,H,W,Cout]  -> dM [36][T][Cout]             (A dY A^T per 4x4 output tile)
//   3. wino43_wgrad_gemm_kernel              V, dM            -> dU [36][Cin][Cout]           (36 GEMMs Cin x T x Cout)
//   4. wino_dfilter_kernel                   dU               -> dw [3,3,Cin,Cout] += G^T dU G
//
// 36 multiplies per 4x4 outputs and channel pair instead of 144 (conv_wino_wgrad.hip, F(2x2,3x3): 64).  The GEMM reduces
// over the tiles (K = T): block 256 ci x 256 co, K step 32 tiles; both operand panels are [tile][channel] rows of 1 KiB that
// go global -> LDS by DMA, one row per wave instruction, their sixteen 64-byte granules XOR-swizzled with (row & 3) so that
// the two k-lanes of a fragment read (rows t, t+1; 32 channels each) hit different banks.  v_mfma_f32_32x32x2_f32 with dM as the A
// operand: accumulator registers come in groups of four consecutive output channels of one input channel (16-byte stores into dU).  Same persistent, XCD-aware
// enumeration and one-sub-group-ahead fragment pipeline as the forward GEMM (conv_wino43.hip).
#include "rn_common.h"
#include "wino_mats.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int WBK = 32;                            // tiles per K step
constexpr int W_OPB = WBK * 256 * 4;               // one operand of a stage: 32 rows x 1 KiB
constexpr int W_STAGE = 2 * W_OPB;                 // V rows | dM rows
__device__ __forceinline__ unsigned xcd_contiguous(unsigned blk, unsigned nblk8) { return (blk & 7u) * (nblk8 >> 3) + (blk >> 3); }
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// 2. dM = A dY A^T: the 4x4 tile of the output gradient -> 6x6 (the adjoint of the output transform).  thread = (tile, 4 ch)
template <class S>
__global__ __launch_bounds__(256)
void wino_dout_kernel(const float* __restrict__ dz, float* __restrict__ dM, int H, int W, int C, int th, int tw,
                      long long T, unsigned nblk8)
{
    constexpr int A = S::TA;
    const unsigned blk = xcd_contiguous(blockIdx.x, nblk8);
    const long long idx = (long long)blk * 256 + threadIdx.x;
    const int C4 = C >> 2;
    const int c4 = (int)(idx % C4);
    const long long t = idx / C4;
    if (t >= T) return;
    const int tx = (int)(t % tw), ty = (int)((t / tw) % th);
    const long long b = t / ((long long)tw * th);
    const float* zb = dz + ((size_t)b * H * W) * C + c4 * 4;
    f32x4 tt[A][4];                                            // (A dY)[i][q]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 d[4];
        const int ox = 4 * tx + q;
#pragma unroll
        for (int p_ = 0; p_ < 4; ++p_) {
            const int oy = 4 * ty + p_;
            d[p_] = (oy < H && ox < W) ? *reinterpret_cast<const f32x4*>(zb + ((size_t)oy * W + ox) * C) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < A; ++i) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p_ = 0; p_ < 4; ++p_) {
                const float c = S::AT(p_, i);
                if (c != 0.f) acc += c * d[p_];
            }
            tt[i][q] = acc;
        }
    }
    float* mb = dM + (size_t)t * C + c4 * 4;
    const size_t plane = (size_t)T * C;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j < A; ++j) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float c = S::AT(q, j);
                if (c != 0.f) acc += c * tt[i][q];
            }
            *reinterpret_cast<f32x4*>(mb + (size_t)(i * A + j) * plane) = acc;
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// 4. dw[a][b][ci][co] += sum_{i,j} G[i][a] G[j][b] dU[xi = (i,j)][ci][co].  thread = (ci, 4 co)
template <class S>
__global__ __launch_bounds__(256)
void wino_dfilter_kernel(const float* __restrict__ dU, float* __restrict__ dw, int Cin, int Cout)
{
    constexpr int A = S::TA, R = S::R;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)Cin * (Cout / 4);
    if (idx >= n) return;
    const size_t plane = (size_t)Cin * Cout;
    const float* ub = dU + idx * 4;
    f32x4 e[R][A];                                             // (G^T dU)[a][j]
#pragma unroll
    for (int a_ = 0; a_ < R; ++a_)
#pragma unroll
        for (int j = 0; j < A; ++j) e[a_][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(ub + (size_t)(i * A + j) * plane);
#pragma unroll
            for (int a_ = 0; a_ < R; ++a_) {
                const float c = (float)S::G(i, a_);
                if (c != 0.f) e[a_][j] += c * u;
            }
        }
#pragma unroll
    for (int a_ = 0; a_ < R; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < R; ++b_) {
            f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < A; ++j) {
                const float c = (float)S::G(j, b_);
                if (c != 0.f) w += c * e[a_][j];
            }
            f32x4* d = reinterpret_cast<f32x4*>(dw + (size_t)(a_ * R + b_) * plane + idx * 4);
            *d += w;
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3. dU[xi] (Cin x Cout) = V[xi]^T (Cin x T) . dM[xi] (T x Cout).  One item = one 256 x 256 block of one xi over ALL tiles
// -- except the blocks of a last, partial round of the persistent grid, which are cut into `ksplit` parts along the tiles so
// that they fill the machine: those write partial blocks (SPLIT = true) that wino43_wgrad_reduce_kernel sums into dU.
// (fp32 atomics into a zeroed block instead: 0.31 ms for the 64 tail blocks of the res2 shape against 0.15 ms this way.)
struct W43WgradArgs {
    const float* V; const float* dM; float* dU; float* parts;   // parts: [tail block][part][256][256] partial blocks of the split items
    long long T;
    int Cin, Cout;
    int ciblocks, coblocks;         // 256-channel blocks
    int item_begin, item_end;       // this launch's blocks L = (xi*ciblocks + cib)*coblocks + cob
    int ksplit, steps_per_split, ksteps;   // parts per block; K steps (32 tiles) per part; K steps in all
    unsigned v_bytes, m_bytes, u_bytes;   // one xi plane of V / dM / dU
};

// dU block = sum of its parts, for the blocks [item_begin, item_end)
__global__ __launch_bounds__(256)
void wino43_wgrad_reduce_kernel(const W43WgradArgs a)
{
    const int tb = blockIdx.x / 64;                                   // 64 workgroups per block: 4 rows of 256 floats each
    const int L = a.item_begin + tb;
    const int cob = L % a.coblocks;
    const int rest = L / a.coblocks;
    const int cib = rest % a.ciblocks, xi = rest / a.ciblocks;
    const int row = (blockIdx.x % 64) * 4 + (threadIdx.x >> 6), c4 = threadIdx.x & 63;
    const float* p = a.parts + ((size_t)tb * a.ksplit) * 65536 + row * 256 + c4 * 4;
    f32x4 acc = *reinterpret_cast<const f32x4*>(p);
    for (int k = 1; k < a.ksplit; ++k) acc += *reinterpret_cast<const f32x4*>(p + (size_t)k * 65536);
    float* d = a.dU + (size_t)xi * a.Cin * a.Cout + (size_t)(cib * 256 + row) * a.Cout + cob * 256 + c4 * 4;
    *reinterpret_cast<f32x4*>(d) = acc;
}

template <bool SPLIT>
__global__ __launch_bounds__(512, 1)
void wino43_wgrad_gemm_kernel(const W43WgradArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [stage][V 32 x 256 | dM 32 x 256]
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hb = lane >> 5;                       // 32x32x2 MFMA: lane = (channel of the tile, k)
    const int wm = wave >> 1, wn = wave & 1;                          // 64-ci group (0..3), 128-co half (0..1)

    // fragment reads: group g of a stage = rows 4g .. 4g+3, consumed as two MFMA k-steps u = 0, 1 with rows 4g + 2u + hb.
    // A row's 64-byte granule x sits at granule x ^ swz(row & 3), swz = {0, 2, 1, 3}: the two rows of a k-step differ in
    // swizzle bit 1, so the 32 channels (two granules) of lanes hb = 0 and hb = 1 land in different banks.
    unsigned xk[2][2];                                                // [k-step u][tile parity c2] -> byte offset inside the group
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int row = 2 * u + hb, sw = ((row & 1) << 1) | (row >> 1);
            xk[u][c2] = (unsigned)((((c2 * 2 + (l32 >> 4)) ^ sw) << 6) + row * 1024 + (l32 & 15) * 4);
        }
    const unsigned vbase = (unsigned)(wm * 4 * 64);                   // + xk[u][mt]               (ci tile mt = 0, 1)
    const unsigned mbase = (unsigned)(W_OPB + wn * 8 * 64);           // + (nt >> 1) * 256 + xk[u][nt & 1]   (co tile nt = 0..3)
    // DMA: piece p = wave + 8i is row p of the stage (p & 3 == wave & 3); lane L moves the 16 bytes that land at position L
    const int dsw = (((wave & 3) & 1) << 1) | ((wave & 3) >> 1);
    const unsigned dlane = (unsigned)((((lane >> 2) ^ dsw) << 6) + (lane & 3) * 16);

    struct Item { const float* vplane; const float* mplane; float* ubase; int upitch, ci0, co0, nsteps; long long t0; };
    const int perm = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    const int nids = (a.item_end - a.item_begin) * a.ksplit;
    auto decode = [&](int r, Item& it) -> bool {
        const int id = r * (int)gridDim.x + perm;
        if (id >= nids) return false;
        const int L = a.item_begin + id / a.ksplit, sp = id % a.ksplit;
        const int cob = L % a.coblocks;
        const int rest = L / a.coblocks;
        const int cib = rest % a.ciblocks, xi = rest / a.ciblocks;
        it.ci0 = cib * 256; it.co0 = cob * 256;
        it.t0 = (long long)sp * a.steps_per_split * WBK;
        it.nsteps = min(a.steps_per_split, a.ksteps - sp * a.steps_per_split);
        it.vplane = a.V + (size_t)xi * a.T * a.Cin;
        it.mplane = a.dM + (size_t)xi * a.T * a.Cout;
        if (SPLIT) { it.ubase = a.parts + (size_t)id * 65536; it.upitch = 256; }
        else { it.ubase = a.dU + (size_t)xi * ((size_t)a.Cin * a.Cout) + (size_t)it.ci0 * a.Cout + it.co0; it.upitch = a.Cout; }
        return true;
    };
    auto issue = [&](const Item& it, int s, int stage) {
        const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(it.vplane), 0, a.v_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(it.mplane), 0, a.m_bytes, 0x00020000);
        char* sb = smem + stage * W_STAGE;
        const long long t = it.t0 + (long long)s * WBK;                // rows >= T fall outside the plane: zeros
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = wave + 8 * i;
            // the row offset rides in the VGPR offset: the hardware's range check does not see the scalar offset
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (lds_void*)(sb + p * 1024), 16,
                                                     dlane + (unsigned)(((t + p) * a.Cin + it.ci0) * 4), 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = wave + 8 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(mrsrc, (lds_void*)(sb + W_OPB + p * 1024), 16,
                                                     dlane + (unsigned)(((t + p) * a.Cout + it.co0) * 4), 0, 0, 0);
        }
    };

    f32x16 acc[2][4];
    // fragments of group g: for each of its two k-steps 2 ci tiles of V and 4 co tiles of dM
    auto load_frags = [&](const char* sb, int g, float (&v)[2][2], float (&m)[2][4]) {
        const char* base = sb + g * 4096;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) v[u][mt] = *reinterpret_cast<const float*>(base + vbase + xk[u][mt]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) m[u][nt] = *reinterpret_cast<const float*>(base + mbase + (nt >> 1) * 256 + xk[u][nt & 1]);
        }
    };
    auto mfmas = [&](const float (&v)[2][2], const float (&m)[2][4]) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(m[u][nt], v[u][mt], acc[mt][nt], 0, 0, 0);
    };

    Item cur, nxt;
    if (!decode(0, cur)) return;
    float v0[2][2], m0[2][4], v1[2][2], m1[2][4];
    issue(cur, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_frags(smem, 0, v0, m0);
    int stage = 0;
    for (int r = 0;; ++r) {
        const bool have_next = decode(r + 1, nxt);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        for (int s = 0; s < cur.nsteps; ++s) {
            const char* sb = smem + stage * W_STAGE;
            const char* sn = smem + (stage ^ 1) * W_STAGE;
            const bool last = s + 1 == cur.nsteps;
            if (!last) issue(cur, s + 1, stage ^ 1);
            else if (have_next) issue(nxt, 0, stage ^ 1);
            // eight groups of 4 tiles (two 32x32x2 k-steps each); the fragments of group g+1 are read while the 16 MFMAs of group g run
            load_frags(sb, 1, v1, m1); mfmas(v0, m0);
            load_frags(sb, 2, v0, m0); mfmas(v1, m1);
            load_frags(sb, 3, v1, m1); mfmas(v0, m0);
            load_frags(sb, 4, v0, m0); mfmas(v1, m1);
            load_frags(sb, 5, v1, m1); mfmas(v0, m0);
            load_frags(sb, 6, v0, m0); mfmas(v1, m1);
            load_frags(sb, 7, v1, m1); mfmas(v0, m0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (!last || have_next) load_frags(sn, 0, v0, m0);
            mfmas(v1, m1);
            stage ^= 1;
        }
        // D (32 x 32) = dM-tile (rows: co) x V-tile (cols: ci): register r of lane (l32, hb) is co (r & 3) + 8*(r >> 2) + 4*hb of
        // the tile, ci l32 -> four 16-byte stores per MFMA tile
        {
            float* ub = cur.ubase + (size_t)(wm * 64 + l32) * cur.upitch + wn * 128 + hb * 4;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4*>(ub + (size_t)mt * 32 * cur.upitch + nt * 32 + g * 8) =
                            f32x4{acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
        }
        if (!have_next) break;
        cur = nxt;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
bool rn_wino43_wgrad_supported(int scheme, int Cin, int Cout)
{
    if ((scheme != RN_WINO_F43 && scheme != RN_WINO_F44) || !rn_wino43_supported(scheme, 256, 256)) return false;   // 4x4-output schemes only
    static const bool off = getenv("RN_NO_WINOGRAD43_WGRAD") != nullptr || getenv("RN_NO_WINOGRAD43") != nullptr ||
                            getenv("RN_NO_WINOGRAD") != nullptr;
    return !off && Cin >= 256 && Cin % 256 == 0 && Cout >= 256 && Cout % 256 == 0;
}

size_t rn_wino43_wgrad_workspace_floats(int scheme, int B, int H, int W, int Cin, int Cout)
{
    const size_t T = (size_t)B * ((H + 3) / 4) * ((W + 3) / 4);
    const size_t nxi = (size_t)rn_wino_scheme_nxi(scheme);
    return nxi * T * ((size_t)Cin + Cout) + nxi * Cin * Cout + (size_t)256 * 65536;   // V, dM, dU, <= 256 partial blocks
}

// x [B,H,W,Cin], dz [B,H,W,Cout] -> dw [R,R,Cin,Cout] += conv2d_backprop_filter (RxR, stride 1, SAME); scheme F43: R = 3, F44: R = 4
int rn_launch_conv_wino43_wgrad(int scheme, const float* x, const float* dz, float* dw, float* ws, int B, int H, int W, int Cin,
                                int Cout, hipStream_t st)
{
    if (!rn_wino43_wgrad_supported(scheme, Cin, Cout))
        return rn_set_error(RN_E_UNSUPPORTED, "conv_wino43_wgrad: scheme=%d Cin=%d Cout=%d", scheme, Cin, Cout);
    const int nxi = rn_wino_scheme_nxi(scheme);
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const long long T = (long long)B * th * tw;
    const int cmax = Cin > Cout ? Cin : Cout;
    const long long lim = rn_wino43_plane_limit();
    if ((long long)th * tw * cmax * 4 >= lim)
        return rn_set_error(RN_E_UNSUPPORTED, "conv_wino43_wgrad: one image's transform plane exceeds the 2 GiB buffer window");
    if (T * cmax * 4 >= lim) {                                  // batch chunks (dw accumulates); rows T .. T+31 of a K step lie
        const int chunk = (int)((lim - 1) / ((long long)th * tw * cmax * 4));   // past the plane: offsets < 2^32, read as zeros
        for (int b0 = 0; b0 < B; b0 += chunk) {
            const int nb = B - b0 < chunk ? B - b0 : chunk;
            const int rc = rn_launch_conv_wino43_wgrad(scheme, x + (size_t)b0 * H * W * Cin, dz + (size_t)b0 * H * W * Cout, dw, ws,
                                                       nb, H, W, Cin, Cout, st);
            if (rc != RN_OK) return rc;
        }
        return RN_OK;
    }
    float* V = ws;
    float* dM = V + (size_t)nxi * T * Cin;
    float* dU = dM + (size_t)nxi * T * Cout;
    int rc = rn_launch_wino_input(scheme, x, V, B, H, W, Cin, 1, st);
    if (rc != RN_OK) return rc;
    {
        const unsigned long long n = ((unsigned long long)T * (Cout / 4) + 255) / 256;
        const unsigned nblk8 = (unsigned)((n + 7) / 8 * 8);
        if (scheme == RN_WINO_F43) hipLaunchKernelGGL(wino_dout_kernel<WinoF43>, dim3(nblk8), dim3(256), 0, st, dz, dM, H, W, Cout, th, tw, T, nblk8);
        else hipLaunchKernelGGL(wino_dout_kernel<WinoF44>, dim3(nblk8), dim3(256), 0, st, dz, dM, H, W, Cout, th, tw, T, nblk8);
        rc = rn_check_launch("wino_dout");
        if (rc != RN_OK) return rc;
    }
    W43WgradArgs a;
    a.V = V; a.dM = dM; a.dU = dU; a.parts = dU + (size_t)nxi * Cin * Cout; a.T = T; a.Cin = Cin; a.Cout = Cout;
    a.ciblocks = Cin / 256; a.coblocks = Cout / 256;
    a.ksteps = (int)((T + WBK - 1) / WBK);
    a.v_bytes = (unsigned)(T * Cin * 4); a.m_bytes = (unsigned)(T * Cout * 4); a.u_bytes = (unsigned)((size_t)Cin * Cout * 4);
    const int blocks = nxi * a.ciblocks * a.coblocks;
    // one workgroup per CU takes blocks id, id + 256, ...; the blocks of a last, partial round (or all of them when there are
    // fewer than 256) are cut along the tiles into as many parts as fill the machine once
    static const int forced = getenv("RN_WINO43_WGRAD_SPLIT") ? atoi(getenv("RN_WINO43_WGRAD_SPLIT")) : 0;
    const int rem = blocks % 256;
    int split = 1;
    if (rem > 0) {
        while (rem * split * 2 <= 256 && split * 2 <= a.ksteps) split *= 2;
        if (forced > 0 && rem * forced <= 256) split = forced < a.ksteps ? forced : a.ksteps;
    }
    const int tail = split > 1 ? rem : 0;
    const size_t lds = (size_t)2 * W_STAGE;
    if (blocks - tail > 0) {
        a.item_begin = 0; a.item_end = blocks - tail; a.ksplit = 1; a.steps_per_split = a.ksteps;
        const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(wino43_wgrad_gemm_kernel<false>), lds);
        if (rc_ != RN_OK) return rc_;
        const int n = blocks - tail;
        hipLaunchKernelGGL(wino43_wgrad_gemm_kernel<false>, dim3(n < 256 ? (unsigned)((n + 7) / 8 * 8) : 256u), dim3(512), lds, st, a);
        rc = rn_check_launch("wino43_wgrad_gemm");
        if (rc != RN_OK) return rc;
    }
    if (tail > 0) {
        a.item_begin = blocks - tail; a.item_end = blocks; a.ksplit = split;
        a.steps_per_split = (a.ksteps + split - 1) / split;
        a.ksplit = (a.ksteps + a.steps_per_split - 1) / a.steps_per_split;          // no empty part
        const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(wino43_wgrad_gemm_kernel<true>), lds);
        if (rc_ != RN_OK) return rc_;
        const int n = tail * a.ksplit;
        hipLaunchKernelGGL(wino43_wgrad_gemm_kernel<true>, dim3(n < 256 ? (unsigned)((n + 7) / 8 * 8) : 256u), dim3(512), lds, st, a);
        rc = rn_check_launch("wino43_wgrad_gemm (split)");
        if (rc != RN_OK) return rc;
        hipLaunchKernelGGL(wino43_wgrad_reduce_kernel, dim3((unsigned)(tail * 64)), dim3(256), 0, st, a);
        rc = rn_check_launch("wino43_wgrad_reduce");
        if (rc != RN_OK) return rc;
    }
    {
        const size_t n = (size_t)Cin * (Cout / 4);
        if (scheme == RN_WINO_F43) hipLaunchKernelGGL(wino_dfilter_kernel<WinoF43>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dU, dw, Cin, Cout);
        else hipLaunchKernelGGL(wino_dfilter_kernel<WinoF44>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dU, dw, Cin, Cout);
        return rn_check_launch("wino_dfilter");
    }
}
