// The 3x3x3, 32 -> 32 channel convs of the 3-D encoder (res_block_3d / res1_skip: tools/layer_util.py:60-73, RenderNet_Shader.py:51-64;
// 21 launches per render, and their input gradients) on the bf16 matrix pipe at fp32 accuracy -- the fused counterpart of
// conv_wino_bf3.hip for a layer that is too narrow for the three-launch path (32 channels: V / M round trips would cost more
// than the multiplies they save).
//
// Algorithm: as conv_wino.hip -- in channels-last [B,H,W,D,C] a depth slice of the output is a 2-D conv over (H,W) with 3*C
// input channels (the three depth taps), done as Winograd F(2x2,3x3) over (H,W): Y = A^T [ sum_{dz,c} U[xi][dz][c][n] .* V[xi][dz][c] ] A,
// 16 xi planes, V = B^T d B of the 4x4 input tile.  What changes is the multiply: every fp32 V and U value is the exact sum of
// three bf16 pieces and a product is the six piece products with i + j <= 2 on v_mfma_f32_16x16x32_bf16, fp32 accumulate
// (conv_wino_bf3.hip explains the arithmetic; same error class as an fp32 FMA chain).  The kernel is a template on the operand
// format: C3B3 as described, C3H2 = two fp16 pieces of value / power-of-two tensor scale and three products (24 filter fragments per
// wave, all in the accumulator half; 36 MFMAs per step; scales from max|x| of the tensor, handed over by the producing launch).
//
// Design for gfx950 -- what is different from the fp32 kernel and why:
//   * K is only 96 and N only 32, so per MFMA the FILTER fragments are as large as the data fragments: streaming U through
//     LDS (295 KB per item in split form) would saturate the L2 -> LDS path.  U therefore lives in REGISTERS for the whole
//     persistent kernel: a workgroup is 8 waves, two per SIMD (256 registers each); wave (r, c) owns xi row r, columns 2c and
//     2c + 1 of the 4x4 xi grid: 2 xi x 3 taps x 3 pieces x 2 channel tiles = 36 fragments = 144 registers, pinned to the
//     ACCUMULATOR half of the register file (the MFMAs are issued through inline asm with the "a" constraint: the allocator
//     otherwise shuffles fragments between the two halves in front of every MFMA).
//   * DEPTH RUN: an item is one row of 16 tiles (32 x 2 outputs) of one image walked through ALL depth slices.  Step d
//     fetches input slice d + 2, transforms and splits slice d ONCE (a wave computes exactly what it needs: row r of B^T d, then
//     its two column combinations; v_cvt_pk_bf16_f32 + one subtraction per piece) and feeds it to the three output slices it
//     belongs to -- tap 0 of slice d + 1, tap 1 of slice d, tap 2 of slice d - 1: three accumulator sets (2 xi x 2 channel tiles
//     each, 48 registers), the ring index = output slice modulo 3 is compile-time in the three-fold unrolled loop.  That is 72
//     MFMAs for about 100 vector instructions per wave and step, and the other wave of the SIMD runs its MFMAs meanwhile (bf16
//     MFMA and VALU are separate pipes).  A set is (re)started by the first product of its tap 0 (C = 0), never cleared.
//   * after step d output slice d - 1 is complete: the column half of Y = A^T M A is split over the two waves of a row (each
//     writes its partial sums), the row half goes through a 32-KiB LDS exchange, after which wave w owns (channel tile w / 4,
//     output row (w / 2) % 2, output column w % 2) of every tile and runs the epilogue on 16-byte pixels.  One barrier per step.
//   * raw patch of a step = 34 x 4 pixels x 128 B = 17 KiB, global -> LDS by DMA (8 pixels per wave instruction, SAME padding =
//     out-of-range offsets = hardware zero fill), three stages, counted vmcnt.  Output stores, the residual and the
//     pre-activation go through buffer instructions with a per-item lane offset (out-of-range pixels: offset >= 2^31, dropped).
//   * no compiler hazard padding exists around inline-asm MFMAs: every MFMA carries an s_nop 1 in front (vector write ->
//     MFMA operand read) and the accumulators are read by vector code only behind an explicit wait.
//   * persistent grid, XCD-contiguous item order.
//   * DEPTH SEGMENTS (round 5): with few images the rows of tiles do not fill the 256 workgroups (the res1 layers, 64 x 64 x 32 deep, at
//     batch 6: 384 rows = one full round and a half-empty one); the launcher then cuts every row into 2 .. 8 depth segments, each an item
//     of its own that walks one halo slice more per end (c3_depth_segments).  Same bits for any count (every output slice sums the same
//     taps in the same order); res1 layer at batch 6: 0.196 -> 0.160 ms, batch 1: 0.097 -> 0.047 ms, batch >= 8: one segment, as before.
#include "rn_common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int C3 = 32;                          // channels in and out
constexpr int C3PW = 34, C3PH = 4;              // raw patch of one input depth slice: 34 x 4 pixels (one row of 16 tiles + halo)
constexpr int C3NPIX = C3PW * C3PH;             // 136
constexpr int C3NPIECE = 17;                    // DMA pieces of 8 pixels x 128 B
constexpr int C3STAGE = 18 * 1024;              // bytes per patch stage (17 pieces, padded)
constexpr int C3NSTG = 3;
constexpr int C3XCH = 4 * 2 * 2 * 2 * 1024;     // one exchange buffer: [xi row][q][column half][channel tile] x 1 KiB = 32 KiB (two of them)
constexpr int C3TAB = 256;                      // bias[32], alpha[32]
constexpr unsigned C3OOB = 0x80000000u;
}

struct C3Args {
    const float* x; const char* u; const float* bias; const float* alpha; const float* res; float* y; float* z;
    unsigned x_bytes;
    int B, H, W, D;
    int bh, bw;                 // blocks per image along H (2 output rows each) and W (32 output columns each)
    int nitems;                 // nseg * B * bh * bw: an item is a row of 16 tiles through the depth slices of one SEGMENT
    int nseg, seglen;           // depth segments per row of tiles (1 = the whole depth run) and output slices per segment; see c3_depth_segments
    int act;
    int probe;                  // RN_C3_PROBE (timing experiments, wrong results): 1 no DMA in the loop, 2 no compute, 4 no epilogue, 8 no barriers
    const unsigned* amax_x; const unsigned* amax_u;   // format H2: bit patterns of max|x| of the input tensor and of the filter (device words)
    unsigned* amax_y;                                 // either format, may be null: receives max|y| (atomic maximum onto a zeroed word)
};

// Operand formats (conv_wino_bf3.hip has the arithmetic): B3 = three bf16 pieces, six products; H2 = two fp16 pieces of value / scale,
// three products, the scale a power of two from max|x| of the tensor times the growth bound of the transform (F(2x2,3x3): rows of
// B^T sum to 2 -> 4 for the input, rows of G to 1.5 -> 2.25 for the filter), so that |value / scale| < 2^15.
struct C3B3 {
    static constexpr int NP = 3, NPROD = 6, ID = 0;
    typedef bf16x8 frag;
    static constexpr int PU[6] = {2, 1, 0, 1, 0, 0}, PV[6] = {0, 1, 2, 0, 1, 0};     // i + j <= 2, smallest terms first
};
struct C3H2 {
    static constexpr int NP = 2, NPROD = 3, ID = 1;
    typedef f16x8 frag;
    static constexpr int PU[6] = {1, 0, 0, 0, 0, 0}, PV[6] = {0, 1, 0, 0, 0, 0};
};
constexpr float C3_BOUND_X = 4.f, C3_BOUND_U = 2.25f;

__host__ __device__ inline float c3_h2_scale(float amax, float bound)
{
    const float t = amax * bound * (1.0f / 32768.0f);
    if (!(t > 0.f)) return 1.f;
    int e;
    const float m = frexpf(t, &e);
    return ldexpf(1.f, m == 0.5f ? e - 1 : e);
}

// filter transform U = G g G^T over (k1, k2) per depth tap (double), rounded to fp32, split into three bf16 pieces and stored
// in MFMA A-fragment order: [xi row i][xi column j][depth tap][piece][channel tile nt][lane = (n % 16) + 16 (c / 8)][c % 8],
// n = output channel, c = input channel.  transposed: the filter of the input gradient (taps flipped in all three axes,
// channel roles swapped) -- the same TF tensor read the other way round.
__global__ __launch_bounds__(256)
void conv3d_wino_pack_bf3_kernel(const float* __restrict__ w_tf, unsigned short* __restrict__ us, int transposed)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;            // (xi, dz, c, n)
    if (idx >= 16 * 3 * C3 * C3) return;
    const int n = idx % C3, c = (idx / C3) % C3, dz = (idx / (C3 * C3)) % 3, xi = idx / (C3 * C3 * 3);
    const int i = xi >> 2, j = xi & 3;
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    double acc = 0.0;
    for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) {
            const float w = transposed ? w_tf[((((size_t)(2 - p) * 3 + (2 - q)) * 3 + (2 - dz)) * C3 + n) * C3 + c]
                                       : w_tf[((((size_t)p * 3 + q) * 3 + dz) * C3 + c) * C3 + n];
            acc = __builtin_fma(G[i][p] * G[j][q], (double)w, acc);
        }
    const float u = (float)acc;
    const __bf16 h0 = (__bf16)u;
    const float r1 = u - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    const float r2 = r1 - (float)h1;
    const __bf16 h2 = (__bf16)r2;
    const __bf16 h[3] = {h0, h1, h2};
    const int lane = (n & 15) + 16 * (c >> 3), nt = n >> 4, e = c & 7;
    for (int p = 0; p < 3; ++p)
        us[((((((size_t)i * 4 + j) * 3 + dz) * 3 + p) * 2 + nt) * 64 + lane) * 8 + e] = __builtin_bit_cast(unsigned short, h[p]);
}

// ... in format H2: two fp16 pieces of U / scale, [i][j][depth tap][piece][channel tile][lane][8]
__global__ __launch_bounds__(256)
void conv3d_wino_pack_h2_kernel(const float* __restrict__ w_tf, unsigned short* __restrict__ us, const unsigned* __restrict__ amax, int transposed)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;            // (xi, dz, c, n)
    if (idx >= 16 * 3 * C3 * C3) return;
    const float inv = 1.f / c3_h2_scale(__builtin_bit_cast(float, *amax), C3_BOUND_U);
    const int n = idx % C3, c = (idx / C3) % C3, dz = (idx / (C3 * C3)) % 3, xi = idx / (C3 * C3 * 3);
    const int i = xi >> 2, j = xi & 3;
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    double acc = 0.0;
    for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) {
            const float w = transposed ? w_tf[((((size_t)(2 - p) * 3 + (2 - q)) * 3 + (2 - dz)) * C3 + n) * C3 + c]
                                       : w_tf[((((size_t)p * 3 + q) * 3 + dz) * C3 + c) * C3 + n];
            acc = __builtin_fma(G[i][p] * G[j][q], (double)w, acc);
        }
    const float u = (float)acc * inv;
    const _Float16 h0 = (_Float16)u;
    const _Float16 h1 = (_Float16)(u - (float)h0);
    const _Float16 h[2] = {h0, h1};
    const int lane = (n & 15) + 16 * (c >> 3), nt = n >> 4, e = c & 7;
    for (int p = 0; p < 2; ++p)
        us[((((((size_t)i * 4 + j) * 3 + dz) * 2 + p) * 2 + nt) * 64 + lane) * 8 + e] = __builtin_bit_cast(unsigned short, h[p]);
}

// 8 fp32 values (two registers quads) -> three bf16x8 pieces, round to nearest even, remainders exact
__device__ __forceinline__ void c3_split8(const f32x4 lo, const f32x4 hi, bf16x8 (&p)[3])
{
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        f32x2 v = e < 4 ? f32x2{lo[e], lo[e + 1]} : f32x2{hi[e - 4], hi[e - 3]};
        const bf16x2 h0 = __builtin_convertvector(v, bf16x2);
        v -= __builtin_convertvector(h0, f32x2);
        const bf16x2 h1 = __builtin_convertvector(v, bf16x2);
        v -= __builtin_convertvector(h1, f32x2);
        const bf16x2 h2 = __builtin_convertvector(v, bf16x2);
        p[0][e] = h0[0]; p[0][e + 1] = h0[1];
        p[1][e] = h1[0]; p[1][e + 1] = h1[1];
        p[2][e] = h2[0]; p[2][e + 1] = h2[1];
    }
}

// ... scaled by inv (a power of two) -> two f16x8 pieces
__device__ __forceinline__ void c3_split8(const f32x4 lo, const f32x4 hi, float inv, f16x8 (&p)[2])
{
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        f32x2 v = (e < 4 ? f32x2{lo[e], lo[e + 1]} : f32x2{hi[e - 4], hi[e - 3]}) * inv;
        const f16x2 h0 = __builtin_convertvector(v, f16x2);
        v -= __builtin_convertvector(h0, f32x2);
        const f16x2 h1 = __builtin_convertvector(v, f16x2);
        p[0][e] = h0[0]; p[0][e + 1] = h0[1];
        p[1][e] = h1[0]; p[1][e + 1] = h1[1];
    }
}
__device__ __forceinline__ void c3_split8(const f32x4 lo, const f32x4 hi, float, bf16x8 (&p)[3]) { c3_split8(lo, hi, p); }

// MFMA with the filter fragment pinned to the accumulator half of the register file ("a"); `s_nop 1`: the wait states between a
// vector write and an MFMA operand read that the compiler would insert for a builtin.  _start: C = 0 (restarts a set).
// Format B3: 36 fragments are 144 registers, the accumulator half of a wave at two waves per SIMD has 128 -- the four fragments
// of (tap 0, piece 2), which the first product of a set takes, live in the vector half ("v").  Format H2: 24 fragments, all "a".
__device__ __forceinline__ void c3_mfma(f32x4& c, const bf16x8& u, const bf16x8& p)
{
    asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(u), "v"(p));
}
__device__ __forceinline__ void c3_mfma_start(f32x4& c, const bf16x8& u, const bf16x8& p)
{
    asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(u), "v"(p));
}
__device__ __forceinline__ void c3_mfma(f32x4& c, const f16x8& u, const f16x8& p)
{
    asm("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "a"(u), "v"(p));
}
__device__ __forceinline__ void c3_mfma_start(f32x4& c, const f16x8& u, const f16x8& p)
{
    asm("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(c) : "a"(u), "v"(p));
}

// A ds_read_b128 is conflict-free when the 16 lanes of a read group (here: the 16 tiles, at one channel quarter) hit 16 different
// 16-byte bank groups of a 256-byte window.  A pixel is a 128-byte slot and a tile advances by two pixels, so the XOR swizzle of
// the chunk index alone reaches only 8 of the 16: pixels 16..33 of a patch row additionally swap places with their neighbour
// (slot = px ^ 1), which puts tiles 8..15 into the other half of the window (one 2-way collision is left in the columns b >= 2).
__device__ __forceinline__ int c3_slot_swap(int px) { return ((px >> 4) | (px >> 5)) & 1; }

#define C3_WAIT_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")

// PROBE (timing experiments, wrong results; RN_C3_PROBE): 1 no DMA in the loop, 2 no arithmetic, 4 no epilogue, 8 no barriers
template <int PROBE, class F, bool AMAX>
__global__ __launch_bounds__(512, 1)
void conv3d_wino_bf3_kernel(const C3Args a)
{
    typedef typename F::frag frag;
    constexpr int NP = F::NP;
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [patch stage 0 | 1 | 2][exchange 0 | 1][bias, alpha]
    typedef __attribute__((address_space(3))) void lds_void;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = wave >> 1, c = wave & 1;                            // xi row, xi column pair
    const int tx = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(a.z ? a.z : a.y, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.x), 0, a.x_bytes, 0x00020000);

    // ---- this wave's filter fragments [column jj of its pair][tap][piece][channel tile], xi = (r, 2 c + jj)
    frag U[2][3][NP][2];
    {
        const frag* up = reinterpret_cast<const frag*>(a.u) + (size_t)(r * 4 + 2 * c) * (3 * NP * 2 * 64) + lane;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int dz = 0; dz < 3; ++dz)
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) U[jj][dz][p][nt] = up[(((jj * 3 + dz) * NP + p) * 2 + nt) * 64];
    }
    // format H2: 1 / scale of the input tensor for the split, scale_x * scale_u for the way out (powers of two: exact)
    float inv_x = 1.f, out_scale = 1.f;
    if constexpr (F::ID == 1) {
        const float sx = c3_h2_scale(__builtin_bit_cast(float, *a.amax_x), C3_BOUND_X);
        inv_x = 1.f / sx;
        out_scale = sx * c3_h2_scale(__builtin_bit_cast(float, *a.amax_u), C3_BOUND_U);
    }
    float ymax = 0.f;                                                 // max |y| of this lane's outputs (a.amax_y)
    float* tab = reinterpret_cast<float*>(smem + C3NSTG * C3STAGE + 2 * C3XCH);
    if (tid < C3) {
        tab[tid] = a.bias ? a.bias[tid] : 0.f;
        tab[C3 + tid] = a.alpha ? a.alpha[tid] : 0.f;
    }
    // ---- row r of B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]: t = d[rowA] + sgn * d[rowB]
    const int rowA = (0x1210 >> (4 * r)) & 3, rowB = (0x3122 >> (4 * r)) & 3;
    const float sgn = r == 1 ? 1.f : -1.f;
    // the wave needs patch columns b = c + k, k = 0..2, of each tile: pixel (py, px = 2 tx + b) sits at (py * 34 + px) * 128, the
    // lane's 8 channels = logical 16-byte chunks 2 kq + h, physical chunk = logical ^ ((px >> 1) & 7), in slot px ^ c3_slot_swap(px)
    unsigned fo[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int b = c + k;
            const int px = 2 * tx + b;
            fo[k][h] = (unsigned)((px ^ c3_slot_swap(px)) * 128) + ((((unsigned)(2 * kq + h)) ^ (unsigned)((px >> 1) & 7)) << 4);
        }
    const unsigned offA = (unsigned)(rowA * C3PW * 128), offB = (unsigned)(rowB * C3PW * 128);

    const int G = gridDim.x;
    const unsigned pix_bytes = (unsigned)a.D * (C3 * 4);
    // DMA piece wave + 8 i of a stage (17 pieces: i = 2 only on wave 0) holds pixels q = 8 piece + lane / 8 (q = py * 34 + px) in
    // 128-byte slots; the lane fetches LOGICAL chunk (lane % 8) ^ ((px >> 1) & 7) into slot lane % 8
    const bool three = wave == 0;
    // this wave's share of every flush: channel tile ent, output pixel (dy, dx) of the 2x2 tile; C/D layout of the 16x16 MFMA with
    // the filter as A: a lane holds channels 4 kq .. 4 kq + 3 of tile tx -> 16-byte pixels
    const int ent = wave >> 2, dy = (wave >> 1) & 1, dx = wave & 1;
    const int nch = ent * 16 + 4 * kq;
    __syncthreads();

    f32x4 acc[3][2][2];                                               // [output slice % 3][jj][channel tile]
    const int perm = (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);       // XCD-contiguous slot in a round (G % 8 == 0)
    for (int id = perm; id < a.nitems; id += G) {
        // ---- item: block column bx, block row by, image bi, depth segment seg = output slices o0 .. o1 - 1.  It walks the INPUT slices
        // ds = max(o0 - 1, 0) .. dl = min(o1, D - 1); the ring indices (stage, accumulator set) count from ds.  With o0 > 0 the first
        // step also feeds the sets of output slices o0 - 2 and o0 - 1 -- never flushed, like the sets that slice o1 restarts.
        const int bx = id % a.bw, by = (id / a.bw) % a.bh, bi = (id / (a.bw * a.bh)) % a.B, seg = id / (a.bw * a.bh * a.B);
        const int o0 = seg * a.seglen, o1 = o0 + a.seglen < a.D ? o0 + a.seglen : a.D;
        if (o0 >= o1) continue;                                                       // (D not a multiple of the segment count: uniform for the workgroup)
        const int ds = o0 > 0 ? o0 - 1 : 0, dl = o1 < a.D ? o1 : a.D - 1;
        const int y0 = by * 2 - 1, x0 = bx * 32 - 1;                                  // patch origin (may be -1: SAME padding)
        // byte offset of depth slice 0 of the patch origin; wraps below zero for the first row / column: all sums are modulo
        // 2^32 and only used where the pixel is inside
        const unsigned base = (unsigned)((bi * a.H + y0) * a.W + x0) * pix_bytes;
        unsigned roff[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = (wave + 8 * i) * 8 + (lane >> 3);
            const int py = (q * 241) >> 13, pxs = q - py * C3PW;                       // q / 34 for q < 256; pxs: the SLOT's column
            const int px = pxs ^ c3_slot_swap(pxs);                                    // the pixel that lives in it
            const bool ok = q < C3NPIX && (unsigned)(y0 + py) < (unsigned)a.H && (unsigned)(x0 + px) < (unsigned)a.W;
            roff[i] = ok ? base + (unsigned)(py * a.W + px) * pix_bytes + (unsigned)((((lane & 7) ^ ((px >> 1) & 7))) << 4) : C3OOB;
        }
        const int oy = by * 2 + dy, ox = bx * 32 + 2 * tx + dx;
        const unsigned ooff = (oy < a.H && ox < a.W) ? (unsigned)((bi * a.H + oy) * a.W + ox) * pix_bytes + (unsigned)(nch * 4) : C3OOB;

        // input slice d -> stage d % 3 (an out-of-range pixel stays out of range: C3OOB + d * 128 >= 2^31)
        auto issue = [&](int d, int stage) {
            if (PROBE & 1) return;
            char* sb = smem + stage * C3STAGE;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (i == 2 && !three) break;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)(sb + (wave + 8 * i) * 1024), 16,
                                                         roff[i] + (unsigned)(d * (C3 * 4)), 0, 0, 0);
            }
        };
        // wait until only the newest fetch of this wave (3 | 2 DMAs) may be outstanding, then the workgroup barrier
        auto wait_newest = [&](bool issued) {
            if (PROBE & 8) return;
            if (!issued) C3_WAIT_BARRIER(0);
            else if (three) C3_WAIT_BARRIER(3);
            else C3_WAIT_BARRIER(2);
        };
        issue(ds, 0);
        if (ds + 1 <= dl) issue(ds + 1, 1);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[0][jj][nt] = f32x4{0.f, 0.f, 0.f, 0.f};       // output slice 0 has no tap-0 step (o0 > 0: the set of slice o0 - 1, unused)
        wait_newest(ds + 1 <= dl);                                    // slice ds has landed (slice ds + 1 may still be in flight)

        // step d (S = d % 3 at compile time): see the file comment.  `fl` = output slice d - 1 exists and is flushed.
        auto step = [&](auto sc, int d) {
            constexpr int S = decltype(sc)::value;                    // (d - ds) % 3: stage of slice d; accumulator set of OUTPUT slice d
            constexpr int SP = (S + 2) % 3, SN = (S + 1) % 3;         // ... of output slices d - 1 and d + 1
            const bool fl = d > o0 && !(PROBE & 4);                   // output slice d - 1 belongs to this segment
            const bool arith = d <= dl && !(PROBE & 2);               // d == D: the pass that only flushes output slice D - 1
            // the flush's own load (residual of slice d - 1) goes out FIRST, then the fetch of slice d + 2: the flush can then wait
            // for everything but that fetch
            f32x4 rv = {0.f, 0.f, 0.f, 0.f};
            if (fl && a.res) rv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ooff, (d - 1) * (C3 * 4), 0));
            asm volatile("" ::: "memory");
            const bool issued = d + 2 <= dl;
            if (issued) issue(d + 2, SP);                             // stage (d + 2) % 3 = (d - 1) % 3: read during step d - 1, free since its barrier
            if (arith) {
                const char* sb = smem + S * C3STAGE;
                f32x4 v[2][2];                                        // [jj][h]: the lane's channels 4 h .. 4 h + 3 of its 8
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 t[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const f32x4 dA = *reinterpret_cast<const f32x4*>(sb + fo[k][h] + offA);
                        const f32x4 dB = *reinterpret_cast<const f32x4*>(sb + fo[k][h] + offB);
                        t[k] = dA + sgn * dB;
                    }
                    // columns of B: v0 = t0 - t2, v1 = t1 + t2, v2 = t2 - t1, v3 = t1 - t3; t[k] is column c + k
                    if (c == 0) { v[0][h] = t[0] - t[2]; v[1][h] = t[1] + t[2]; }
                    else        { v[0][h] = t[1] - t[0]; v[1][h] = t[0] - t[2]; }
                }
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    frag p[NP];
                    c3_split8(v[jj][0], v[jj][1], inv_x, p);
                    // the piece products of the format, smallest terms first;
                    // input slice d is tap 2 of output d - 1, tap 1 of output d, tap 0 of output d + 1 (whose set it restarts)
#pragma unroll
                    for (int k = 0; k < F::NPROD; ++k)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            c3_mfma(acc[SP][jj][nt], U[jj][2][F::PU[k]][nt], p[F::PV[k]]);
                            c3_mfma(acc[S][jj][nt], U[jj][1][F::PU[k]][nt], p[F::PV[k]]);
                            if (k == 0) c3_mfma_start(acc[SN][jj][nt], U[jj][0][F::PU[k]][nt], p[F::PV[k]]);
                            else c3_mfma(acc[SN][jj][nt], U[jj][0][F::PU[k]][nt], p[F::PV[k]]);
                        }
                }
            }
            // flush of output slice d - 1 (set SP).  Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]].  Column half: c0 = m0 + m1 + m2,
            // c1 = m1 - m2 - m3, split over the two waves of the row: (m0 + m1 | m1) and (m2 | -m2 - m3) -> exchange [r][q][c][tile]
            char* xch = smem + C3NSTG * C3STAGE + (d & 1) * C3XCH;
            if (fl) {
                // MFMA results -> vector reads (the operands order the MFMAs of this step in front of the wait)
                asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[SP][0][0]), "+v"(acc[SP][0][1]), "+v"(acc[SP][1][0]), "+v"(acc[SP][1][1]) :: "memory");
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const f32x4 m0 = acc[SP][0][nt], m1 = acc[SP][1][nt], s = m0 + m1;
                    const f32x4 p0 = c == 0 ? s : m0, p1 = c == 0 ? m1 : -s;
                    *reinterpret_cast<f32x4*>(xch + ((((r * 2 + 0) * 2 + c) * 2 + nt) * 64 + lane) * 16) = p0;
                    *reinterpret_cast<f32x4*>(xch + ((((r * 2 + 1) * 2 + c) * 2 + nt) * 64 + lane) * 16) = p1;
                }
            }
            // one barrier: the exchange is written, the residual is in, slice d + 1 has landed (older than the fetch just issued),
            // and every wave is done reading stage S
            wait_newest(issued);
            if (fl) {
                // rows of A^T: dy = 0: c0 + c1 + c2;  dy = 1: c1 - c2 - c3
                f32x4 ci[3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    ci[k] = *reinterpret_cast<const f32x4*>(xch + (((((dy + k) * 2 + dx) * 2 + 0) * 2 + ent) * 64 + lane) * 16)
                          + *reinterpret_cast<const f32x4*>(xch + (((((dy + k) * 2 + dx) * 2 + 1) * 2 + ent) * 64 + lane) * 16);
                f32x4 o = (dy == 0 ? (ci[0] + ci[1]) + ci[2] : (ci[0] - ci[1]) - ci[2]);
                if constexpr (F::ID == 1) o *= out_scale;
                o += *reinterpret_cast<const f32x4*>(tab + nch);
                // (the slice offset goes into the VECTOR offset: with a scalar-register offset the compiler assumes that a 16-byte
                // store's data registers may be overwritten by the very next instruction and puts no wait state behind the store --
                // on gfx950 that lost the first dword of lanes 12..15 of every row, now and then)
                const unsigned so = ooff + (unsigned)((d - 1) * (C3 * 4));
                if (a.z) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), zrsrc, so, 0, 0);
                if (a.act & RN_ACT_PRELU) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(tab + C3 + nch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f) + av[e] * fminf(o[e], 0.f);
                }
                if (a.act & RN_ACT_ELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : expf(o[e]) - 1.f;
                }
                if (a.res) o += rv;
                if (a.act & RN_ACT_SIGMOID) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = 1.f / (1.f + __expf(-o[e]));
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, so, 0, 0);
                if (AMAX && ooff != C3OOB) ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
            }
        };
        // input slices ds .. o1 (the last one flushes output slice o1 - 1; with o1 = D it is the pass that only flushes)
        for (int d = ds; d <= o1; d += 3) {
            step(std::integral_constant<int, 0>{}, d);
            if (d + 1 <= o1) step(std::integral_constant<int, 1>{}, d + 1);
            if (d + 2 <= o1) step(std::integral_constant<int, 2>{}, d + 2);
        }
        // the next item's first fetch overwrites stage 0, which the last arithmetic step may still be read from by a slower
        // wave: every wave has passed the barrier of the last step, which comes after that step's arithmetic -- safe.
    }
    if (AMAX) {                                                       // max |y| for a consumer in format H2: one atomic per wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o));
        if (lane == 0) atomicMax(a.amax_y, __builtin_bit_cast(unsigned, ymax));
    }
#endif
}

bool rn_conv3d_wino_bf3_supported(int Cin, int Cout)
{
    static const bool off = getenv("RN_NO_WINOGRAD3D_BF3") != nullptr || getenv("RN_NO_WINOGRAD3D") != nullptr || getenv("RN_NO_WINOGRAD") != nullptr;
    return !off && Cin == C3 && Cout == C3;
}

// fmt 0: three bf16 pieces (6 bytes per filter element); fmt 1: two fp16 pieces of U / scale (4 bytes) + a 256-byte tail whose first
// word is max|w| (bit pattern)
size_t rn_conv3d_wino_bf3_packed_bytes() { return (size_t)16 * 3 * C3 * C3 * 3 * 2; }
size_t rn_conv3d_wino_split_packed_bytes(int fmt) { return fmt == 1 ? (size_t)16 * 3 * C3 * C3 * 2 * 2 + 256 : rn_conv3d_wino_bf3_packed_bytes(); }

int rn_launch_conv3d_wino_pack_bf3(const float* w_tf, void* us, int transposed, hipStream_t st)
{
    return rn_launch_conv3d_wino_split_pack(0, w_tf, us, transposed, st);
}

int rn_launch_conv3d_wino_split_pack(int fmt, const float* w_tf, void* us, int transposed, hipStream_t st)
{
    if (fmt == 1) {
        unsigned* amax = reinterpret_cast<unsigned*>(static_cast<char*>(us) + (size_t)16 * 3 * C3 * C3 * 2 * 2);
        const int rc = rn_launch_absmax(w_tf, (size_t)27 * C3 * C3, amax, st);
        if (rc != RN_OK) return rc;
        hipLaunchKernelGGL(conv3d_wino_pack_h2_kernel, dim3((16 * 3 * C3 * C3 + 255) / 256), dim3(256), 0, st, w_tf,
                           static_cast<unsigned short*>(us), amax, transposed);
        return rn_check_launch("conv3d_wino_pack_h2");
    }
    hipLaunchKernelGGL(conv3d_wino_pack_bf3_kernel, dim3((16 * 3 * C3 * C3 + 255) / 256), dim3(256), 0, st, w_tf,
                       static_cast<unsigned short*>(us), transposed);
    return rn_check_launch("conv3d_wino_pack_bf3");
}

int rn_launch_conv3d_wino_bf3(const float* x, const void* us, const float* bias, const float* alpha, const float* residual,
                              float* y, float* preact, int B, int H, int W, int D, int act, hipStream_t st)
{
    return rn_launch_conv3d_wino_split(0, x, us, bias, alpha, residual, y, preact, B, H, W, D, act, nullptr, nullptr, nullptr, st);
}

// Depth segments per row of tiles.  A row of 16 tiles walked through all D slices is one item; with few images the rows do not fill the
// 256 persistent workgroups (res1 layers, 64 x 64 maps 32 deep: batch 3 = 192 rows = 192 busy CUs for the whole launch, batch 6 = 384 = a
// full round and a half-empty one).  A segment of L output slices costs L + 2 steps (a halo slice per end; L + 1 for the whole run) + about
// one step of item start-up -- fitted to the res1 layer at batch 1 .. 12 x 1 / 2 / 4 / 8 segments (profiles/r05o_depth_segments_layer.txt:
// batch 3: 0.104 / 0.113 / 0.098 / 0.112 ms, batch 6: 0.196 / 0.162 / 0.174 / 0.203); the count that minimises rounds x steps wins:
// 1 from batch 8 on, 2 at batch 6, 4 at batch 3.  RN_C3_DEPTH_SEGMENTS = n forces n (tests, timing).
static int c3_depth_segments(long long rows, int D)
{
    static const int forced = getenv("RN_C3_DEPTH_SEGMENTS") ? atoi(getenv("RN_C3_DEPTH_SEGMENTS")) : 0;
    if (forced >= 1) return forced < D ? forced : D;
    int best = 1;
    long long best_cost = 0;
    for (int n = 1; n <= 8 && n <= D; n *= 2) {
        const int len = (D + n - 1) / n;
        const long long rounds = (rows * n + 255) / 256;
        const long long cost = rounds * (len + (n > 1 ? 2 : 1) + 1);
        if (n == 1 || cost < best_cost) { best = n; best_cost = cost; }
    }
    return best;
}

// fmt 1 (H2): amax_x = device word with the bit pattern of max|x| (or of a bound), or null -> a pass over x into `scratch_amax` (a device
// word the caller provides; required when amax_x is null).  amax_y (either format, may be null): receives max|y|.
int rn_launch_conv3d_wino_split(int fmt, const float* x, const void* us, const float* bias, const float* alpha, const float* residual,
                                float* y, float* preact, int B, int H, int W, int D, int act, const unsigned* amax_x, unsigned* scratch_amax,
                                unsigned* amax_y, hipStream_t st)
{
    if (B < 1 || H < 1 || W < 1 || D < 1) return rn_set_error(RN_E_INVALID, "conv3d_wino_bf3: bad sizes");
    if ((act & RN_ACT_PRELU) && !alpha) return rn_set_error(RN_E_INVALID, "conv3d_wino_bf3: PReLU needs alpha");
    if (fmt < 0 || fmt > 1) return rn_set_error(RN_E_INVALID, "conv3d_wino_bf3: operand format %d", fmt);
    const size_t per_image = (size_t)H * W * D * C3 * 4;
    if (per_image >= 0x7fffff00ULL) return rn_set_error(RN_E_UNSUPPORTED, "conv3d_wino_bf3: one image exceeds the 2 GiB buffer window");
    const int chunk = (int)(0x7fffff00ULL / per_image);              // images per launch: byte offsets stay below 2^31 (the top bit = zero fill)
    const size_t lds = (size_t)C3NSTG * C3STAGE + 2 * C3XCH + C3TAB;
    static const int probe = getenv("RN_C3_PROBE") ? atoi(getenv("RN_C3_PROBE")) : 0;
    void (*kern)(const C3Args) = fmt == 1 ? (amax_y ? conv3d_wino_bf3_kernel<0, C3H2, true> : conv3d_wino_bf3_kernel<0, C3H2, false>)
                                          : (amax_y ? conv3d_wino_bf3_kernel<0, C3B3, true> : conv3d_wino_bf3_kernel<0, C3B3, false>);
    if (probe && fmt == 0 && !amax_y) {
        switch (probe) {
            case 1: kern = conv3d_wino_bf3_kernel<1, C3B3, false>; break;
            case 2: kern = conv3d_wino_bf3_kernel<2, C3B3, false>; break;
            case 4: kern = conv3d_wino_bf3_kernel<4, C3B3, false>; break;
            case 6: kern = conv3d_wino_bf3_kernel<6, C3B3, false>; break;
            case 7: kern = conv3d_wino_bf3_kernel<7, C3B3, false>; break;
            case 15: kern = conv3d_wino_bf3_kernel<15, C3B3, false>; break;
            default: break;
        }
    }
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds); if (rc_ != RN_OK) return rc_; }
    if (fmt == 1 && !amax_x) {
        if (!scratch_amax) return rn_set_error(RN_E_INVALID, "conv3d_wino_bf3: format H2 needs max|x| or a word to gather it in");
        const int rc = rn_launch_absmax(x, (size_t)B * H * W * D * C3, scratch_amax, st);
        if (rc != RN_OK) return rc;
        amax_x = scratch_amax;
    }
    if (amax_y) { const int rc = rn_launch_word(amax_y, nullptr, st); if (rc != RN_OK) return rc; }
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = B - b0 < chunk ? B - b0 : chunk;
        const size_t off = (size_t)b0 * H * W * D * C3;
        C3Args a;
        a.x = x + off; a.u = static_cast<const char*>(us); a.bias = bias; a.alpha = alpha; a.res = residual ? residual + off : nullptr;
        a.y = y + off; a.z = preact ? preact + off : nullptr;
        a.x_bytes = (unsigned)(nb * per_image);
        a.B = nb; a.H = H; a.W = W; a.D = D;
        a.bh = (H + 1) / 2; a.bw = (W + 31) / 32;
        const long long rows = (long long)nb * a.bh * a.bw;
        a.nseg = c3_depth_segments(rows, D);               // <= 8, or what RN_C3_DEPTH_SEGMENTS forces (<= D)
        a.seglen = (D + a.nseg - 1) / a.nseg;
        const long long nitems = rows * a.nseg;
        if (nitems > 0x7fffffffLL) return rn_set_error(RN_E_UNSUPPORTED, "conv3d_wino_bf3: too many items (%lld rows x %d depth segments)", rows, a.nseg);
        a.nitems = (int)nitems; a.act = act;
        a.probe = probe;
        a.amax_x = amax_x; a.amax_y = amax_y;
        a.amax_u = fmt == 1 ? reinterpret_cast<const unsigned*>(static_cast<const char*>(us) + (size_t)16 * 3 * C3 * C3 * 2 * 2) : nullptr;
        const unsigned grid = nitems < 256 ? (unsigned)((nitems + 7) / 8 * 8) : 256u;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
        const int rc = rn_check_launch("conv3d_wino_bf3");
        if (rc != RN_OK) return rc;
    }
    return RN_OK;
}
