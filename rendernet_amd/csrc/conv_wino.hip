// Winograd F(2x2,3x3) convolution on the gfx950 matrix cores, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// Serves the stride-1 3x3 SAME 2-D convs of the RenderNet trunk -- res_block_2d / the *_skip convs,
// tools/layer_util.py:101-104, RenderNet_Shader.py:71-84,91-99 (86.9 % of the path's FLOPs) -- and
// their input gradients.  The direct implicit-GEMM kernel (conv_igemm.hip) already runs at 0.90 of
// the fp32 MFMA peak on these layers, so the only lever left is to issue fewer MFMAs: F(2x2,3x3)
// computes a 2x2 output tile from a 4x4 input tile with 16 multiplies per (cin, cout) pair instead
// of 36 (2.25x fewer), still in fp32.
//
//     Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A          per 2x2 output tile and output channel
//
// One fused kernel, no transformed tensors in HBM:
//   * filters are pre-transformed once by rn_pack_weights (RN_PACK_CONV_WINO): U[xi][c][n], 16 "xi"
//     planes, stored [Cout/32][Cin/16][16 xi][4][32][4] so that every (n-block, 16-channel step) is one
//     contiguous 32 KiB piece that goes global -> LDS with buffer_load ... lds (Cout % 32 != 0: 16-wide n-blocks);
//   * a workgroup (512 threads = 8 waves, two per SIMD, 128 accumulator registers each) owns a block of
//     16x8 tiles (32x16 outputs) x 32 output channels x all 16 xi.  Its RAW 34x18-pixel input patch goes
//     global -> LDS, 16 channels per stage (SAME padding from the buffer bounds check);
//   * the input transform B^T d B happens at fragment-read time: wave w owns tile row w; a lane reads the
//     16 pixels of its tile (ds_read_b128 = 4 channels each) and 32 vector adds give the 16 xi fragments;
//     no two waves transform the same tile.  fp32 MFMA runs at the fp32 VECTOR rate and the adds measurably
//     take issue time from it, so they are v_pk_add_f32 (one packed add per MFMA pair);
//   * per xi the wave multiplies its 16 tiles x 32 channels (two 16x16 MFMA tiles), so all 16 xi of a
//     (tile, channel) sit in one lane and the output transform A^T M A is a per-lane sum; the epilogue
//     (bias, PReLU, residual, pre-activation) applies to the 2x2 outputs directly.
#include "rn_common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WinoArgs {
    const float* x; const float* u; const float* bias; const float* alpha; const float* res; float* y; float* z;
    unsigned x_bytes, u_bytes;
    int B, H, W, D, Cin, Cout;  // D = 1: 2-D conv.  D > 1: 3x3x3 conv over [B,H,W,D,Cin], Winograd in (H,W), direct in D
    int KD;                     // depth taps (1 or 3)
    int bh, bw;                 // 16x8-tile blocks per image along H (16 rows each) and W (32 columns each)
    int mblocks, nblocks;       // B*D*bh*bw, Cout/32
    int spt;                    // 16-channel steps per depth tap = Cin/16; the K loop has KD*spt steps (x4 sub-filters for 4x4)
    int pad;                    // pad_lo of the conv: 1 (3x3, 4x4 SAME) or 2 (4x4 flipped conv of a stride-1 transposed conv)
    int act;
};

namespace {
constexpr int WPW = 34, WPH = 18;         // patch: 32+2 columns, 16+2 rows
constexpr int WNPIX = WPW * WPH;          // 612
constexpr int WRAW_PIECES = 40;           // 1 KiB DMA pieces of 16 pixels x 64 B (612 -> 640 pixel slots: 5 per wave)
constexpr int WRAW_B = WRAW_PIECES * 1024;   // bytes per raw stage (40 960)
constexpr unsigned WOOB = 0x80000000u;
// MODE 0: F(2x2,3x3), 16 xi planes, 4x4 input tiles.  MODE 1: a 4x4 filter as four 2x2 sub-filters, each F(2x2,2x2): 9 xi
// planes, 3x3 input tiles, the sub-filters are four consecutive K steps that read the patch shifted by (2a, 2b) pixels.
// MODE 2: a 4x4 STRIDE-2 transposed conv (slim.conv2d_transpose, RenderNet_Shader.py:105-119: e_conv7, e_conv8, e_conv9): output
// phase (pa, pb) = pixels (2m+pa, 2n+pb) is a 2x2 conv of the input, y[2m+pa] = x[m-1+pa] w[3-pa] + x[m+pa] w[1-pa] per axis, so
// every phase is one F(2x2,2x2) conv (9 multiplies per 4 outputs instead of 16) with its own sub-filter and its own output
// pixels: the four phases are four ITEMS of one launch (the phase rides in WinoBlock::dz), each with Cin/16 K steps.
constexpr int wino_nxi(int mode) { return mode ? 9 : 16; }
constexpr int wino_upieces(int mode, int nt) { return wino_nxi(mode) * nt; }          // 1-KiB filter pieces per step
constexpr int wino_upw(int mode, int nt) { return (wino_upieces(mode, nt) + 7) / 8; }   // ... per wave
constexpr int wino_ustage(int mode, int nt) { return wino_upw(mode, nt) * 8 * 1024; }    // bytes per filter stage
}

static size_t wino_lds_bytes(int mode, int nt) { return (size_t)2 * WRAW_B + 2 * (size_t)wino_ustage(mode, nt); }

__device__ __forceinline__ f32x4 pk_add(f32x4 x, f32x4 y)
{
    f32x4 r;
    asm("v_pk_add_f32 %0, %2, %3\n\tv_pk_add_f32 %1, %4, %5"
        : "=&v"(*reinterpret_cast<double*>(&r)), "=&v"(*(reinterpret_cast<double*>(&r) + 1))
        : "v"(*reinterpret_cast<double*>(&x)), "v"(*reinterpret_cast<double*>(&y)),
          "v"(*(reinterpret_cast<double*>(&x) + 1)), "v"(*(reinterpret_cast<double*>(&y) + 1)));
    return r;
}
__device__ __forceinline__ f32x4 pk_sub(f32x4 x, f32x4 y)
{
    f32x4 r;
    asm("v_pk_add_f32 %0, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %4, %5 neg_lo:[0,1] neg_hi:[0,1]"
        : "=&v"(*reinterpret_cast<double*>(&r)), "=&v"(*(reinterpret_cast<double*>(&r) + 1))
        : "v"(*reinterpret_cast<double*>(&x)), "v"(*reinterpret_cast<double*>(&y)),
          "v"(*(reinterpret_cast<double*>(&x) + 1)), "v"(*(reinterpret_cast<double*>(&y) + 1)));
    return r;
}

// One block of work: which tiles / channels / K steps, and the per-lane DMA offsets that go with it.
struct WinoBlock {
    int by, bx, dz, b, nb;      // 16x8-tile block (rows by*16.., columns bx*32..), depth slice, batch item, n-block
    int s_begin, s_end;         // K steps [s_begin, s_end) of 16 channels
    unsigned roff[5];           // raw-patch DMA: per-lane byte offset of piece wave + 8 i at channel step 0.  MODE 0: WOOB
                                // where SAME padding applies (zero fill).  MODE 1: the offset for sub-filter 0, unmasked --
                                // sub-filter (a, b) adds the constant (2a*W + 2b)*pix_bytes and tests its bit of vmask
    unsigned vmask;             // MODE 1: bit sub*5 + i = piece i's pixel is inside the image for sub-filter `sub`
    unsigned uoff;              // filter DMA: per-lane byte offset of piece `wave` at step 0
};

// work item `id` (0 <= id < mblocks*nblocks) -> block.  Enumeration e: groups of 8 m-blocks, n-major inside a group, so
// that the 32 workgroups of one XCD (id % 8; one workgroup per CU, persistent, round r works on id = r*G + blockIdx)
// stream 4 filter slabs and 8 patches between them, and the 8 XCDs of a round of 256 read the same 8 patches.
template <int NT, int MODE>
__device__ __forceinline__ void wino_block(const WinoArgs& a, int id, int wave, int lane, WinoBlock& k)
{
    constexpr int NSUB = MODE == 1 ? 4 : 1;
    const int T = a.mblocks * a.nblocks;
    int e = id;
    if (id < (T & ~255)) { const int s = id >> 3; e = (s >> 5) * 256 + (id & 7) * 32 + (s & 31); }
    const int per = 8 * a.nblocks;
    const int g = e / per, full = a.mblocks >> 3;
    int rem = e - g * per, gs = 8, g0 = g;
    if (g >= full) { rem = e - full * per; gs = a.mblocks - full * 8; g0 = full; }
    k.nb = rem / gs;
    const int mb = g0 * 8 + rem % gs;
    k.bx = mb % a.bw; k.by = (mb / a.bw) % a.bh; k.dz = (mb / (a.bw * a.bh)) % a.D; k.b = mb / (a.bw * a.bh * a.D);
    // 3-D: output depth slice dz reads the KD*Cin contiguous floats of input depths dz-1..dz+1 at every (h, w) -- in
    // channels-last [B,H,W,D,C] the conv IS a 2-D conv with 3*Cin channels per depth slice.  A depth tap outside the
    // volume (SAME padding) is a run of `spt` whole steps of zeros: those steps are skipped.
    k.s_begin = (a.KD == 3 && k.dz == 0) ? a.spt : 0;
    k.s_end = a.KD * a.spt * NSUB - ((a.KD == 3 && k.dz == a.D - 1) ? a.spt : 0);
    // (MODE 2: D = 4 output phases, the input has no such axis; phase (pa, pb) reads from pixel (m - 1 + pa, n - 1 + pb))
    const unsigned pix_bytes = (MODE == 2 ? 1u : (unsigned)a.D) * (unsigned)a.Cin * 4u;
    const unsigned win_off = MODE == 2 ? 0u : (unsigned)((k.dz - (a.KD == 3 ? 1 : 0)) * a.Cin * 4);   // may wrap below 0: only used with s >= s_begin
    const int pady = MODE == 2 ? 1 - (k.dz >> 1) : a.pad, padx = MODE == 2 ? 1 - (k.dz & 1) : a.pad;
    // raw-patch DMA: piece p = wave + 8 i (i < 5) holds pixels q = 16 p + lane/4 (q = py*34 + px); the lane fetches
    // LOGICAL chunk (lane%4) ^ swz(px) into physical slot lane%4, swz(px) = (px>>1)&3 (two lanes of a ds_read_b128
    // group at most share a 16-B slot)
    k.vmask = 0u;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int p = wave + 8 * i;
        const int q = p * 16 + (lane >> 2);
        const int py = q / WPW, px = q - py * WPW;
        const int iy = k.by * 16 - pady + py, ix = k.bx * 32 - padx + px;
        const unsigned o = (unsigned)((k.b * a.H + iy) * a.W + ix) * pix_bytes + win_off + (unsigned)(((lane & 3) ^ ((px >> 1) & 3)) * 16);
        if (MODE != 1) {
            const bool ok = q < WNPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            k.roff[i] = ok ? o : WOOB;
        } else {
            k.roff[i] = o;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const bool ok = q < WNPIX && (unsigned)(iy + 2 * (sub >> 1)) < (unsigned)a.H && (unsigned)(ix + 2 * (sub & 1)) < (unsigned)a.W;
                k.vmask |= ok ? (1u << (sub * 5 + i)) : 0u;
            }
        }
    }
    // filter DMA: the piece of (nb, step) is lane-linear, 1 KiB per instruction; wave w moves pieces w, w + 8, ...
    k.uoff = ((unsigned)(MODE == 2 ? k.dz * a.nblocks + k.nb : k.nb) * (unsigned)(a.KD * a.spt * NSUB)) * (1024u * wino_upieces(MODE, NT)) +
             (unsigned)wave * 1024u + (unsigned)lane * 16u;
}

// PROBE (measurement switches, RN_WINO_PROBE; 0 = the product kernel): 1 = skip the input transform (wrong results),
// 2 = no DMA inside the loop (wrong results), 4 = input transform with plain v_add/v_sub instead of v_pk_add_f32,
// 8 = all DMAs of a step at its top instead of interleaved with the MFMAs, 64 = no epilogue, 128 = no per-step barrier.
//
// Persistent: the grid is one workgroup per CU; workgroup g works on items g, g + G, g + 2G, ...  The K loop runs
// straight across item boundaries: during the LAST step of an item the first step of the NEXT item is fetched into the
// free LDS stage, so the epilogue (output transform + stores) and the next item's cold start overlap its latency --
// with 6 K steps per item (the 3-D encoder layers) prologue + epilogue used to cost as much as the steps themselves.
// NT = 16-channel n-tiles per wave: 2 (32 output channels per workgroup; Cout % 32 == 0) or 1 (Cout % 16 == 0 only: the
// 16-wide 3-D encoder of the texture net).  The filter pack's n-block is 16*NT wide (misc_kernels.hip: pack_wino_kernel).
// TAG does nothing in the kernel: the persistent grid is the same for every layer, so a kernel trace could not tell the layers
// apart; the launcher picks TAG by layer class (0: >= 1024 input channels -- res2; 1: 2-D, fewer -- res3 and the rest;
// 2: 3x3x3 -- the 3-D encoder), which gives each class its own kernel name in rocprofv3's per-kernel statistics.
template <int PROBE, int NT, int MODE, int TAG>
__global__ __launch_bounds__(512, 1)
void conv_wino_kernel(const WinoArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [raw 0][raw 1][U 0][U 1]
    typedef __attribute__((address_space(3))) void lds_void;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    const int T = a.mblocks * a.nblocks, G = gridDim.x;

    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ursrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u_bytes, 0x00020000);

    // fragment-read addresses (bytes) in stage 0.  Tile (ty, tx) = (wave, l16); pixel (2ty+ai, 2tx+bi) sits at q*64 with
    // q = (2ty+ai)*34 + 2tx+bi, physical chunk kq ^ ((tx + (bi>>1)) & 3): two address registers, the rest immediates.
    unsigned raddr0[2];
#pragma unroll
    for (int hj = 0; hj < 2; ++hj)
        raddr0[hj] = (unsigned)((2 * wave * WPW + 2 * l16) * 64 + ((kq ^ ((l16 + hj) & 3)) << 4));
    const unsigned uaddr0 = (unsigned)(2 * WRAW_B + kq * (256 * NT) + l16 * 16);

    constexpr int NXI = wino_nxi(MODE), TP = MODE ? 3 : 4, NSUB = MODE == 1 ? 4 : 1;
    constexpr int UPW = wino_upw(MODE, NT), UPIECES = wino_upieces(MODE, NT), WU_B = wino_ustage(MODE, NT);
    constexpr int NDMA = 5 + UPW;                  // DMA instructions per wave and step: 5 raw-patch + UPW filter pieces
    constexpr unsigned USTEP = 1024u * UPIECES;    // filter bytes per step and n-block
    f32x4 acc[NXI][NT];
#pragma unroll
    for (int t = 0; t < NXI; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][nt][r] = 0.f;

    // DMA number idx_ (0..NDMA-1) of this wave into stage st_: 0..4 = raw-patch pieces (per-lane offsets ro_[]), 5.. = filter
    // pieces (per-lane offset uo_, step offset us_ in an SGPR).  When nothing follows, the offsets are out of range: the
    // hardware then writes zeros into the stage nobody reads -- no branches in the loop.
#define WINO_DMA_ONE(st_, idx_)                                                                           \
    {                                                                                                     \
        if ((idx_) < 5) {                                                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)(smem + (st_) * WRAW_B + (wave + 8 * (idx_)) * 1024), \
                                                     16, ro_[(idx_) < 5 ? (idx_) : 0], 0, 0, 0);          \
        } else {                                                                                          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lds_void*)(smem + 2 * WRAW_B + (st_) * WU_B + (wave + 8 * ((idx_) - 5)) * 1024), \
                                                     16, (UPIECES % 8 != 0 && wave + 8 * ((idx_) - 5) >= UPIECES) ? WOOB : uo_, \
                                                     us_ + (unsigned)((idx_) - 5) * 8192u, 0, 0);          \
        }                                                                                                 \
    }
    // input transform of one xi row i: t = (B^T d)[i][*], v = t B.  MODE 0: B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]];
    // MODE 1 (F(2,2)): B^T = [[1,-1,0],[0,1,0],[0,1,-1]]
#define WINO_ROW(i)                                                                                       \
            f32x4 t_[TP], v_[TP];                                                                         \
            if (PROBE & 1) {                                                                              \
                _Pragma("unroll") for (int bi = 0; bi < TP; ++bi) v_[bi] = d_[i][bi];                     \
            } else if (MODE != 0) {                                                                       \
                _Pragma("unroll") for (int bi = 0; bi < 3; ++bi)                                          \
                    t_[bi] = i == 0 ? pk_sub(d_[0][bi], d_[1][bi]) : i == 1 ? d_[1][bi] : pk_sub(d_[1][bi], d_[2][bi]); \
                v_[0] = pk_sub(t_[0], t_[1]); v_[1] = t_[1]; v_[2] = pk_sub(t_[1], t_[2]);                \
                asm volatile("s_nop 1" : "+v"(v_[0]), "+v"(v_[1]), "+v"(v_[2]));                          \
            } else if (!(PROBE & 4)) {                                                                    \
                _Pragma("unroll") for (int bi = 0; bi < 4; ++bi)                                          \
                    t_[bi] = i == 0 ? pk_sub(d_[0][bi], d_[2][bi]) : i == 1 ? pk_add(d_[1][bi], d_[2][bi]) \
                           : i == 2 ? pk_sub(d_[2][bi], d_[1][bi]) : pk_sub(d_[1][bi], d_[TP - 1][bi]);   \
                v_[0] = pk_sub(t_[0], t_[2]); v_[1] = pk_add(t_[1], t_[2]); v_[2] = pk_sub(t_[2], t_[1]); \
                v_[TP - 1] = pk_sub(t_[1], t_[TP - 1]);                                                   \
                asm volatile("s_nop 1" : "+v"(v_[0]), "+v"(v_[1]), "+v"(v_[2]), "+v"(v_[TP - 1]));        \
            } else {                                                                                      \
                _Pragma("unroll") for (int bi = 0; bi < 4; ++bi)                                          \
                    t_[bi] = i == 0 ? d_[0][bi] - d_[2][bi] : i == 1 ? d_[1][bi] + d_[2][bi]              \
                           : i == 2 ? d_[2][bi] - d_[1][bi] : d_[1][bi] - d_[TP - 1][bi];                 \
                v_[0] = t_[0] - t_[2]; v_[1] = t_[1] + t_[2]; v_[2] = t_[2] - t_[1]; v_[TP - 1] = t_[1] - t_[TP - 1]; \
            }

    WinoBlock cur, nxt;
    int id = blockIdx.x;
    if (id >= T) return;
    wino_block<NT, MODE>(a, id, wave, lane, cur);
    int stage = 0;
    {   // the first step of the first item
        unsigned ro_[5];
#pragma unroll
        for (int i = 0; i < 5; ++i)
            ro_[i] = (MODE != 1 || ((cur.vmask >> i) & 1u)) ? cur.roff[i] + (unsigned)(cur.s_begin / NSUB) * 64u : WOOB;
        const unsigned uo_ = cur.uoff, us_ = (unsigned)cur.s_begin * USTEP;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) WINO_DMA_ONE(0, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (;;) {
        const bool has_next = id + G < T;
        if (has_next) wino_block<NT, MODE>(a, id + G, wave, lane, nxt);
        for (int s = cur.s_begin; s < cur.s_end; ++s) {
            // what the other stage receives during this step: the item's next step, or the next item's first, or nothing
            const bool last = s + 1 == cur.s_end;
            const bool fetch = !last || has_next;
            unsigned ro_[5];
            const int sn = s + 1;                 // the item's next step: sub-filter sn % NSUB of channel step sn / NSUB
            // MODE 1: sub-filter sn & 3 = (a, b) reads the patch shifted by (2a, 2b) pixels
            const unsigned sd_ = MODE != 1 ? 0u : (unsigned)((((sn >> 1) & 1) * 2 * a.W + (sn & 1) * 2) * a.Cin * 4);
            const unsigned vm_ = MODE != 1 ? ~0u : last ? nxt.vmask : cur.vmask >> ((sn & 3) * 5);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const unsigned o_ = last ? nxt.roff[i] + (unsigned)(nxt.s_begin / NSUB) * 64u : cur.roff[i] + sd_ + (unsigned)(sn / NSUB) * 64u;
                ro_[i] = (!fetch || !((vm_ >> i) & 1u)) ? WOOB : o_;
            }
            const unsigned uo_ = !fetch ? WOOB : last ? nxt.uoff : cur.uoff;
            const unsigned us_ = !fetch ? 0u : last ? (unsigned)nxt.s_begin * USTEP : (unsigned)sn * USTEP;
            const int st1 = stage ^ 1;
            if ((PROBE & 10) == 8) {
#pragma unroll
                for (int i = 0; i < NDMA; ++i) WINO_DMA_ONE(st1, i);
            } else if (!(PROBE & 10)) {
#pragma unroll
                for (int i = NXI; i < NDMA; ++i) WINO_DMA_ONE(st1, i);      // more DMAs than xi groups to hide them behind
            }
            // one 16-channel step on `stage`; the nine DMAs are issued one at a time behind the MFMA groups of xi 0..8
            // (all at the top of the step: 7.64 ms instead of 7.10 on res2 -- they stall the step's head)
            const char* rb0_ = smem + raddr0[0] + stage * WRAW_B;
            const char* rb1_ = smem + raddr0[1] + stage * WRAW_B;
            unsigned ua_ = uaddr0 + (unsigned)stage * WU_B;
            asm volatile("" : "+v"(ua_));     // opaque: ONE base register + 16-bit immediates for the 32 filter-fragment reads
            const char* ub_ = smem + ua_;
            f32x4 d_[TP][TP];
#pragma unroll
            for (int ai = 0; ai < TP; ++ai)
#pragma unroll
                for (int bi = 0; bi < TP; ++bi)
                    d_[ai][bi] = *reinterpret_cast<const f32x4*>(((bi >> 1) ? rb1_ : rb0_) + (ai * WPW + bi) * 64);
#pragma unroll
            for (int i = 0; i < TP; ++i) {
                WINO_ROW(i)
#pragma unroll
                for (int jj = 0; jj < TP; ++jj) {
                    f32x4 b_[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        b_[nt] = *reinterpret_cast<const f32x4*>(ub_ + (i * TP + jj) * (1024 * NT) + nt * 256);
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_) {
                        // A operand = filter fragment, B operand = transformed-input fragment: the accumulator then holds
                        // FOUR CONSECUTIVE CHANNELS of one tile per lane (rows = channels), which the epilogue stores as 16 B
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[i * TP + jj][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b_[nt][s_], v_[jj][s_], acc[i * TP + jj][nt], 0, 0, 0);
                    }
                    if (!(PROBE & 10) && i * TP + jj < NDMA) WINO_DMA_ONE(st1, i * TP + jj);
                }
            }
            if (!last) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(PROBE & 128)) __syncthreads();              // measurement: 128 = no barrier between the steps (wrong results)
                stage = st1;
            }
        }
        if (PROBE & 64) {                                      // measurement: no epilogue at all (wrong results)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            f32x4 keep_ = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i_ = 0; i_ < NXI; ++i_)
#pragma unroll
                for (int nt_ = 0; nt_ < NT; ++nt_) { keep_ += acc[i_][nt_]; acc[i_][nt_] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            if (a.B < 0) *reinterpret_cast<f32x4*>(a.y) = keep_;      // never taken: keeps the MFMAs alive
        } else {
        // epilogue of the finished item (its successor's first step is in flight).  C/D layout of the 16x16 MFMA with the
        // filter as A: row = 4*(lane>>4) + r = channel within the 16-wide n-tile, col = lane&15 = the tile's tx: a lane
        // holds channels 4kq..4kq+3 of tile (ty, tx) = (wave, l16) -> 16-B loads and stores, 128 contiguous bytes per pixel
        // and workgroup.  Y = A^T M A with A^T = [[1,1,1,0],[0,1,-1,-1]], M[i][j] = acc[4i+j].
        // vmcnt retires in issue order: a wait for ANY load also waits for every store issued before it.  So all loads (bias,
        // alpha, residual) come first, then ONE wait (which also publishes the next item's first stage, whose DMAs were issued
        // during the last step), then the stores, which drain under the next item's first step.  (Measured: no change against
        // the interleaved load/store order on the 3-D layers -- 0.90 ms either way; kept because it is the order that cannot
        // serialise.)
        const int ty = wave, tx = l16;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        constexpr int NG = NT > 2 ? 2 : NT;                // n-tiles per pass (NT = 4: two passes, registers)
        size_t oo[4];
        bool inb[4];
#pragma unroll
        for (int p4 = 0; p4 < 4; ++p4) {
            const int oy = cur.by * 16 + 2 * ty + (p4 >> 1), ox = cur.bx * 32 + 2 * tx + (p4 & 1);
            inb[p4] = oy < a.H && ox < a.W;
            if (MODE == 2)          // phase (pa, pb) = cur.dz of the 2H x 2W output image
                oo[p4] = ((size_t)(cur.b * 2 * a.H + 2 * oy + (cur.dz >> 1)) * (2 * a.W) + 2 * ox + (cur.dz & 1)) * a.Cout + cur.nb * (16 * NT) + 4 * kq;
            else
                oo[p4] = (((size_t)(cur.b * a.H + oy) * a.W + ox) * a.D + cur.dz) * a.Cout + cur.nb * (16 * NT) + 4 * kq;
        }
#pragma unroll
        for (int n0 = 0; n0 < NT; n0 += NG) {
            f32x4 bv[NG], av[NG], rv[NG][4], v[NG][4];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int nt = n0 + g;
                const int n = cur.nb * (16 * NT) + nt * 16 + 4 * kq;
                bv[g] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n) : zero4;
                av[g] = a.alpha ? *reinterpret_cast<const f32x4*>(a.alpha + n) : zero4;
#pragma unroll
                for (int p4 = 0; p4 < 4; ++p4)
                    rv[g][p4] = (a.res && inb[p4]) ? *reinterpret_cast<const f32x4*>(a.res + oo[p4] + nt * 16) : zero4;
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int nt = n0 + g;
                f32x4 c_[TP][2];                     // column transform of every xi row: M[i][*] A
#pragma unroll
                for (int i = 0; i < TP; ++i) {
                    if (MODE != 0) {                 // F(2,2): A^T = [[1,1,0],[0,1,-1]]
                        c_[i][0] = acc[i * 3 + 0][nt] + acc[i * 3 + 1][nt];
                        c_[i][1] = acc[i * 3 + 1][nt] - acc[i * 3 + 2][nt];
                    } else {
                        c_[i][0] = (acc[i * TP + 0][nt] + acc[i * TP + 1][nt]) + acc[i * TP + 2][nt];
                        c_[i][1] = (acc[i * TP + 1][nt] - acc[i * TP + 2][nt]) - acc[i * TP + TP - 1][nt];
                    }
                }
#pragma unroll
                for (int i = 0; i < NXI; ++i) acc[i][nt] = zero4;
#pragma unroll
                for (int p4 = 0; p4 < 4; ++p4) {
                    const int dy = p4 >> 1, dx = p4 & 1;
                    v[g][p4] = (MODE != 0 ? (dy == 0 ? c_[0][dx] + c_[1][dx] : c_[1][dx] - c_[2][dx])
                                          : (dy == 0 ? (c_[0][dx] + c_[1][dx]) + c_[2][dx] : (c_[1][dx] - c_[2][dx]) - c_[TP - 1][dx])) + bv[g];
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // bias / alpha / residual are in, and so is the next item's first stage
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int p4 = 0; p4 < 4; ++p4) {
                    if (!inb[p4]) continue;
                    const int nt = n0 + g;
                    f32x4 o = v[g][p4];
                    if (a.z) *reinterpret_cast<f32x4*>(a.z + oo[p4] + nt * 16) = o;
                    if (a.act & RN_ACT_PRELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f) + av[g][e] * fminf(o[e], 0.f);
                    }
                    if (a.act & RN_ACT_ELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : expf(o[e]) - 1.f;
                    }
                    if (a.res) o += rv[g][p4];
                    if (a.act & RN_ACT_SIGMOID) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = 1.f / (1.f + __expf(-o[e]));
                    }
                    *reinterpret_cast<f32x4*>(a.y + oo[p4] + nt * 16) = o;
                }
        }
        }
        __syncthreads();                                       // every wave waited for its own DMAs above; the stores drain on their own
        stage ^= 1;
        if (!has_next) break;
        id += G;
        cur = nxt;
    }
#undef WINO_ROW
#undef WINO_DMA_ONE
}

bool rn_wino_supported(int Cin, int Cout)
{
    static const bool off = getenv("RN_NO_WINOGRAD") != nullptr;
    return !off && Cin % 16 == 0 && Cout % 16 == 0;
}

bool rn_wino3d_supported(int Cin, int Cout)
{
    static const bool off = getenv("RN_NO_WINOGRAD3D") != nullptr;
    return !off && rn_wino_supported(Cin, Cout);
}

bool rn_wino4_supported(int Cin, int Cout)
{
    static const bool off = getenv("RN_NO_WINOGRAD4") != nullptr;
    return !off && rn_wino_supported(Cin, Cout);
}

// 16-channel n-tiles per wave; rn_pack_weights follows the same rule (misc_kernels.hip)
int rn_wino_ntiles(int mode, int Cout) { return mode ? (Cout % 64 == 0 ? 4 : Cout % 32 == 0 ? 2 : 1) : (Cout % 32 == 0 ? 2 : 1); }

template <int PROBE, int NT, int MODE, int TAG>
static int wino_launch_tag(const WinoArgs& a, unsigned grid, hipStream_t st)
{
    const size_t lds = wino_lds_bytes(MODE, NT);
    auto kern = conv_wino_kernel<PROBE, NT, MODE, TAG>;
    { const int rc_ = rn_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)160 * 1024); if (rc_ != RN_OK) return rc_; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
    return rn_check_launch("conv_wino");
}

template <int PROBE, int NT, int MODE>
static int wino_launch(const WinoArgs& a, unsigned grid, hipStream_t st)
{
    if (PROBE == 0 && NT == 2 && MODE == 0) {       // the product kernel of the trunk: one name per layer class
        if (a.KD == 3) return wino_launch_tag<0, 2, 0, 2>(a, grid, st);
        if (a.Cin >= 1024) return wino_launch_tag<0, 2, 0, 0>(a, grid, st);
        return wino_launch_tag<0, 2, 0, 1>(a, grid, st);
    }
    return wino_launch_tag<PROBE, NT, MODE, 0>(a, grid, st);
}

// x [B,H,W,(D,)Cin] -> y [B,H,W,(D,)Cout], stride 1.
//   mode 0: 3x3(x3) SAME conv, u from rn_pack_weights(RN_PACK_CONV_WINO | RN_PACK_CONVT_S1_WINO) with the matching ndim;
//           D = 1, KD = 1: 2-D.  KD = 3: 3-D (D >= 1).
//   mode 1: 4x4 2-D conv with pad_lo = pad (1: SAME conv; 2: the flipped conv of a stride-1 transposed conv), u from
//           rn_pack_weights(RN_PACK_CONV_WINO4 | RN_PACK_CONVT_S1_WINO4).
//   mode 2: 4x4 stride-2 SAME transposed conv, x [B,H,W,Cin] -> y [B,2H,2W,Cout], u from rn_pack_weights(RN_PACK_CONVT_S2_WINO).
int rn_launch_conv_wino(const float* x, const float* u, const float* bias, const float* alpha, const float* residual,
                        float* y, float* preact, int B, int H, int W, int D, int KD, int Cin, int Cout, int act,
                        int mode, int pad, hipStream_t st)
{
    if (Cin % 16 != 0 || Cout % 16 != 0)
        return rn_set_error(RN_E_UNSUPPORTED, "conv_wino: Cin=%d Cout=%d (both must be multiples of 16)", Cin, Cout);
    if (D < 1 || (KD != 1 && KD != 3) || (KD == 1 && D != 1) || (mode && KD != 1) || mode < 0 || mode > 2)
        return rn_set_error(RN_E_INVALID, "conv_wino: D=%d KD=%d mode=%d", D, KD, mode);
    const int Dout = mode == 2 ? 4 : D;                 // mode 2: the four output phases of a stride-2 transposed conv
    const long long per_item = (long long)H * W * D * Cin * 4;
    if (per_item >= 0x80000000LL)
        return rn_set_error(RN_E_UNSUPPORTED, "conv_wino: one batch item of %lld bytes exceeds the 2 GiB buffer window", per_item);
    if (per_item * B >= 0x80000000LL) {
        // 32-bit byte offsets with the upper half reserved for the hardware zero fill: batch chunks that fit the window
        const int chunk = (int)(0x7fffffffLL / per_item);
        const size_t ostep = (size_t)H * W * Dout * Cout;
        for (int b0 = 0; b0 < B; b0 += chunk) {
            const int nbi = B - b0 < chunk ? B - b0 : chunk;
            const int rc = rn_launch_conv_wino(x + (size_t)b0 * (per_item / 4), u, bias, alpha,
                                               residual ? residual + b0 * ostep : nullptr, y + b0 * ostep,
                                               preact ? preact + b0 * ostep : nullptr, nbi, H, W, D, KD, Cin, Cout, act, mode, pad, st);
            if (rc != RN_OK) return rc;
        }
        return RN_OK;
    }
    if ((((uintptr_t)x | (uintptr_t)u | (uintptr_t)bias | (uintptr_t)alpha | (uintptr_t)residual | (uintptr_t)y | (uintptr_t)preact) & 15) != 0)
        return rn_set_error(RN_E_INVALID, "conv_wino: every pointer must be 16-byte aligned (16-B loads and stores)");
    WinoArgs a;
    a.x = x; a.u = u; a.bias = bias; a.alpha = alpha; a.res = residual; a.y = y; a.z = preact;
    a.x_bytes = (unsigned)(per_item * B);
    const long long ub = (mode ? 36LL : 16LL * KD) * Cin * Cout * 4;
    if (ub >= 0x80000000LL) return rn_set_error(RN_E_UNSUPPORTED, "conv_wino: transformed filter of %lld bytes exceeds 2 GiB", ub);
    a.u_bytes = (unsigned)ub;
    a.B = B; a.H = H; a.W = W; a.D = Dout; a.KD = KD; a.Cin = Cin; a.Cout = Cout;
    a.bh = (H + 15) / 16; a.bw = (W + 31) / 32;
    const long long mbl = (long long)B * Dout * a.bh * a.bw;
    const int NTv = rn_wino_ntiles(mode, Cout);
    a.nblocks = Cout / (16 * NTv);
    if (mbl * a.nblocks > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "conv_wino: grid too large");
    a.mblocks = (int)mbl;
    a.spt = Cin / 16;
    a.pad = pad;
    a.act = act;
    static const int probe = getenv("RN_WINO_PROBE") ? atoi(getenv("RN_WINO_PROBE")) : 0;
    // persistent grid: one workgroup per CU (144-160 KiB of LDS each), every one walking its share of the items
    static int ncu[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!ncu[dev]) {
        hipDeviceProp_t pr;
        ncu[dev] = hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }
    const long long total = (long long)a.mblocks * a.nblocks;
    static const int grid_env = getenv("RN_WINO_GRID") ? atoi(getenv("RN_WINO_GRID")) : 0;     // measurement: 0 = one per CU
    const long long want = grid_env > 0 ? grid_env : ncu[dev];
    const unsigned grid = (unsigned)(total < want ? total : want);
    if (mode == 1)
        return NTv == 4 ? wino_launch<0, 4, 1>(a, grid, st) : NTv == 2 ? wino_launch<0, 2, 1>(a, grid, st) : wino_launch<0, 1, 1>(a, grid, st);
    if (mode == 2)
        return NTv == 4 ? wino_launch<0, 4, 2>(a, grid, st) : NTv == 2 ? wino_launch<0, 2, 2>(a, grid, st) : wino_launch<0, 1, 2>(a, grid, st);
    if (NTv == 1) return wino_launch<0, 1, 0>(a, grid, st);
    switch (probe) {
        case 1: return wino_launch<1, 2, 0>(a, grid, st);
        case 2: return wino_launch<2, 2, 0>(a, grid, st);
        case 3: return wino_launch<3, 2, 0>(a, grid, st);
        case 4: return wino_launch<4, 2, 0>(a, grid, st);
        case 8: return wino_launch<8, 2, 0>(a, grid, st);
        case 64: return wino_launch<64, 2, 0>(a, grid, st);     // no epilogue
        case 67: return wino_launch<67, 2, 0>(a, grid, st);     // no transform, no DMA in the loop, no epilogue
        case 195: return wino_launch<195, 2, 0>(a, grid, st);   // ... and no per-step barrier: the MFMA loop + fragment reads alone
        default: return wino_launch<0, 2, 0>(a, grid, st);
    }
}
