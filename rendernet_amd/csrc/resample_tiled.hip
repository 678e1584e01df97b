// Tiled resampler: the production path of rn_resample_fwd / rn_resample_affine_fwd.
//
// Same arithmetic, bit for bit, as resample.hip (clamp-then-weight trilinear sampling of
// tools/resampling_voxel_grid.py:381-486, one rounding per op, add_n order), but organised around
// what bounds this op on MI355X: it WRITES N^3 floats per item (201 MB at B=24) and that stream should
// run at HBM speed, while ~98 % of the output is exactly zero -- 7/8 of the 128^3 target grid lies
// outside the 64^3 source (s=1), and most of the inside is empty space.  Two levels of culling keep the
// trilinear arithmetic to the ~4 % of samples that can be non-zero, and the rest of the grid is a pure
// 16-B-store stream.  Three launches:
//   1 resample_prepare_kernel (25 MB read): per item the inverted 3x4 matrix, a voxel bitmap (one bit
//     per voxel, 64-bit words along x), a cell bitmap (one bit per 4^3-voxel cell, one 32-bit word per
//     (cz,cy) row) and a flag telling whether the item is an occupancy grid (values in {0,1});
//   2 resample_classify_kernel: one bit per 8^3 output tile -- can it hold a non-zero sample?
//   3 resample_main_kernel: zero-fill workgroups and sampler workgroups interleaved in one launch.
// Measured history (B=24, five fixtures): see DESIGN.md.
#include "rn_common.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>

#pragma clang fp contract(off)

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int VROWS_MAX = 544;         // rows (z,y) of the voxel-bitmap box kept in LDS (<= 512 used: two per thread;
                                       // 8 x 544 words also hold the padded output staging of a sub-column)
constexpr int MAX_KT = 32;             // tiles per axis (N <= 256)
constexpr int MAT_STRIDE = 12;         // per item: the inverse 3x4 (floats)

__device__ __forceinline__ void pose_to_affine_t(const float* pose, int S, int N, float* m)
{
    // closed form of inverse(T_new_inv * Sc * (Rot_Z*Rot_Y) * T) rows 0:3 (tools/resampling_voxel_grid.py:526-602)
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

// ------------------------------------------------------------------------------------------------
// 1. prepare.  One thread per (z,y) source row; 16 consecutive lanes = the 4x4 rows of one cell row
//    (cz,cy), so the cell bitmap word is an OR over xor-shuffles.  grid = (S*S/256, B).
//    ws_vbit [B][S][S][VW] u32 (VW = max(1, S/32)): bit x of row (z,y) = voxel (z,y,x) has a non-zero channel
//    ws_occ  [B][NC][NC]   u32: bit cx of row (cz,cy) = cell has a non-zero voxel
// ------------------------------------------------------------------------------------------------
template <int CT, bool FROM_POSE>
__global__ __launch_bounds__(256)
void resample_prepare_kernel(const float* __restrict__ vox, const float* __restrict__ mat_or_pose,
                             float* __restrict__ ws_mat, unsigned* __restrict__ ws_occ, unsigned* __restrict__ ws_vbit,
                             unsigned* __restrict__ ws_nb, int S, int N, int NC)
{
    const int b = blockIdx.y;
    // the LAST block of each item only builds the matrix: the double-precision trigonometry is a ~4 us dependent
    // chain on one lane, which used to sit behind the loads of block 0 on the kernel's critical path
    const int nblk = gridDim.x - 1;
    if ((int)blockIdx.x == nblk) {
        if (threadIdx.x == 0) {
            float* m = ws_mat + MAT_STRIDE * b;
            if (FROM_POSE) pose_to_affine_t(mat_or_pose + 3 * b, S, N, m);
            else for (int q = 0; q < 12; ++q) m[q] = mat_or_pose[12 * b + q];
        }
        return;
    }
    int nonbin = (CT == 1) ? 0 : 1;            // some non-zero voxel differs from 1.0f (occupancy grids are {0,1})
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int VW = S >= 32 ? S >> 5 : 1;
    // e -> (cz, cy, zl, yl): the low 4 bits pick the row inside the cell row
    const int yl = e & 3, zl = (e >> 2) & 3, cy = (e >> 4) % NC, cz = (e >> 4) / NC;
    const bool live = cz < NC;
    const int z = cz * 4 + zl, y = cy * 4 + yl;
    unsigned cellbits = 0;
    if (live) {
        const float4* p = reinterpret_cast<const float4*>(vox + (((size_t)b * S + z) * S + y) * S * CT);
        unsigned* vrow = ws_vbit + (((size_t)b * S + z) * S + y) * VW;
        for (int w = 0; w < VW; ++w) {
            unsigned bits = 0;
            const int nx = S < 32 ? S : 32;
            if (CT == 1) {
                for (int q = 0; q < nx / 4; ++q) {
                    const float4 v = p[w * 8 + q];
                    nonbin |= ((v.x != 0.f) & (v.x != 1.f)) | ((v.y != 0.f) & (v.y != 1.f)) |
                              ((v.z != 0.f) & (v.z != 1.f)) | ((v.w != 0.f) & (v.w != 1.f));
                    bits |= ((v.x != 0.f) ? 1u : 0u) << (4 * q) | ((v.y != 0.f) ? 2u : 0u) << (4 * q) |
                            ((v.z != 0.f) ? 4u : 0u) << (4 * q) | ((v.w != 0.f) ? 8u : 0u) << (4 * q);
                }
            } else {
                for (int q = 0; q < nx; ++q) {
                    const float4 v = p[w * 32 + q];
                    bits |= (((v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f)) ? 1u : 0u) << q;
                }
            }
            vrow[w] = bits;
            // nibble-any: bit cx (of this 32-voxel word: cx = 8*w + n)
            for (int n = 0; n < nx / 4; ++n)
                cellbits |= (((bits >> (4 * n)) & 15u) ? 1u : 0u) << (8 * w + n);
        }
    }
    cellbits |= __shfl_xor(cellbits, 1);
    cellbits |= __shfl_xor(cellbits, 2);
    cellbits |= __shfl_xor(cellbits, 4);
    cellbits |= __shfl_xor(cellbits, 8);
    if (live && (threadIdx.x & 15) == 0) ws_occ[((size_t)b * NC + cz) * NC + cy] = cellbits;
    // every (block, item) slot is rewritten by every call: no zeroing, no atomics
    nonbin = __syncthreads_or(nonbin);
    if (threadIdx.x == 0) ws_nb[(size_t)b * nblk + blockIdx.x] = (unsigned)nonbin;
}

__device__ __forceinline__ float coord_t(float m0, float m1, float m2, float m3, float x, float y, float z)
{
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m0, x), __fmul_rn(m1, y)), __fmul_rn(m2, z)), m3);
}

// clamped tap indices of one sample (tools/resampling_voxel_grid.py:410-422)
struct Taps { int x0, x1, y0, y1, z0, z1; };
__device__ __forceinline__ Taps sample_taps(int S, float x, float y, float z)
{
    const int mx = S - 1;
    Taps t;
    t.x0 = (int)floorf(x); t.y0 = (int)floorf(y); t.z0 = (int)floorf(z);
    t.x1 = t.x0 + 1; t.y1 = t.y0 + 1; t.z1 = t.z0 + 1;
    t.x0 = min(max(t.x0, 0), mx); t.x1 = min(max(t.x1, 0), mx);
    t.y0 = min(max(t.y0, 0), mx); t.y1 = min(max(t.y1, 0), mx);
    t.z0 = min(max(t.z0, 0), mx); t.z1 = min(max(t.z1, 0), mx);
    return t;
}

// weights from the clamped indices and the add_n of the eight products (:465-485).  tv[n][c]: the eight taps in
// add_n order a..h = (z0,y0,x0) (z0,y1,x0) (z0,y0,x1) (z0,y1,x1) (z1,y0,x0) (z1,y1,x0) (z1,y0,x1) (z1,y1,x1)
template <int CT>
__device__ __forceinline__ void sample_eval(const Taps& t, float x, float y, float z, const float (&tv)[8][CT],
                                            float* __restrict__ o)
{
    const float ax = __fsub_rn((float)t.x1, x), bx = __fsub_rn(x, (float)t.x0);
    const float ay = __fsub_rn((float)t.y1, y), by = __fsub_rn(y, (float)t.y0);
    const float az = __fsub_rn((float)t.z1, z), bz = __fsub_rn(z, (float)t.z0);
    const float wa = __fmul_rn(__fmul_rn(ax, ay), az);
    const float wb = __fmul_rn(__fmul_rn(ax, by), az);
    const float wc = __fmul_rn(__fmul_rn(bx, ay), az);
    const float wd = __fmul_rn(__fmul_rn(bx, by), az);
    const float we = __fmul_rn(__fmul_rn(ax, ay), bz);
    const float wf = __fmul_rn(__fmul_rn(ax, by), bz);
    const float wg = __fmul_rn(__fmul_rn(bx, ay), bz);
    const float wh = __fmul_rn(__fmul_rn(bx, by), bz);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        float v = __fmul_rn(wa, tv[0][c]);
        v = __fadd_rn(v, __fmul_rn(wb, tv[1][c]));
        v = __fadd_rn(v, __fmul_rn(wc, tv[2][c]));
        v = __fadd_rn(v, __fmul_rn(wd, tv[3][c]));
        v = __fadd_rn(v, __fmul_rn(we, tv[4][c]));
        v = __fadd_rn(v, __fmul_rn(wf, tv[5][c]));
        v = __fadd_rn(v, __fmul_rn(wg, tv[6][c]));
        v = __fadd_rn(v, __fmul_rn(wh, tv[7][c]));
        o[c] = v;
    }
}

struct TiledArgs {
    const float* vox;        // [B,S,S,S,C]
    const float* ws_mat;     // [B,MAT_STRIDE]
    const unsigned* ws_occ;  // [B,NC,NC]
    const unsigned* ws_vbit; // [B,S,S,VW]
    unsigned* ws_nb;         // [B, nprep] per prepare-block flag: a non-zero voxel other than 1.0f was seen; then [B]: their OR (classify)
    int nprep;
    unsigned* ws_colmask;    // [B, ph/8, pw/8]  bit kt = tile (column, kt) is a candidate
    uint2* ws_box;           // [B, ph/8, pw/8, 32]  box of tile kt: {bx0|bx1<<8|by0<<16|by1<<24, bz0|bz1<<8}
    float* out;
    int B, S, N, NC;
    int h0, w0, ph, pw, image_layout;
    int debug;               // RN_RS_DEBUG, exact results either way: 3 = no per-sample bit test, 5 = no occupancy-grid fast path,
                             // 6 = no voxel-level tile cull in the sampler
    int ratio;               // sampler workgroups per fill workgroup in the interleaved launch order
    int nfill, nsub;         // fill rows (B*ph) and sampler sub-columns (B * ph/8 * pw/8 * ceil(N/32)) of the main launch
};

// bounding box (source voxel indices, inclusive) of the pre-image of the 8^3 tile at (i0, j0, k0): the 8 corners go
// through the sampler's own coordinate arithmetic on lanes (lane & 7), min/max over xor-shuffles of 1,2,4, so every
// lane of each group of 8 returns the same box
__device__ __forceinline__ void tile_bbox(const float* m, int S, int N, int image_layout, int i0, int j0, int k0,
                                          int lane, int* b0, int* b1)
{
    const int ci = lane & 1, cj = (lane >> 1) & 1, ck = (lane >> 2) & 1;
    const int i = i0 + 7 * ci, j = j0 + 7 * cj, k = k0 + 7 * ck;
    const float gx = (float)k;
    const float gy = image_layout ? (float)(N - 1 - i) : (float)j;
    const float gz = image_layout ? (float)j : (float)i;
    float lo[3], hi[3];
    lo[0] = hi[0] = coord_t(m[0], m[1], m[2], m[3], gx, gy, gz);
    lo[1] = hi[1] = coord_t(m[4], m[5], m[6], m[7], gx, gy, gz);
    lo[2] = hi[2] = coord_t(m[8], m[9], m[10], m[11], gx, gy, gz);
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s = 1; s < 8; s <<= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], s));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], s));
        }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        // every tap index of every sample of the tile is clamp(floor(c)) or clamp(floor(c)+1) with c in [lo,hi] -- exactly,
        // no rounding margin is needed: coord_t is a chain of correctly rounded multiplies and adds, each monotone in
        // its varying operand, so the COMPUTED coordinate is monotone in each of (i, j, k) and its extremes over the
        // tile are attained at the corners, which are themselves samples evaluated with the same arithmetic.
        // (A margin of one voxel on either side, as first written, doubled the box volume and with it the candidates.)
        const float l = fminf(fmaxf(floorf(lo[d]), 0.f), (float)(S - 1));
        const float h = fminf(fmaxf(floorf(hi[d]) + 1.f, 0.f), (float)(S - 1));
        b0[d] = (int)l; b1[d] = (int)h;
    }
}

// ------------------------------------------------------------------------------------------------
// 2. classify.  One wave per (b, ti, tj) column; lane = (g = tile of the pass, c = corner / cell row), eight
// tiles per pass.  A tile whose box holds no occupied cell is exactly zero everywhere.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void resample_classify_kernel(const TiledArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nti = a.ph >> 3, ntj = a.pw >> 3, nkt = a.N >> 3;
    const long long col = (long long)blockIdx.x * 4 + wave;
    if (col >= (long long)a.B * nti * ntj) return;
    const int tj = (int)(col % ntj), ti = (int)((col / ntj) % nti), b = (int)(col / ((long long)ntj * nti));
    if (ti == 0 && tj == 0) {
        // the item's first column also folds the prepare blocks' "non-binary voxel seen" flags into one word, so that
        // a sampler workgroup reads one scalar instead of walking nprep of them
        unsigned nb = 0u;
        for (int q = lane; q < a.nprep; q += 64) nb |= a.ws_nb[(size_t)b * a.nprep + q];
        const unsigned long long any = __ballot(nb != 0u);
        if (lane == 0) a.ws_nb[(size_t)a.B * a.nprep + b] = any ? 1u : 0u;
    }
    float m[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) m[q] = a.ws_mat[MAT_STRIDE * b + q];
    const unsigned* occ = a.ws_occ + (size_t)b * a.NC * a.NC;
    const int i0 = a.h0 + ti * 8, j0 = a.w0 + tj * 8;
    const int g = lane >> 3, c = lane & 7;
    unsigned mask = 0;
    const int npass = (nkt + 7) >> 3;
    for (int pass = 0; pass < npass; ++pass) {
        const int kt = pass * 8 + g;
        int b0[3], b1[3];
        tile_bbox(m, a.S, a.N, a.image_layout, i0, j0, kt * 8, lane, b0, b1);
        const int cx0 = b0[0] >> 2, cx1 = b1[0] >> 2;
        const int cy0 = b0[1] >> 2, ncy = (b1[1] >> 2) - cy0 + 1;
        const int cz0 = b0[2] >> 2, ncz = (b1[2] >> 2) - cz0 + 1;
        // lane c owns cell row cy0 + c and walks the cz of the box (independent loads)
        bool hit = false;
        if (kt < nkt && c < ncy) {
            const unsigned mhi = cx1 >= 31 ? 0xffffffffu : ((1u << (cx1 + 1)) - 1u);
            const unsigned mlo = (1u << cx0) - 1u;
            unsigned acc = 0;
            for (int zz = 0; zz < ncz; ++zz) acc |= occ[(cz0 + zz) * a.NC + cy0 + c];
            hit = (acc & mhi & ~mlo) != 0u;
        }
        if (kt < nkt && ncy > 8) hit = true;             // box wider than 8 cell rows: be conservative
        const unsigned long long bal = __ballot(hit);
        unsigned bits = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) bits |= (((bal >> (8 * q)) & 0xffull) ? 1u : 0u) << q;
        mask |= bits << (pass * 8);
        if (c == 0 && kt < nkt)
            a.ws_box[(size_t)col * MAX_KT + kt] =
                make_uint2((unsigned)b0[0] | ((unsigned)b1[0] << 8) | ((unsigned)b0[1] << 16) | ((unsigned)b1[1] << 24),
                           (unsigned)b0[2] | ((unsigned)b1[2] << 8));
    }
    if (lane == 0) a.ws_colmask[col] = mask;
}

// ------------------------------------------------------------------------------------------------
// 3. main.  Two kinds of workgroups interleaved in one launch (one fill workgroup, then `ratio` samplers, ...):
//   fill: one (b,i) row of the output (pw*N floats, 64 KiB) zero-filled with 16-B stores, skipping the 128-B
//     line segments of sub-columns that hold a candidate tile.  One tiny load, then stores only.
//   sampler: one (b, ti, tj, kq) sub-column = four 8^3 tiles; exits at once if none is a candidate.  Strictly
//     LOADS -> COMPUTE -> STORES: vector memory operations complete in issue order (one vmcnt for loads and
//     stores), so a load issued behind stores into an HBM-saturated write stream waits for them -- every earlier
//     form of this kernel that reloaded after storing (persistent loops over tiles, fill chunks between tiles)
//     spent its time exactly there, not in arithmetic or gathers.
//       a. wave w: box of tile w (as the classifier computed it);
//       b. the voxel-bitmap rows of the candidate tiles' boxes -> LDS (<= 4 KiB each);
//       c. every sample tests its own eight taps against those rows (a sample whose taps are all zero IS zero);
//          the ~20 % that pass run the trilinear arithmetic -- on occupancy grids ({0,1} values, flagged by
//          the prepare pass) with taps read from the bitmap itself, otherwise gathered from L1/L2;
//       d. 16-B stores (results of four depth-adjacent lanes are combined first).
// ------------------------------------------------------------------------------------------------
// f(integral_constant<int,0>{}), ..., f(integral_constant<int,TPW-1>{}): compile-time tile indices keep the per-tile
// arrays in registers
template <class F, int... I>
__device__ __forceinline__ void for_each_tile(F& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}

// TPW = depth-adjacent tiles per sampler workgroup (8: measured 4 -> 65 us, 8 -> 57-59 us, 16 -> 83 us for the main
// launch at B=24: fatter workgroups amortise the load -> barrier -> store chain, too fat ones leave too few in flight)
template <int CT, int TPW>
__global__ __launch_bounds__(256)
void resample_main_kernel(const TiledArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned vrows[TPW][VROWS_MAX];   // bit x of row r = voxel bx0 + x of the box row (32-bit window)
    __shared__ unsigned rowmask[MAX_KT];
    __shared__ unsigned tflag;                 // sampler: bit tt = the bitmap rows of tile tt's box hold an occupied voxel
    const int tid = threadIdx.x;
    const int N = a.N, S = a.S;
    const int VW = S >= 32 ? S >> 5 : 1;
    const int nti = a.ph >> 3, ntj = a.pw >> 3, nkt = N >> 3, nkq = (nkt + TPW - 1) / TPW;
    // interleaved 1:ratio (measured at 1:8 with 4-tile sub-columns: 65 us; all samplers first 81 us, all fill first 84 us)
    const int grp = blockIdx.x / (a.ratio + 1), rem = blockIdx.x % (a.ratio + 1);

    if (rem == 0) {
        // ---------------- fill ----------------
        if (grp >= a.nfill) return;
        const int il = grp % a.ph, b = grp / a.ph;
        if (tid < ntj) rowmask[tid] = a.ws_colmask[((size_t)b * nti + (il >> 3)) * ntj + tid];
        __syncthreads();
        float4* op = reinterpret_cast<float4*>(a.out + ((size_t)b * a.ph + il) * a.pw * N * CT);
        const int per_line = (CT == 1) ? (N >> 2) : N;        // 16-B units per (i,j) depth line
        const int total = a.pw * per_line;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int u = tid; u < total; u += 256) {
            const int j = u / per_line, w = u - j * per_line;
            const int kt = (CT == 1) ? (w >> 1) : (w >> 3);
            // a sub-column (4 tiles = one 128-B line per (i,j) at C=1) with ANY candidate belongs to its sampler
            // workgroup entirely, so that every line is written once, in full, by one workgroup
            if (!((rowmask[j >> 3] >> (kt / TPW * TPW)) & ((1u << TPW) - 1u))) op[u] = z4;
        }
        return;
    }

    // ---------------- sampler ----------------
    int id = grp * a.ratio + rem - 1;
    if (id >= a.nsub) return;
    const int kq = id % nkq; id /= nkq;
    const int tj = id % ntj; id /= ntj;
    const int ti = id % nti; const int b = id / nti;
    const unsigned cmask = (a.ws_colmask[((size_t)b * nti + ti) * ntj + tj] >> (TPW * kq)) & ((1u << TPW) - 1u);   // uniform
    if (cmask == 0u) return;
    if (tid == 0) tflag = 0u;
    const int i0 = a.h0 + ti * 8, j0 = a.w0 + tj * 8;

    const float* mp = a.ws_mat + (size_t)MAT_STRIDE * b;      // uniform address -> scalar loads
    float m[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) m[q] = mp[q];
    const unsigned nonbin = a.ws_nb[(size_t)a.B * a.nprep + b];
    const bool binary = CT == 1 && nonbin == 0u && a.debug != 5;

    // ---- a. the boxes the classifier recorded (uniform addresses -> scalar loads, no barrier) ----
    int tinfo[TPW][8];
    {
        const uint2* bp = a.ws_box + (((size_t)b * nti + ti) * ntj + tj) * MAX_KT + kq * TPW;
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const uint2 bx = bp[tt];
            tinfo[tt][1] = bx.x & 255; tinfo[tt][2] = (bx.x >> 8) & 255; tinfo[tt][3] = (bx.x >> 16) & 255; tinfo[tt][4] = bx.x >> 24;
            tinfo[tt][5] = bx.y & 255; tinfo[tt][6] = (bx.y >> 8) & 255;
        }
    }

    // ---- b. voxel-bitmap rows of the candidate tiles: all loads first, then LDS ----
    // (the per-tile steps are generic lambdas instantiated with a compile-time tile index: with runtime-indexed
    //  loops the compiler kept rv / vt / res in scratch memory -- loads and stores behind our own stores)
    unsigned rv[TPW][2];
    bool vt[TPW];
    auto load_rows = [&](auto TT) {
        constexpr int tt = decltype(TT)::value;
        rv[tt][0] = 0u; rv[tt][1] = 0u;
        vt[tt] = false;
        if ((cmask >> tt) & 1u) {                                         // uniform
            const int bx0 = tinfo[tt][1], bx1 = tinfo[tt][2], by0 = tinfo[tt][3], by1 = tinfo[tt][4];
            const int bz0 = tinfo[tt][5], bz1 = tinfo[tt][6];
            const int ny = by1 - by0 + 1, rows = ny * (bz1 - bz0 + 1);
            const int xw0 = min(bx0 >> 5, max(VW - 2, 0));
            vt[tt] = rows <= 512 && (bx1 - xw0 * 32) < 64 && (bx1 - bx0) < 32 && a.debug != 3;
            if (vt[tt]) {
                const float rny = 1.0f / (float)ny;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int r = tid + 256 * half;
                    if (r < rows) {
                        // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): every factor here is < 2^12
                        const int z = (int)(((float)r + 0.5f) * rny), y = r - __mul24(z, ny);
                        const unsigned* vr = a.ws_vbit + (size_t)b * S * S * VW +
                                             (unsigned)(__mul24(__mul24(bz0 + z, S) + by0 + y, VW) + xw0);
                        // the 32 bits starting at voxel bx0: the only 64-bit shift of this row (none per sample)
                        const unsigned long long w64 = ((unsigned long long)((xw0 + 1 < VW) ? vr[1] : 0u) << 32) | vr[0];
                        rv[tt][half] = (unsigned)(w64 >> (bx0 - xw0 * 32));
                    }
                }
            }
        }
    };
    for_each_tile(load_rows, std::make_integer_sequence<int, TPW>{});
    __syncthreads();                                                       // tflag = 0 is visible (the loads are in flight)
    // The classifier tested the box against the 4^3-CELL bitmap; the rows just fetched are the box at VOXEL
    // granularity: a box without a single occupied voxel makes the whole tile exactly zero.  On the bench batch this
    // drops 42 % of the cell-level candidates (3455 -> 2001 of 20480 tiles over five frames; 1092 hold a non-zero sample).
    unsigned novt = 0u;
    auto put_rows = [&](auto TT) {
        constexpr int tt = decltype(TT)::value;
        if (vt[tt]) {
            vrows[tt][tid] = rv[tt][0];
            vrows[tt][tid + 256] = rv[tt][1];
            const int w = tinfo[tt][2] - tinfo[tt][1];                    // bx1 - bx0 < 32
            const unsigned xmask = w >= 31 ? 0xffffffffu : ((2u << w) - 1u);
            if ((rv[tt][0] | rv[tt][1]) & xmask) atomicOr(&tflag, 1u << tt);
        } else if ((cmask >> tt) & 1u) {
            novt |= 1u << tt;                                              // no rows staged: cannot be culled here
        }
    };
    for_each_tile(put_rows, std::make_integer_sequence<int, TPW>{});
    __syncthreads();
    const unsigned live = a.debug == 6 ? cmask : (cmask & (__builtin_amdgcn_readfirstlane(tflag) | novt));

    // ---- c. samples ----
    const float* vb = a.vox + (size_t)b * S * S * S * CT;
    float res[TPW][2][CT];
    auto sample_tile = [&](auto TT) {
        constexpr int tt = decltype(TT)::value;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int cc = 0; cc < CT; ++cc) res[tt][q][cc] = 0.f;
        if (!((live >> tt) & 1u)) return;                                  // uniform
        const int bx0 = tinfo[tt][1], by0 = tinfo[tt][3], by1 = tinfo[tt][4], bz0 = tinfo[tt][5];
        const int ny = by1 - by0 + 1;
        const int k0 = (kq * TPW + tt) * 8;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = tid + 256 * q;
            const int kl = p & 7, jl = (p >> 3) & 7, il = p >> 6;
            const int i = i0 + il, j = j0 + jl, k = k0 + kl;
            const float gx = (float)k;
            const float gy = a.image_layout ? (float)(N - 1 - i) : (float)j;
            const float gz = a.image_layout ? (float)j : (float)i;
            const float xs = coord_t(m[0], m[1], m[2], m[3], gx, gy, gz);
            const float ys = coord_t(m[4], m[5], m[6], m[7], gx, gy, gz);
            const float zs = coord_t(m[8], m[9], m[10], m[11], gx, gy, gz);
            const Taps u = sample_taps(S, xs, ys, zs);
            bool hit = true;
            unsigned w00 = 0u, w01 = 0u, w10 = 0u, w11 = 0u;                        // rows (z0,y0) (z0,y1) (z1,y0) (z1,y1)
            const int sx0 = u.x0 - bx0, sx1 = u.x1 - bx0;
            if (vt[tt]) {
                // a sample whose eight taps are all zero is exactly zero
                const int rz0 = __mul24(u.z0 - bz0, ny), rz1 = __mul24(u.z1 - bz0, ny), ry0 = u.y0 - by0, ry1 = u.y1 - by0;
                w00 = vrows[tt][rz0 + ry0]; w01 = vrows[tt][rz0 + ry1];
                w10 = vrows[tt][rz1 + ry0]; w11 = vrows[tt][rz1 + ry1];
                hit = ((w00 | w01 | w10 | w11) & ((1u << sx0) | (1u << sx1))) != 0u;
            }
            if (hit) {
                float tv[8][CT];
                if (binary && vt[tt]) {
                    // occupancy grid: a tap is exactly 0.0f or 1.0f and the bitmap says which (no gather at all)
                    tv[0][0] = (float)((w00 >> sx0) & 1u); tv[1][0] = (float)((w01 >> sx0) & 1u);
                    tv[2][0] = (float)((w00 >> sx1) & 1u); tv[3][0] = (float)((w01 >> sx1) & 1u);
                    tv[4][0] = (float)((w10 >> sx0) & 1u); tv[5][0] = (float)((w11 >> sx0) & 1u);
                    tv[6][0] = (float)((w10 >> sx1) & 1u); tv[7][0] = (float)((w11 >> sx1) & 1u);
#pragma unroll
                    for (int n = 0; n < 8; ++n)
#pragma unroll
                        for (int cc = 1; cc < CT; ++cc) tv[n][cc] = 0.f;   // (CT > 1 never takes this branch)
                } else {
                    const float* r00 = vb + (((size_t)u.z0 * S + u.y0) * S) * CT, * r01 = vb + (((size_t)u.z0 * S + u.y1) * S) * CT;
                    const float* r10 = vb + (((size_t)u.z1 * S + u.y0) * S) * CT, * r11 = vb + (((size_t)u.z1 * S + u.y1) * S) * CT;
#pragma unroll
                    for (int cc = 0; cc < CT; ++cc) {
                        tv[0][cc] = r00[u.x0 * CT + cc]; tv[1][cc] = r01[u.x0 * CT + cc];
                        tv[2][cc] = r00[u.x1 * CT + cc]; tv[3][cc] = r01[u.x1 * CT + cc];
                        tv[4][cc] = r10[u.x0 * CT + cc]; tv[5][cc] = r11[u.x0 * CT + cc];
                        tv[6][cc] = r10[u.x1 * CT + cc]; tv[7][cc] = r11[u.x1 * CT + cc];
                    }
                }
                float r[CT];
                sample_eval<CT>(u, xs, ys, zs, tv, r);
#pragma unroll
                for (int cc = 0; cc < CT; ++cc) res[tt][q][cc] = r[cc];
            }
        }
    };
    for_each_tile(sample_tile, std::make_integer_sequence<int, TPW>{});

    // ---- d. stores (nothing is loaded after this point).  The sub-column is 64 (i,j) lines of TPW*8 consecutive depth
    // samples; results go through LDS so that every wave instruction writes whole contiguous lines (16 lanes x 16 B
    // per line at C=1): written straight from the sampling layout (32-B pieces at a 512-B stride) this kernel's
    // stores ran at 1.5-2.5 TB/s against 7.4 TB/s for the fill workgroups' 1-KiB-contiguous stores. ----
    constexpr int LINE = TPW * 8 * CT;                         // floats per (i,j) line of the sub-column
    constexpr int LSTR = LINE + 4;                             // padded line stride (LDS banks), still 16-B aligned
    float* obuf = reinterpret_cast<float*>(&vrows[0][0]);      // the bitmap rows are dead: reuse their LDS
    static_assert(sizeof(vrows) >= 64 * LSTR * sizeof(float) || CT > 1, "output staging fits in the bitmap rows");
    __shared__ __attribute__((aligned(16))) float obuf4[(CT > 1) ? 64 * LSTR : 4];
    float* ob = (CT > 1) ? obuf4 : obuf;
    __syncthreads();                                           // everyone is done reading vrows
    auto stage_tile = [&](auto TT) {
        constexpr int tt = decltype(TT)::value;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = tid + 256 * q;
            const int kl = p & 7, line = p >> 3;               // line = il*8 + jl
#pragma unroll
            for (int cc = 0; cc < CT; ++cc) ob[line * LSTR + (tt * 8 + kl) * CT + cc] = res[tt][q][cc];
        }
    };
    for_each_tile(stage_tile, std::make_integer_sequence<int, TPW>{});
    __syncthreads();
    const size_t patch_base = (((size_t)b * a.ph + ti * 8) * a.pw + tj * 8) * N;   // in voxels
    const int kvalid = min(TPW, nkt - kq * TPW) * 8 * CT;      // floats of the line that exist (N may end mid sub-column)
    constexpr int UPL = LINE / 4;                              // 16-B units per line
    for (int u = tid; u < 64 * UPL; u += 256) {
        const int line = u / UPL, w = (u - line * UPL) * 4;
        if (w < kvalid) {
            const int il = line >> 3, jl = line & 7;
            float* op = a.out + (patch_base + ((size_t)il * a.pw + jl) * N + (size_t)kq * TPW * 8) * CT + w;
            *reinterpret_cast<float4*>(op) = *reinterpret_cast<const float4*>(ob + line * LSTR + w);
        }
    }
}

// ------------------------------------------------------------------------------------------------
bool rn_resample_tiled_supported(int B, int S, int N, int C, int ph, int pw)
{
    if (C != 1 && C != 4) return false;
    if (S != 16 && S != 32 && S != 64 && S != 128) return false;
    if (N % 8 != 0 || N > 8 * MAX_KT || N < 16 || ph % 8 != 0 || pw % 8 != 0) return false;
    (void)B;
    return true;
}

// [B,12] matrices | [B,NC,NC] cell bitmap | [B,S,S,VW] voxel bitmap | [B,nprep] non-binary flags | [B,32,32] column masks | [B,32,32,32] tile boxes
// (sized for the largest supported grid, N = 256: the entry point does not know N)
static size_t ws_layout(int B, int S, size_t* o_occ, size_t* o_vbit, size_t* o_nb, size_t* o_mask, size_t* o_box)
{
    const int NC = S / 4, VW = S >= 32 ? S / 32 : 1;
    size_t off = (size_t)B * MAT_STRIDE * sizeof(float);
    *o_occ = off;  off += (size_t)B * NC * NC * sizeof(unsigned);
    *o_vbit = off; off += (size_t)B * S * S * VW * sizeof(unsigned);
    *o_nb = off;   off += (size_t)B * (S * S / 256 + 1) * sizeof(unsigned);
    *o_mask = off; off += (size_t)B * MAX_KT * MAX_KT * sizeof(unsigned);
    off = (off + 15) & ~(size_t)15;
    *o_box = off;  off += (size_t)B * MAX_KT * MAX_KT * MAX_KT * sizeof(uint2);
    return off;
}

size_t rn_resample_tiled_workspace(int B, int S)
{
    size_t a, b, c, d, e;
    return ws_layout(B, S, &a, &b, &c, &d, &e) + 128;           // slack: the caller's buffer may be 16-B aligned only
}

int rn_launch_resample_tiled(const float* vox, const float* mat_or_pose, bool from_pose, float* out,
                             int B, int S, int N, int C, int h0, int w0, int ph, int pw, int image_layout,
                             void* workspace, hipStream_t st)
{
    const int NC = S / 4;
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 127) & ~(uintptr_t)127);
    size_t o_occ, o_vbit, o_nb, o_mask, o_box;
    ws_layout(B, S, &o_occ, &o_vbit, &o_nb, &o_mask, &o_box);
    float* ws_mat = reinterpret_cast<float*>(ws);
    unsigned* ws_occ = reinterpret_cast<unsigned*>(ws + o_occ);
    unsigned* ws_vbit = reinterpret_cast<unsigned*>(ws + o_vbit);
    unsigned* ws_nb = reinterpret_cast<unsigned*>(ws + o_nb);
    unsigned* ws_colmask = reinterpret_cast<unsigned*>(ws + o_mask);
    uint2* ws_box = reinterpret_cast<uint2*>(ws + o_box);
    const int nprep = (NC * NC * 16 + 255) / 256;             // prepare blocks (= flag slots) per item
    dim3 pgrid(nprep + 1, B);                                 // + the matrix block
    if (C == 1) {
        if (from_pose) hipLaunchKernelGGL((resample_prepare_kernel<1, true>), pgrid, dim3(256), 0, st, vox, mat_or_pose, ws_mat, ws_occ, ws_vbit, ws_nb, S, N, NC);
        else hipLaunchKernelGGL((resample_prepare_kernel<1, false>), pgrid, dim3(256), 0, st, vox, mat_or_pose, ws_mat, ws_occ, ws_vbit, ws_nb, S, N, NC);
    } else {
        if (from_pose) hipLaunchKernelGGL((resample_prepare_kernel<4, true>), pgrid, dim3(256), 0, st, vox, mat_or_pose, ws_mat, ws_occ, ws_vbit, ws_nb, S, N, NC);
        else hipLaunchKernelGGL((resample_prepare_kernel<4, false>), pgrid, dim3(256), 0, st, vox, mat_or_pose, ws_mat, ws_occ, ws_vbit, ws_nb, S, N, NC);
    }
    int rc = rn_check_launch("resample_prepare");
    if (rc != RN_OK) return rc;
    static const int dbg = getenv("RN_RS_DEBUG") ? atoi(getenv("RN_RS_DEBUG")) : 0;
    const long long ncol = (long long)B * (ph / 8) * (pw / 8);
    const int TPW = 8;
    const int nkq = (N / 8 + TPW - 1) / TPW;
    const long long nfill = (long long)B * ph, nsub = ncol * nkq;
    // sampler ids per fill id in the launch order: twice the natural count (the surplus ids exit at once) -- measured
    // at B=24: 4 -> 63 us, 6 -> 58, 8 -> 59, 12 -> 60, 16 -> 65 (fill workgroups spaced further apart start later,
    // closer together they crowd the samplers out of the CUs)
    long long ratio = 2 * ((nsub + nfill - 1) / nfill);
    static const int ratio_env = getenv("RN_RS_RATIO") ? atoi(getenv("RN_RS_RATIO")) : 0;   // tuning knob
    if (ratio_env > 0) ratio = ratio_env;
    if (ratio < 1) ratio = 1;
    const long long groups = nfill > (nsub + ratio - 1) / ratio ? nfill : (nsub + ratio - 1) / ratio;
    if (groups * (ratio + 1) > 0x7fffffffLL) return rn_set_error(RN_E_INVALID, "resample: grid too large");
    TiledArgs a{vox, ws_mat, ws_occ, ws_vbit, ws_nb, nprep, ws_colmask, ws_box, out, B, S, N, NC, h0, w0, ph, pw,
                image_layout, dbg, (int)ratio, (int)nfill, (int)nsub};
    hipLaunchKernelGGL(resample_classify_kernel, dim3((unsigned)((ncol + 3) / 4)), dim3(256), 0, st, a);
    rc = rn_check_launch("resample_classify");
    if (rc != RN_OK) return rc;
    if (C == 1) hipLaunchKernelGGL((resample_main_kernel<1, 8>), dim3((unsigned)(groups * (ratio + 1))), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((resample_main_kernel<4, 8>), dim3((unsigned)(groups * (ratio + 1))), dim3(256), 0, st, a);
    return rn_check_launch("resample_main");
}
