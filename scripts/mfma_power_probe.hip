// What does the chip sustain on bf16 MFMAs alone, per instruction shape and per operand statistics?  (gfx950; the split GEMM stage sits at a power
// ceiling -- MFMA busy 0.74-0.75 at 1.73-1.78 GHz -- so the question is whether another shape or another accumulator arrangement buys joules.)
// Every workgroup = 8 waves (two per SIMD), one workgroup per CU x 256 CUs, register operands only (no LDS, no memory in the loop):
//   A  v_mfma_f32_32x32x16_bf16, wave tile 4 x 2 MFMA tiles (128 accumulators, the product kernel's arrangement: 6 fragments feed 8 MFMAs)
//   B  v_mfma_f32_16x16x32_bf16, wave tile 8 x 4 tiles (128 accumulators, 12 fragments feed 32 MFMAs: the same operand bytes per MAC)
//   C  as A, all operands zero           (the clock the chip reaches when the multipliers do not toggle)
//   D  as A, operands = third pieces of a bf16x3 split (tiny magnitudes, random mantissas)
//   hipcc --offload-arch=gfx950 -O2 scripts/mfma_power_probe.hip -o /tmp/mfma_power_probe && /tmp/mfma_power_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 1) void k32(const bf16x8* __restrict__ src, float* __restrict__ out, int iters)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = src[((wave * 6 + i) * 64 + lane)];
    for (int j = 0; j < 2; ++j) b[j] = src[((wave * 6 + 4 + j) * 64 + lane)];
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ __launch_bounds__(512, 1) void k16(const bf16x8* __restrict__ src, float* __restrict__ out, int iters)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = src[((wave * 12 + i) * 64 + lane)];
    for (int j = 0; j < 4; ++j) b[j] = src[((wave * 12 + 8 + j) * 64 + lane)];
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    if (s == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = s;
}


typedef float f32x16_ __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512, 1) void kf32_32(const float* __restrict__ src, float* __restrict__ out, int iters)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = src[(wave * 6 + i) * 64 + lane];
    for (int j = 0; j < 2; ++j) b[j] = src[(wave * 6 + 4 + j) * 64 + lane];
    f32x16_ acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(512, 1) void kf32_16(const float* __restrict__ src, float* __restrict__ out, int iters)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = src[(wave * 12 + i) * 64 + lane];
    for (int j = 0; j < 4; ++j) b[j] = src[(wave * 12 + 8 + j) * 64 + lane];
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    if (s == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = s;
}

static unsigned short bf16_of(float x)
{
    unsigned u; memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float f_of(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main()
{
    const int nfrag = 8 * 12 * 64, n = nfrag * 8;
    std::vector<unsigned short> rnd(n), zero(n, 0), third(n);
    srand(7);
    for (int i = 0; i < n; ++i) {
        float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
        const float x = 0.05f * sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);       // small so that 1e5 accumulations stay finite
        const unsigned short h0 = bf16_of(x);
        const float r1 = x - f_of(h0);
        const unsigned short h1 = bf16_of(r1);
        const float r2 = r1 - f_of(h1);
        rnd[i] = h0; third[i] = bf16_of(r2);
    }
    unsigned short* d; float* out;
    hipMalloc(&d, n * 2); hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;
    struct Case { const char* name; int shape; const std::vector<unsigned short>* data; } cases[] = {
        {"A 32x32x16, random hi pieces", 32, &rnd}, {"B 16x16x32, random hi pieces", 16, &rnd},
        {"C 32x32x16, zeros", 32, &zero}, {"D 32x32x16, third pieces", 32, &third}, {"E 16x16x32, zeros", 16, &zero},
        {"A 32x32x16, random hi pieces (again)", 32, &rnd}, {"B 16x16x32, random hi pieces (again)", 16, &rnd}};
    for (const Case& c : cases) {
        hipMemcpy(d, c.data->data(), n * 2, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            if (c.shape == 32) hipLaunchKernelGGL(k32, dim3(256), dim3(512), 0, 0, (const bf16x8*)d, out, iters);
            else hipLaunchKernelGGL(k16, dim3(256), dim3(512), 0, 0, (const bf16x8*)d, out, iters / 2);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // flops: k32: 8 MFMAs x 32*32*16*2 per wave-iteration; k16: 32 MFMAs x 16*16*32*2 (iters / 2 iterations) -> the same total
            const double fl = 256.0 * 8 * (double)iters * 8 * 32768.0;
            if (rep == 2) printf("%-40s %8.3f ms  %7.1f TFLOP/s  (%.3f of 2500)  MFMA-pipe cycles %.3g -> %.3f GHz if the pipe never idles\n", c.name, ms, fl / ms * 1e-9, fl / ms * 1e-9 / 2500.0,
                                 (double)iters * 8 * 32 * 2, (double)iters * 8 * 32 * 2 / (ms * 1e-3) * 1e-9);
        }
    }
    {   // fp32 MFMA shapes on random fp32 operands: 32x32x2 (the exact GEMM stage) vs 16x16x4
        std::vector<float> fr(8 * 12 * 64);
        for (size_t i = 0; i < fr.size(); ++i) fr[i] = 0.05f * ((rand() / (float)RAND_MAX) * 2.f - 1.f);
        float* df; hipMalloc(&df, fr.size() * 4);
        hipMemcpy(df, fr.data(), fr.size() * 4, hipMemcpyHostToDevice);
        const int it32 = 40000;
        for (int which = 0; which < 4; ++which) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                if (which % 2 == 0) hipLaunchKernelGGL(kf32_32, dim3(256), dim3(512), 0, 0, (const float*)df, out, it32);
                else hipLaunchKernelGGL(kf32_16, dim3(256), dim3(512), 0, 0, (const float*)df, out, it32 / 2);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double fl = 256.0 * 8 * (double)it32 * 8 * (32.0 * 32 * 2 * 2);      // 8 MFMAs of 32x32x2 per wave-iteration (= 32 of 16x16x4 per two)
                if (rep == 2) printf("%-40s %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)\n", which % 2 == 0 ? "F 32x32x2 f32, random" : "G 16x16x4 f32, random", ms, fl / ms * 1e-9, fl / ms * 1e-9 / 157.3);
            }
        }
    }
    return 0;
}
