// Which kernels run NEXT TO the persistent Winograd GEMM stage (one 512-thread workgroup per CU, 2 waves of 216 VGPRs per SIMD,
// 128 KiB LDS) when launched on a second stream?  A streaming copy kernel whose register allocation is forced to N VGPRs and
// whose workgroup size is WG, timed alone, and together with 5 GEMM stages (res2 shape at batch 12).
// build: hipcc --offload-arch=gfx950 -O3 -o coresidency_probe scripts/coresidency_probe.hip -Iinclude -Lrendernet_amd/lib -lrendernet_hip
// run:   LD_LIBRARY_PATH=rendernet_amd/lib ./coresidency_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include "rendernet_hip.h"

template <int NV, int WG>
__global__ __launch_bounds__(WG) void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
    if (NV == 16) asm volatile("v_mov_b32 v15, 0" ::: "v15");
    if (NV == 32) asm volatile("v_mov_b32 v31, 0" ::: "v31");
    if (NV == 40) asm volatile("v_mov_b32 v39, 0" ::: "v39");
    if (NV == 48) asm volatile("v_mov_b32 v47, 0" ::: "v47");
    if (NV == 56) asm volatile("v_mov_b32 v55, 0" ::: "v55");
    if (NV == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (NV == 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");
    if (NV == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    if (NV == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) dst[i] = src[i];
}

static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const long long T = 12 * 11 * 11;
    const int C = 1024, NXI = 64;
    float *V, *U, *M; float4 *a, *b;
    const size_t vn = (size_t)NXI * T * C, un = (size_t)NXI * C * C, cn = (size_t)48 << 20;      // copy: 768 MB in, 768 MB out
    hipMalloc(&V, vn * 4); hipMalloc(&M, vn * 4); hipMalloc(&U, un * 4); hipMalloc(&a, cn * 16); hipMalloc(&b, cn * 16);
    hipMemset(V, 0x3c, vn * 4); hipMemset(U, 0x3c, un * 4); hipMemset(a, 1, cn * 16);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    auto gemm = [&]() { if (rn_winograd_gemm(RN_WINO_F63, V, U, M, T, C, C, s1) != 0) { printf("gemm: %s\n", rn_last_error()); exit(1); } };
    auto wall = [&](auto f) {
        double best = 1e30;
        for (int r = 0; r < 3; ++r) { hipDeviceSynchronize(); const double t0 = now(); f(); hipDeviceSynchronize(); const double t = now() - t0; if (t < best) best = t; }
        return best;
    };
    gemm(); gemm();
    const double tg = wall([&]() { for (int i = 0; i < 5; ++i) gemm(); });
    printf("5 GEMM stages alone: %.3f ms\n", tg);
#define PROBE(NV, WG)                                                                                              \
    {                                                                                                              \
        auto cp = [&]() { hipLaunchKernelGGL((k_copy<NV, WG>), dim3(256 * 16), dim3(WG), 0, s2, a, b, cn); };      \
        cp();                                                                                                      \
        const double tc = wall([&]() { for (int i = 0; i < 10; ++i) cp(); });                                      \
        const double tb = wall([&]() { for (int i = 0; i < 5; ++i) { gemm(); cp(); cp(); } });                     \
        printf("copy NV=%3d WG=%4d: 10 alone %.3f ms, with 5 GEMM stages %.3f ms (sum %.3f, hidden %.0f %% of the copies)\n", NV, WG, tc, tb, tg + tc, \
               100.0 * (tg + tc - tb) / tc);                                                                       \
    }
    PROBE(16, 256) PROBE(32, 256) PROBE(40, 256) PROBE(48, 256) PROBE(56, 256) PROBE(64, 256) PROBE(72, 256) PROBE(80, 256) PROBE(96, 256)
    PROBE(32, 64) PROBE(64, 64) PROBE(80, 64) PROBE(32, 1024)
    return 0;
}
