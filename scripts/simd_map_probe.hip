// Which SIMD does wave w of a 512-thread workgroup land on?  (gfx950; HW_REG_HW_ID bits 5:4 = simd_id, 11:8 = cu_id)
//   hipcc --offload-arch=gfx950 -O2 scripts/simd_map_probe.hip -o scripts/_build/simd_map_probe && scripts/_build/simd_map_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512, 1) void probe(unsigned* out)
{
    const unsigned id = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);    // HW_ID[15:0]
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main()
{
    unsigned* d; const int nb = 512;
    hipMalloc(&d, nb * 8 * 4);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 100 * 1024, 0, d);
    unsigned h[nb * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int pat[8][4] = {};
    for (int b = 0; b < nb; ++b) {
        if (b < 6) { printf("block %d:", b); for (int w = 0; w < 8; ++w) printf(" w%d->simd%u(cu%u)", w, (h[b * 8 + w] >> 4) & 3, (h[b * 8 + w] >> 8) & 15); printf("\n"); }
        for (int w = 0; w < 8; ++w) pat[w][(h[b * 8 + w] >> 4) & 3]++;
    }
    int same04 = 0, same01 = 0, same02 = 0;
    for (int b = 0; b < nb; ++b) {
        same04 += ((h[b * 8] >> 4) & 3) == ((h[b * 8 + 4] >> 4) & 3);
        same01 += ((h[b * 8] >> 4) & 3) == ((h[b * 8 + 1] >> 4) & 3);
        same02 += ((h[b * 8] >> 4) & 3) == ((h[b * 8 + 2] >> 4) & 3);
    }
    printf("of %d workgroups: wave 0 shares its SIMD with wave 4 in %d, with wave 1 in %d, with wave 2 in %d\n", nb, same04, same01, same02);
    return 0;
}
