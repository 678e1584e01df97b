// Workgroup dispatch cost on MI355X: how long does a launch of N workgroups x 256 threads take when every
// workgroup exits at once, or after one scalar load?  (Sizing question for the resampler's interleaved launch,
// whose non-candidate sampler workgroups do exactly that.)
// build: hipcc --offload-arch=gfx950 -O3 -o dispatch_bench scripts/dispatch_bench.hip ; run: ./dispatch_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void k_empty(const unsigned* p, unsigned* out) {}

__global__ __launch_bounds__(256) void k_sload(const unsigned* p, unsigned* out)
{
    if (p[blockIdx.x >> 4] == 0xdeadbeefu) out[threadIdx.x] = 1u;      // uniform address -> s_load, never true
}

template <int WG>
__global__ __launch_bounds__(WG) void k_sload_wg(const unsigned* p, unsigned* out)
{
    if (p[blockIdx.x >> 4] == 0xdeadbeefu) out[threadIdx.x] = 1u;
}

int main()
{
    unsigned *p, *out;
    hipMalloc(&p, 1 << 22); hipMemset(p, 0, 1 << 22);
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int sizes[] = {256, 3072, 6144, 12288, 26112, 52224, 104448};
    for (int variant = 0; variant < 4; ++variant)
        for (int n : sizes) {
            float best = 1e9f;
            for (int rep = 0; rep < 8; ++rep) {
                hipEventRecord(e0);
                for (int it = 0; it < 20; ++it) {
                    if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(n), dim3(256), 0, 0, p, out);
                    else if (variant == 1) hipLaunchKernelGGL(k_sload, dim3(n), dim3(256), 0, 0, p, out);
                    else if (variant == 2) hipLaunchKernelGGL(k_sload_wg<64>, dim3(n), dim3(64), 0, 0, p, out);
                    else hipLaunchKernelGGL(k_sload_wg<1024>, dim3(n / 4), dim3(1024), 0, 0, p, out);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const char* names[] = {"empty wg256", "s_load+exit wg256", "s_load+exit wg64", "s_load+exit wg1024 (n/4 wgs)"};
            printf("%-30s %7d workgroups: %7.2f us per launch  (%.2f ns per workgroup)\n", names[variant],
                   variant == 3 ? n / 4 : n, best / 20 * 1e3, best / 20 * 1e6 / (variant == 3 ? n / 4 : n));
        }
    return 0;
}
