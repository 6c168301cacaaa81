// Development micro-benchmark: how many single-wave workgroups of a K1a-like kernel (a ~6 500-instruction
// dependent fp64 chain per wave) run side by side on gfx950, and what scratch, dynamic LDS and a register
// budget of three waves per SIMD do to that.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_wave_conc tools_micro/wave_concurrency.hip && ./gpurun_wave_conc
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool SCRATCH, bool LDS>
__device__ __forceinline__ void body(double *out, int iters, double seed)
{
    extern __shared__ double sh[];
    volatile double spill[2];
    double a[4];
    for (int i = 0; i < 4; i++) a[i] = seed + i * 1e-3 + threadIdx.x * 1e-6;
    if (SCRATCH) { spill[0] = a[0]; spill[1] = a[1]; }
    if (LDS) sh[threadIdx.x] = a[2];
    const double m = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = fma(a[i], m, c);
    }
    double s = 0; for (int i = 0; i < 4; i++) s += a[i];
    if (SCRATCH) s += spill[threadIdx.x & 1];
    if (LDS) s += sh[63 - threadIdx.x];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <bool SCRATCH, bool LDS>
__global__ void __launch_bounds__(64) k_free(double *out, int iters, double seed) { body<SCRATCH, LDS>(out, iters, seed); }
template <bool SCRATCH, bool LDS>
__global__ void __launch_bounds__(64) k_168(double *out, int iters, double seed)
{
    asm volatile("" ::: "v167");                             // the kernel is allocated 168 registers: three waves per SIMD
    body<SCRATCH, LDS>(out, iters, seed);
}
template <class K> void run(const char *name, K kern, size_t lds)
{
    double *out; hipMalloc(&out, 8192 * 64 * 8);
    const int iters = 1625;                                  // x 4 chains = 6 500 fma per wave
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-34s", name);
    for (int blocks : {256, 512, 1024, 1536, 2048, 3072, 4096, 6144}) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, out, iters, 1.0); hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, out, iters, 1.0); hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf(" %5d:%6.1f", blocks, best * 1e3);
    }
    printf("  us\n");
    hipFree(out);
}
int main()
{
    printf("launch time against the number of single-wave workgroups (1 024 SIMDs):\n");
    run("no scratch, no LDS", k_free<false, false>, 0);
    run("scratch 16 B/lane", k_free<true, false>, 0);
    run("dynamic LDS 10.5 KB", k_free<false, true>, 10496);
    run("scratch + LDS", k_free<true, true>, 10496);
    run("168 VGPR", k_168<false, false>, 0);
    run("168 VGPR, scratch", k_168<true, false>, 0);
    run("168 VGPR, LDS 10.5 KB", k_168<false, true>, 10496);
    run("168 VGPR, scratch + LDS", k_168<true, true>, 10496);
    return 0;
}
