// Development micro-benchmark: issue rate of fp64 VALU instructions on one SIMD (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_fp64_rate tools_micro/fp64_rate.hip && ./gpurun_fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void __launch_bounds__(64) rate_kernel(double *out, long long *cyc, int iters, double seed)
{
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + i * 1e-3 + threadIdx.x * 1e-6;
    const double m = 1.0000001, c = 1e-9;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = fma(a[i], m, c);
            else if (OP == 1) a[i] = a[i] + c;
            else if (OP == 2) a[i] = a[i] * m;
            else if (OP == 3) a[i] = __builtin_amdgcn_rcp(a[i]);
            else if (OP == 4) { float f = (float)a[i]; f = fmaf(f, 1.0000001f, 1e-9f); a[i] = f; }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char *name, int blocks)
{
    double *out; long long *cyc; hipMalloc(&out, blocks * 64 * 8); hipMalloc(&cyc, blocks * 8);
    const int iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<OP><<<blocks, 64>>>(out, cyc, iters, 1.0); hipDeviceSynchronize();
    hipEventRecord(e0); rate_kernel<OP><<<blocks, 64>>>(out, cyc, iters, 1.0); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-10s blocks %6d: %.1f us, %.2f ns per wave-instruction per wave (8 independent chains), cycle counter %lld per %d instr\n", name, blocks, ms * 1e3,
           ms * 1e6 / (iters * 8.0), h[0], iters * 8);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int blocks : {1, 1024, 2048, 4096, 8192}) {
        run<0>("fma_f64", blocks); run<1>("add_f64", blocks); run<2>("mul_f64", blocks); run<3>("rcp_f64", blocks); run<4>("fma_f32", blocks);
    }
    return 0;
}
