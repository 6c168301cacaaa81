// Micro-benchmark (development tool, not product code): what a wave pays per fp64 instruction on gfx950 as a function of the
// independent chains it interleaves (ILP) and of the waves that share its SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_fp64_chain tools/micro/fp64_chain.hip && ./gpurun_fp64_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int ILP, int OP>
__global__ void __launch_bounds__(64) chain(double *out, int iters, double a, double b, double one, double nzero)
{
    double v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) v[i] = a + 1e-3 * (threadIdx.x + 64 * i);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (OP == 0) v[i] = __builtin_fma(v[i], a, b);          // v_fma_f64
                else if (OP == 1) v[i] = v[i] * a;                      // v_mul_f64
                else if (OP == 2) v[i] = v[i] + b;                      // v_add_f64
                else if (OP == 3) v[i] = b / v[i];                      // the division sequence
                else if (OP == 4) v[i] = __builtin_amdgcn_rcp(v[i]);    // v_rcp_f64
                else if (OP == 5) v[i] = v[i] > b ? v[i] * a : v[i] + b; // compare + select + two ops
                else if (OP == 6) v[i] = __builtin_fma(v[i], one, b);   // v + b as an fma (one = 1.0 at run time: not folded)
                else if (OP == 7) v[i] = __builtin_fma(v[i], a, nzero); // v * a as an fma (nzero = -0.0 at run time)
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int ILP, int OP>
static int run(const char *name, double *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((chain<ILP, OP>), dim3(blocks), dim3(64), 0, 0, d, iters / 10, 0.999999, 1e-7, 1.0, -0.0);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((chain<ILP, OP>), dim3(blocks), dim3(64), 0, 0, d, iters, 0.999999, 1e-7, 1.0, -0.0);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)iters * 8 * ILP;               // per wave
    printf("%-6s ILP %d blocks %5d: %8.3f ms  %6.2f ns per op per wave = %5.1f cycles at 2.4 GHz; per SIMD-op %5.2f cycles\n", name, ILP, blocks, ms,
           ms * 1e6 / ops, ms * 1e6 / ops * 2.4, ms * 1e6 * 2.4 / (ops * (blocks > 1024 ? blocks / 1024.0 : 1.0)));
    return 0;
}

#define ROW(OP, name, iters) \
    for (int blocks : {1, 1024, 2048, 4096}) { \
        if (run<1, OP>(name, d, blocks, iters)) return 1; \
        if (run<2, OP>(name, d, blocks, iters)) return 1; \
        if (run<3, OP>(name, d, blocks, iters)) return 1; \
        if (run<4, OP>(name, d, blocks, iters)) return 1; \
        if (run<8, OP>(name, d, blocks, iters)) return 1; \
    }

int main()
{
    double *d;
    CHK(hipMalloc(&d, 8192 * 64 * 8));
    ROW(0, "fma", 100000)
    ROW(1, "mul", 100000)
    ROW(2, "add", 100000)
    ROW(3, "div", 10000)
    ROW(4, "rcp", 50000)
    ROW(5, "sel", 50000)
    ROW(6, "addfma", 100000)
    ROW(7, "mulfma", 100000)
    CHK(hipFree(d));
    return 0;
}
