// Development micro-benchmark: does the LAYOUT of an agent's rows matter to a kernel that reads them
// the way step_kernel does?  One wavefront per agent at a time, 4 waves per workgroup walking 64 consecutive agents, the
// next agent's rows in flight while the current one is worked on (the step kernel's software pipeline), ~VALU fma per
// agent-step as a stand-in for the state machine, four waves per SIMD (128 registers), one 512-B record + ROWS rows of
// n = 40 doubles read, the record + two rows written back -- and PAIRS history pairs (S row + Y row) read through
// global_load_lds as hist_dma does.
//   layout 0: as the library has it -- rec [B][64], six separate [B][n] arrays, S [B][M][n] and Y [B][M][n]
//   layout 1: ONE block per agent [B][64 + 6 n] (the record and its rows contiguous: 2 432 B), S/Y interleaved per pair
//             [B][M][2][n] (one ring slot = one 640-B burst)
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_row_layout tools/micro/row_layout.hip && ./gpurun_row_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int N = 40, REC = 64, ROWS = 6, M = 20;
struct Ptrs { double *rec; double *row[ROWS]; double *S, *Y; size_t rstride, rec_stride, hs; };
__device__ __forceinline__ void dma(const double *g, double *lds, int bytes, int lane)
{
    for (int off = 0; off < bytes; off += 1024) {
        const int my = off + lane * 16;
        if (my < bytes)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)g + my),
                                             (__attribute__((address_space(3))) void *)((char *)lds + off), 16, 0, 0);
    }
}
template <int LAYOUT>
__global__ void __launch_bounds__(256, 4) walk(Ptrs p, int B, int fmas, int pairs)
{
    extern __shared__ double sh[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *hist = sh + (size_t)wv * 2 * 15 * N;
    const int base = blockIdx.x * 64;
    struct In { double r; double x[ROWS]; };
    auto load = [&](int a) {
        In in;
        in.r = p.rec[(size_t)a * p.rec_stride + lane];
#pragma unroll
        for (int k = 0; k < ROWS; k++) in.x[k] = lane < N ? p.row[k][(size_t)a * p.rstride + lane] : 0.0;
        return in;
    };
    In nxt = load(base + wv);
    for (int i = wv; i < 64; i += 4) {
        const int a = base + i;
        const In cur = nxt;
        if (i + 4 < 64) nxt = load(a + 4);
        if (pairs > 0) {
            if (LAYOUT == 0) {
                dma(p.S + (size_t)a * M * N, hist, pairs * N * 8, lane);
                dma(p.Y + (size_t)a * M * N, hist + 15 * N, pairs * N * 8, lane);
            } else {
                dma(p.S + (size_t)a * M * 2 * N, hist, pairs * 2 * N * 8, lane);
            }
        }
        double acc = cur.r;
#pragma unroll
        for (int k = 0; k < ROWS; k++) acc += cur.x[k];
        for (int f = 0; f < fmas; f++) acc = fma(acc, 1.0000001, 1e-9);
        if (pairs > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int t = 0; t < pairs; t++) acc += hist[(t * (LAYOUT == 0 ? 1 : 2)) * N + (lane < N ? lane : 0)];
        }
        p.rec[(size_t)a * p.rec_stride + lane] = acc;
        if (lane < N) { p.row[0][(size_t)a * p.rstride + lane] = acc; p.row[4][(size_t)a * p.rstride + lane] = acc * 0.5; }
    }
}
int main()
{
    const int B = 65536;
    double *arena; const size_t per = REC + ROWS * N + 2 * M * N;
    hipMalloc(&arena, sizeof(double) * per * B + (1 << 20));
    hipMemset(arena, 0, sizeof(double) * per * B);
    Ptrs p0{}, p1{};
    {   // layout 0
        double *d = arena;
        p0.rec = d; d += (size_t)REC * B; p0.rec_stride = REC;
        for (int k = 0; k < ROWS; k++) { p0.row[k] = d; d += (size_t)N * B; }
        p0.rstride = N; p0.S = d; d += (size_t)M * N * B; p0.Y = d; p0.hs = N;
    }
    {   // layout 1
        double *d = arena;
        const size_t blk = REC + ROWS * N;
        p1.rec = d; p1.rec_stride = blk;
        for (int k = 0; k < ROWS; k++) p1.row[k] = d + REC + k * N;
        p1.rstride = blk; d += blk * B; p1.S = d; p1.Y = d + N; p1.hs = 2 * N;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = sizeof(double) * 4 * 2 * 15 * N;
    printf("B = %d agents, one launch = every agent once (step_kernel's full round); us per launch, best of 7\n", B);
    for (int pairs : {0, 5, 10}) for (int fmas : {0, 150, 550}) {
        float best[2] = {1e9f, 1e9f};
        for (int rep = 0; rep < 8; rep++) for (int lay = 0; lay < 2; lay++) {
            hipEventRecord(e0);
            if (lay == 0) hipLaunchKernelGGL(walk<0>, dim3(B / 64), dim3(256), lds, 0, p0, B, fmas, pairs);
            else hipLaunchKernelGGL(walk<1>, dim3(B / 64), dim3(256), lds, 0, p1, B, fmas, pairs);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best[lay]) best[lay] = ms;
        }
        const double bytes = (double)B * (8.0 * (2 * REC + (ROWS + 2) * N) + pairs * 2.0 * N * 8);
        printf("history pairs %2d, %3d dependent fma per agent-step: separate arrays %7.1f us (%.2f TB/s)   one block + interleaved pairs %7.1f us (%.2f TB/s)   ratio %.3f\n",
               pairs, fmas, best[0] * 1e3, bytes / (best[0] * 1e-3) / 1e12, best[1] * 1e3, bytes / (best[1] * 1e-3) / 1e12, best[1] / best[0]);
    }
    return 0;
}
