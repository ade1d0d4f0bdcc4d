// What does one LDS exchange between the W waves of a workgroup cost (publish 2 doubles, s_waitcnt lgkmcnt(0); s_barrier, read
// the neighbour's, wait), with K fp64 FMAs of independent work per wave between two exchanges?  One workgroup per CU
// (80 KB of LDS each) and two (40 KB).  hipcc --offload-arch=gfx950 -O3 -o gpurun_out/xchg_cost tools/mb/xchg_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int W, int K, bool XCHG, int LDSB>
__global__ __launch_bounds__(64 * W) void k(double *out, unsigned long long *cyc, int n)
{
    __shared__ double pad[LDSB / 8];
    double *xb = pad;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + 0.001 * (threadIdx.x + i);
    int par = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < K; ++j) a[j & 7] = fma(a[j & 7], 1.0000001, 1e-9);
        if (XCHG) {
            if (lane == 0) { xb[(par * W + wave) * 2] = a[0]; xb[(par * W + wave) * 2 + 1] = a[1]; }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int nb = (wave + 1) % W;
            const double u = xb[(par * W + nb) * 2], v = xb[(par * W + nb) * 2 + 1];
            a[2] += 1e-30 * u;
            a[3] += 1e-30 * v;
            par ^= 1;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 64 * W + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int W, int K, int LDSB> void run(double *out, unsigned long long *cyc, const char *what)
{
    const int n = 20000, G = 256;
    unsigned long long h[1024];
    double r[2];
    for (int x = 0; x < 2; ++x) {
        if (x) hipLaunchKernelGGL((k<W, K, true, LDSB>), dim3(G), dim3(64 * W), 0, 0, out, cyc, n);
        else hipLaunchKernelGGL((k<W, K, false, LDSB>), dim3(G), dim3(64 * W), 0, 0, out, cyc, n);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 8 * G, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < G; ++i) m += h[i];
        r[x] = m / G / n;
    }
    printf("%-34s W=%d K=%3d: %7.1f ticks per pass without, %7.1f with the exchange: %6.1f per exchange\n", what, W, K, r[0], r[1], r[1] - r[0]);
}
int main()
{
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 8 * 1024 * 256 * 8); hipMalloc(&cyc, 8 * 1024 * 8);
    run<2, 16, 81920>(out, cyc, "one workgroup per CU");
    run<2, 64, 81920>(out, cyc, "one workgroup per CU");
    run<2, 128, 81920>(out, cyc, "one workgroup per CU");
    run<4, 64, 81920>(out, cyc, "one workgroup per CU");
    run<2, 64, 40960>(out, cyc, "two workgroups per CU (512 WGs: no)");
    run<1, 64, 81920>(out, cyc, "one wave (barrier of one)");
    return 0;
}
