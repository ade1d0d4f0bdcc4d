// microbench.hip -- cycles per call of the building blocks of one integration step, one wave per
// SIMD as in the solve kernel (dev tool).  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../metrotrpl_amd/csrc/solve_kernel.hpp"
using namespace mtrpl;

template <int WHAT>
__global__ __launch_bounds__(64, 1) void bench(const double *par, double *out, long long *cyc, int iters)
{
    constexpr int R = 2, V = 6;
    const int lane = threadIdx.x, L = 128;
    const Par p = make_par(par, 2000.0 / 128);
    double y[V], f[V], w[V], D[8][V];
    for (int v = 0; v < V; ++v) { y[v] = (v < 4) ? 1e-5 * (1.0 + 0.001 * lane + 0.01 * v) : 1e-6 * (lane - 32); w[v] = 0.1 * v + lane; }
    for (int k = 0; k < 8; ++k) for (int v = 0; v < V; ++v) D[k][v] = y[v] * (k == 0 ? 1.0 : 1e-3 / (k + 1));
    __shared__ double2 fac_lds[kFactorLdsBytes / 16];
    Factors<R> F; F.lds = fac_lds;
    build_factors(p, y, 0.1, lane, L, F);
    double acc = 0.0;
    int order = 5 - (iters & 0);  // runtime value
    long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        if (WHAT == 0) { const double n_ = wnorm<V>(w, 1.0 / 385); acc += n_; for (int v = 0; v < V; ++v) w[v] = w[v] * 0.999 + n_ * 1e-3; }
        if (WHAT == 1) { rhs_std<R>(p, y, f, lane, L); for (int v = 0; v < V; ++v) y[v] += f[v] * 1e-9; }
        if (WHAT == 2) { double b[V], x[V]; for (int v = 0; v < V; ++v) b[v] = w[v]; apply_factors(F, b, x, lane, L); for (int v = 0; v < V; ++v) w[v] = w[v] * 0.5 + x[v] * 0.5; }
        if (WHAT == 3) { double yp[V], psi[V]; MTRPL_ORDER_SWITCH(order, (ndf_predict<K, V>(D, yp, psi))) for (int v = 0; v < V; ++v) { D[1][v] += yp[v] * 1e-9; D[2][v] += psi[v] * 1e-9; } }
        if (WHAT == 4) { MTRPL_ORDER_SWITCH(order, (ndf_update<K, V>(D, w))) for (int v = 0; v < V; ++v) w[v] = D[0][v] * 1e-3; }
        if (WHAT == 5) { double yq[V]; MTRPL_ORDER_SWITCH(order, (ndf_dense<K, V>(D, 1.0 + 1e-3 * it, 1.5, 1.0, 1.0, yq))) const double sg_ = signal_of<R>(p, yq, 0, lane, L, 1e23); acc += sg_; for (int v = 0; v < V; ++v) D[0][v] += sg_ * 1e-40; }
        if (WHAT == 6) { build_factors(p, y, 0.1 + 1e-6 * it, lane, L, F); y[0] += F.cd * 1e-30; }
        if (WHAT == 7) { change_D<V>(D, order, 1.0 + 1e-3 * (it & 7), lane); }
        if (WHAT == 8) { double sm, sp; MTRPL_ORDER_SWITCH(order, (ndf_sel<K, V>(D, w, sm, sp))) acc += uniform(norm_sqrt(wave_sum(sm))) + uniform(norm_sqrt(wave_sum(sp))); for (int v = 0; v < V; ++v) { D[5][v] += acc * 1e-12; D[7][v] += acc * 1e-12; } }
        if (WHAT == 9) { const double n_ = wave_sum(w[0]); acc += n_; w[0] = w[0] * 0.999 + n_ * 1e-3; }
    }
    asm volatile("" :: "v"(acc), "v"(y[0]), "v"(y[3]), "v"(w[0]), "v"(w[5]), "v"(D[0][0]), "v"(D[1][0]), "v"(D[5][0]), "v"(D[7][5]) : "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = acc;
    for (int v = 0; v < V; ++v) s += y[v] + w[v] + f[v] * 0 + D[3][v];
    if (lane == 0) { out[blockIdx.x] = s; cyc[blockIdx.x] = t1 - t0; }
}

int main()
{
    double par[16] = {1e-13, 3e-6, 2e6, 2e6, 48, 4.4e4, 4.4e4, 0.1, 0.1, 511, 871, 10, 300, 0, 0, 1};
    double *d_par, *d_out; long long *d_cyc;
    hipMalloc(&d_par, sizeof(par)); hipMalloc(&d_out, 1024 * 8); hipMalloc(&d_cyc, 1024 * 8);
    hipMemcpy(d_par, par, sizeof(par), hipMemcpyHostToDevice);
    const char *names[] = {"wnorm", "rhs_std", "apply_factors", "predict(5)", "update(5)", "dense+signal", "build_factors", "change_D", "sel norms", "wave_sum"};
    const int iters = 2000;
    std::vector<long long> c(1024);
#define RUN(W) { hipLaunchKernelGGL(bench<W>, dim3(1024), dim3(64), 0, 0, d_par, d_out, d_cyc, iters); hipDeviceSynchronize(); \
    hipLaunchKernelGGL(bench<W>, dim3(1024), dim3(64), 0, 0, d_par, d_out, d_cyc, iters); hipDeviceSynchronize(); \
    hipMemcpy(c.data(), d_cyc, 1024 * 8, hipMemcpyDeviceToHost); double m = 0; for (auto x : c) m += x; \
    printf("%-14s %8.0f cycles/call (1024 waves, one per SIMD)\n", names[W], m / 1024 / iters); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
    return 0;
}
