// microbench.hip -- cycles per call of the building blocks of one integration step for each solver
// backend, one solve per workgroup as in the solve kernels (dev tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/microbench tools/microbench.hip && gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../metrotrpl_amd/csrc/backend_wave.hpp"
#include "../metrotrpl_amd/csrc/backend_pair.hpp"
using namespace mtrpl;

enum { RHS = 0, APPLY, BUILD, REDUCE2, REDUCE4, PUBLISH, PREDICT, CORRECT, DENSE, STEP, NWHAT };
static const char *kNames[] = {"rhs", "apply", "build", "reduce<2>", "reduce<4>", "publish", "predict(5)", "correct(5)", "dense+out", "bare step"};

template <class BK, int WHAT>
__device__ inline void body(BK &bk, double *out, long long *cyc, int iters)
{
    constexpr int R = BK::R, V = BK::V;
    double y[V], f[V], w[V], z[6][V], zE[R];
    for (int r = 0; r < R; ++r) {
        const int i = bk.node(r);
        y[r] = 1e-5 * (1.0 + 0.001 * i);
        y[R + r] = 1.2e-5 * (1.0 + 0.0007 * i);
        y[2 * R + r] = 1e-6 * (i - 60);
        zE[r] = 0.0;
    }
    for (int v = 0; v < V; ++v) { w[v] = 0.1 * v + 1e-3 * threadIdx.x; f[v] = 0.0; }
    for (int k = 0; k < 6; ++k)
        for (int v = 0; v < V; ++v) z[k][v] = y[v] * (k == 0 ? 1.0 : 1e-3 / (k + 1));
    bk.publish(y, zE);
    bk.build(y, 0.1, 0.1);
    double acc = 0.0;
    __shared__ double sink[1024];
    bk.out_begin(sink, 0.0, 1.0, 0.0);
    long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (WHAT == RHS) { bk.rhs(y, f); for (int v = 0; v < V; ++v) y[v] += f[v] * 1e-9; }
        if (WHAT == APPLY) { double b[V], x[V]; for (int v = 0; v < V; ++v) b[v] = w[v]; bk.apply(b, x); for (int v = 0; v < V; ++v) w[v] = w[v] * 0.5 + x[v] * 0.5; }
        if (WHAT == BUILD) { bk.build(y, 0.1 + 1e-6 * it, 0.1); y[0] += bk.cd * 1e-30; }
        if (WHAT == REDUCE2) { double s[2] = {w[0] * w[0], w[1] * w[1]}; bk.template reduce<2>(s, y, zE); acc += s[0]; w[0] = w[0] * 0.999 + s[1] * 1e-9; }
        if (WHAT == REDUCE4) { double s[4] = {w[0] * w[0], w[1] * w[1], w[2], w[0]}; bk.template reduce<4>(s, y, zE); acc += s[0] + s[3]; w[0] = w[0] * 0.999 + s[1] * 1e-9 + s[2] * 1e-12; }
        if (WHAT == PUBLISH) { bk.publish(y, zE); y[0] += bk.p.dx * 1e-30; }
        if (WHAT == PREDICT) { nq_predict<5, V>(z); for (int v = 0; v < V; ++v) z[5][v] = z[5][v] * 0.5 + 1e-12; }
        if (WHAT == CORRECT) { nq_correct<5, V>(z, w); for (int v = 0; v < V; ++v) w[v] = z[0][v] * 1e-3; }
        if (WHAT == DENSE) {
            double yq[V];
            nq_dense<5, V>(z, 0.5 + 1e-4 * (it & 15), yq);
            double s = 0.0;
            for (int r = 0; r < R; ++r) s += signal_density(bk.p, 0, yq[r], yq[R + r]);
            bk.out_sample(it & 1023, s);
            for (int v = 0; v < V; ++v) z[0][v] += 1e-30;
        }
        if (WHAT == STEP) {
            // one accepted order-5 step with a single Newton iteration and one output sample, no controller:
            // what the building blocks cost when nothing but data flow connects them
            const double h = 0.5, el0 = NqC<5>::el0;
            double isc[V], acor[V], yn[V], b[V], dx[V];
            nq_predict<5, V>(z);
            for (int v = 0; v < V; ++v) { isc[v] = ctl_rcp(1e-20 + 3e-9 * fabs(z[0][v])); acor[v] = 0.0; yn[v] = z[0][v]; }
            bk.publish(yn, zE);
            bk.rhs(yn, f);
            for (int v = 0; v < V; ++v) b[v] = h * f[v] - z[1][v] - acor[v];
            bk.apply(b, dx);
            double s2[2] = {0.0, 0.0};
            for (int v = 0; v < V; ++v) {
                const double wv = el0 * dx[v] * isc[v];
                s2[0] += wv * wv;
                acor[v] += dx[v] * 1e-9;
                yn[v] += el0 * dx[v];
                const double av = acor[v] * isc[v];
                s2[1] += av * av;
            }
            bk.template reduce<2>(s2, yn, zE);
            acc += s2[0] + s2[1];
            nq_correct<5, V>(z, acor);
            double yq[V];
            nq_dense<5, V>(z, -0.25, yq);
            double sp = 0.0;
            for (int r = 0; r < R; ++r) sp += signal_density(bk.p, 0, yq[r], yq[R + r]);
            bk.out_sample(it & 1023, sp);
            for (int k = 1; k < 6; ++k) for (int v = 0; v < V; ++v) z[k][v] *= 0.5;  // keep the history bounded
        }
    }
    bk.out_finish();
    asm volatile("" ::"v"(acc), "v"(y[0]), "v"(w[0]), "v"(z[0][0]), "v"(z[5][0]), "v"(f[0]) : "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double s = acc;
    for (int v = 0; v < V; ++v) s += y[v] + w[v] + f[v] + z[3][v];
    if (threadIdx.x == 0) { out[blockIdx.x] = s + sink[0]; cyc[blockIdx.x] = t1 - t0; }
}

template <int WHAT, bool F32BLK>
__global__ __launch_bounds__(64, 1) void bench_wave(const double *par, double *out, long long *cyc, int iters)
{
    __shared__ double2 fac_lds[kFactorLdsBytes / 16];
    WaveBackendT<F32BLK> bk(threadIdx.x, fac_lds);
    bk.set_problem(make_par<double>(par, 2000.0 / 128), 128);
    body<WaveBackendT<F32BLK>, WHAT>(bk, out, cyc, iters);
}
template <int WHAT, int NREG>
__global__ __launch_bounds__(128, 1) void bench_pair(const double *par, double *out, long long *cyc, int iters)
{
    using BK = PairBackend<2, NREG>;
    __shared__ double2 blk_lds[BK::kBlkBytes > 0 ? BK::kBlkBytes / 16 : 1];
    __shared__ double2 xs_lds[BK::kXchBytes / 16];
    BK bk(threadIdx.x, blk_lds, reinterpret_cast<double *>(xs_lds));
    bk.set_problem(make_par<double>(par, 2000.0 / 128), 128);
    body<BK, WHAT>(bk, out, cyc, iters);
}

int main(int argc, char **argv)
{
    const int nblk = argc > 1 ? atoi(argv[1]) : 256;  // workgroups in flight (256: one solve per CU)
    double par[16] = {1e-13, 3e-6, 2e6, 2e6, 48, 4.4e4, 4.4e4, 0.1, 0.1, 511, 871, 10, 300, 0, 0, 1};
    double *d_par, *d_out;
    long long *d_cyc;
    hipMalloc(&d_par, sizeof(par)); hipMalloc(&d_out, 4096 * 8); hipMalloc(&d_cyc, 4096 * 8);
    hipMemcpy(d_par, par, sizeof(par), hipMemcpyHostToDevice);
    const int iters = 2000;
    std::vector<long long> c(4096);
    double res[3][NWHAT];
#define RUN(K, W, KERN, T)                                                                            \
    {                                                                                                 \
        for (int rep = 0; rep < 2; ++rep) {                                                           \
            hipLaunchKernelGGL(KERN, dim3(nblk), dim3(T), 0, 0, d_par, d_out, d_cyc, iters);          \
            hipDeviceSynchronize();                                                                   \
        }                                                                                             \
        hipMemcpy(c.data(), d_cyc, nblk * 8, hipMemcpyDeviceToHost);                                  \
        double m = 0;                                                                                 \
        for (int i = 0; i < nblk; ++i) m += c[i];                                                     \
        res[K][W] = m / nblk / iters;                                                                 \
    }
#define BOTH(W) RUN(0, W, (bench_wave<W, true>), 64) RUN(1, W, (bench_wave<W, false>), 64) RUN(2, W, (bench_pair<W, 3>), 128)
    BOTH(RHS) BOTH(APPLY) BOTH(BUILD) BOTH(REDUCE2) BOTH(REDUCE4) BOTH(PUBLISH) BOTH(PREDICT) BOTH(CORRECT) BOTH(DENSE) BOTH(STEP)
    printf("%-12s %12s %12s %12s   (cycles per call, %d workgroups)\n", "block", "wave fp32blk", "wave fp64blk", "pair<2,3>", nblk);
    for (int w = 0; w < NWHAT; ++w) printf("%-12s %12.0f %12.0f %12.0f\n", kNames[w], res[0][w], res[1][w], res[2][w]);
    return 0;
}
