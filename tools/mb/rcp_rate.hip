// issue cost of v_rcp_f64 / v_rcp_f32 / v_fma_f64 / v_sqrt_f64 / v_log_f32 on gfx950: one wave per SIMD, independent chains
// hipcc --offload-arch=gfx950 -O3 -o gpurun_out/rcp_rate tools/mb/rcp_rate.hip && gpurun_out/rcp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(64) void k(double *out, unsigned long long *cyc, int n)
{
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + 0.001 * (threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_rcp(a[i]);
            if (OP == 1) a[i] = (double)__builtin_amdgcn_rcpf((float)a[i]);
            if (OP == 2) a[i] = fma(a[i], 1.0000001, 1e-9);
            if (OP == 3) a[i] = __builtin_amdgcn_sqrt(a[i]);
            if (OP == 4) { float f = (float)a[i]; f = __builtin_amdgcn_rcpf(f); a[i] = (double)f; }
            if (OP == 5) { float f = (float)a[i]; f = __builtin_amdgcn_rcpf(f); double r = (double)f; r = fma(fma(-a[i], r, 1.0), r, r); r = fma(fma(-a[i], r, 1.0), r, r); a[i] = r + 1.0; }
            if (OP == 6) { double r = __builtin_amdgcn_rcp(a[i]); r = fma(fma(-a[i], r, 1.0), r, r); a[i] = r + 1.0; }
            if (OP == 7) a[i] = a[i] + 1.000001;
            if (OP == 8) { v2f p = __builtin_bit_cast(v2f, a[i]); p = p * v2f{1.0000001f, 0.9999999f} + v2f{1e-9f, 1e-9f}; a[i] = __builtin_bit_cast(double, p); }
            if (OP == 9) { float f = __builtin_bit_cast(v2f, a[i]).x; f = fmaf(f, 1.0000001f, 1e-9f); v2f p{f, f}; a[i] = __builtin_bit_cast(double, p); }
            if (OP == 10) { float f = (float)a[i]; asm volatile("" : "+v"(f)); a[i] = (double)f; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 1;  // waves per SIMD
    double *out; unsigned long long *cyc, h[1024];
    hipMalloc(&out, 8 * 1024 * 64 * 8); hipMalloc(&cyc, 8 * 1024 * 8);
    const int n = 20000;
    const char *names[] = {"v_rcp_f64", "cvt+v_rcp_f32+cvt", "v_fma_f64", "v_sqrt_f64", "cvt,rcp_f32,cvt", "f32 seed + 2 NR + add", "rcp_f64 + 1 NR + add", "v_add_f64", "v_pk_fma_f32", "v_fma_f32", "v_cvt_f32_f64 + v_cvt_f64_f32"};
#define RUN(OP) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(64), 0, 0, out, cyc, n); hipDeviceSynchronize(); hipEventRecord(e0, 0); hipLaunchKernelGGL(k<OP>, dim3(1024 * W), dim3(64), 0, 0, out, cyc, n); hipEventRecord(e1, 0); hipDeviceSynchronize(); float ms = 0; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, 8192, hipMemcpyDeviceToHost); double m = 0; for (int i = 0; i < 1024; ++i) m += h[i]; printf("%-26s %7.2f s_memtime ticks per wave-op, %7.2f ns per wave-op per SIMD by the wall clock\n", names[OP], m / 1024 / n / 8, ms * 1e6 / n / 8 / W); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10)
    return 0;
}
