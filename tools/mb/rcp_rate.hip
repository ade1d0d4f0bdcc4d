// issue cost of v_rcp_f64 / v_rcp_f32 / v_fma_f64 / v_sqrt_f64 / v_log_f32 on gfx950: one wave per SIMD, independent chains
// hipcc --offload-arch=gfx950 -O3 -o gpurun_out/rcp_rate tools/mb/rcp_rate.hip && gpurun_out/rcp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int OP, int U> __global__ __launch_bounds__(64) void k(double *out, unsigned long long *cyc, int n)
{
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + 0.001 * (threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // (U passes over the eight chains per trip of the loop: with U = 1, as rounds 3-5 measured, the loop's own scalar
    // compare and branch -- about 20 cycles a trip for a wave with its SIMD to itself -- were counted into the eight
    // operations of the trip: "8.0 ticks per fp64 FMA" was 5.4 + 2.6 of loop; tools/mb/xchg_cost.hip found it out)
    for (int it = 0; it < n; it += U) {
#pragma unroll
        for (int i = 0; i < 8 * U; ++i) {
            if (OP == 0) a[i & 7] = __builtin_amdgcn_rcp(a[i & 7]);
            if (OP == 1) a[i & 7] = (double)__builtin_amdgcn_rcpf((float)a[i & 7]);
            if (OP == 2) a[i & 7] = fma(a[i & 7], 1.0000001, 1e-9);
            if (OP == 3) a[i & 7] = __builtin_amdgcn_sqrt(a[i & 7]);
            if (OP == 4) { float f = (float)a[i & 7]; f = __builtin_amdgcn_rcpf(f); a[i & 7] = (double)f; }
            if (OP == 5) { float f = (float)a[i & 7]; f = __builtin_amdgcn_rcpf(f); double r = (double)f; r = fma(fma(-a[i & 7], r, 1.0), r, r); r = fma(fma(-a[i & 7], r, 1.0), r, r); a[i & 7] = r + 1.0; }
            if (OP == 6) { double r = __builtin_amdgcn_rcp(a[i & 7]); r = fma(fma(-a[i & 7], r, 1.0), r, r); a[i & 7] = r + 1.0; }
            if (OP == 7) a[i & 7] = a[i & 7] + 1.000001;
            if (OP == 8) { v2f p = __builtin_bit_cast(v2f, a[i & 7]); p = p * v2f{1.0000001f, 0.9999999f} + v2f{1e-9f, 1e-9f}; a[i & 7] = __builtin_bit_cast(double, p); }
            if (OP == 9) { float f = __builtin_bit_cast(v2f, a[i & 7]).x; f = fmaf(f, 1.0000001f, 1e-9f); v2f p{f, f}; a[i & 7] = __builtin_bit_cast(double, p); }
            if (OP == 11) { int lo = __builtin_bit_cast(int2, a[i & 7]).x, hi = __builtin_bit_cast(int2, a[i & 7]).y; asm volatile("v_mov_b32 %0, %1" : "=v"(lo) : "v"(hi)); a[i & 7] = __builtin_bit_cast(double, int2{hi, lo}); }
            if (OP == 12) { int lo = __builtin_bit_cast(int2, a[i & 7]).x, hi = __builtin_bit_cast(int2, a[i & 7]).y; asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(lo) : "v"(hi), "v"(lo)); a[i & 7] = __builtin_bit_cast(double, int2{hi, lo}); }
            if (OP == 13) { int lo = __builtin_bit_cast(int2, a[i & 7]).x, hi = __builtin_bit_cast(int2, a[i & 7]).y; asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(lo) : "v"(hi)); a[i & 7] = __builtin_bit_cast(double, int2{hi, lo}); }
            if (OP == 10) { float f; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(a[i & 7])); double d; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(f)); a[i & 7] = d; }  // (as instructions: the compiler folds (float)(double)f)
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
    const int n = 20480;
    constexpr int UN = 16;  // 128 operations per trip of the loop
    const char *names[] = {"v_rcp_f64", "cvt+v_rcp_f32+cvt", "v_fma_f64", "v_sqrt_f64", "cvt,rcp_f32,cvt", "f32 seed + 2 NR + add", "rcp_f64 + 1 NR + add", "v_add_f64", "v_pk_fma_f32", "v_fma_f32", "v_cvt_f32_f64 + v_cvt_f64_f32", "v_mov_b32", "v_cndmask_b32", "s_nop 1 + v_mov_b32_dpp"};
#define RUN(OP) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipLaunchKernelGGL((k<OP, UN>), dim3(1024), dim3(64), 0, 0, out, cyc, n); hipDeviceSynchronize(); hipEventRecord(e0, 0); hipLaunchKernelGGL((k<OP, UN>), dim3(1024 * W), dim3(64), 0, 0, out, cyc, n); hipEventRecord(e1, 0); hipDeviceSynchronize(); float ms = 0; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, 8192, hipMemcpyDeviceToHost); double m = 0; for (int i = 0; i < 1024; ++i) m += h[i]; printf("%-26s %7.2f s_memtime ticks per wave-op, %7.2f ns per wave-op per SIMD by the wall clock\n", names[OP], m / 1024 / n / 8, ms * 1e6 / n / 8 / W); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13)
    return 0;
}
