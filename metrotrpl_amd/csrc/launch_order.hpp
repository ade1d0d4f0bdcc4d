// launch_order.hpp -- issue order of the solves of one launch.
//
// A solve holds its SIMD slot (wave kernels) or its CU (team kernels) from start to finish and the durations of the
// solves of a batch spread 1 : 1.5; the hardware hands out workgroups in index order.  When a launch holds more
// solves than the device has slots -- or shares the device with other launches -- the expensive solves should go
// first.  The expected cost of a solve is a property of its descriptor: a linear model in the logarithms of the
// parameters that decide the step count (the tauP / tauN ratio that tightens the tolerance, lifetimes, mobilities,
// surface velocities, doping, fluence), fitted to the step counts of 5600 draws of the config-2 / config-3 priors
// (tools/cost_dump.py; R^2 0.75, rank correlation 0.86).  It only ever decides an ORDER: results, and where they
// are written, do not depend on it.  The solve kernels read their solve index from the permutation built here.
#pragma once
#include "device_math.hpp"

namespace mtrpl {

__device__ inline float cost_model(const mtrpl_solve_desc &d)
{
    auto lg = [](double x) { return (x > 0.0) ? (float)log10(x) : 0.0f; };
    const float ltn = lg(d.par[MTRPL_P_TAUN]), ltp = lg(d.par[MTRPL_P_TAUP]);
    const float lsf = lg(d.par[MTRPL_P_SF]) + 2.0f, lsb = lg(d.par[MTRPL_P_SB]) + 2.0f;           // cm/s
    const float lmn = lg(d.par[MTRPL_P_MU_N]) - 5.0f, lmp = lg(d.par[MTRPL_P_MU_P]) - 5.0f;       // cm^2/Vs
    const float lp0 = lg(d.par[MTRPL_P_P0]) + 21.0f;                                              // cm^-3
    const float lfl = (d.ini_mode == MTRPL_INI_FLUENCE) ? lg(d.ini[0]) : 12.5f;                   // cm^-2
    float c = -45.8f + 62.2f * ltn + 33.6f * ltp + 299.6f * fmaxf(ltp - ltn - 1.0f, 0.0f) + 19.2f * lsf - 9.3f * lsb +
              129.7f * lmn + 122.0f * lmp - 16.1f * lp0 + 86.0f * lfl + ((d.thickness < 1000.0) ? 12.1f : 0.0f) -
              52.5f * fminf(ltn, ltp) - 138.2f * fminf(lmn, lmp);
    c *= (float)d.nx * (1.0f / 128.0f);  // mixed grids in one launch: the wider solve is the longer one
    return (c == c) ? c : 0.0f;
}

// perm[0..B) = solve indices by descending expected cost.  ONE workgroup: bitonic sort of (key, index) pairs in LDS,
// n_pad = power of two >= B (<= kOrderMax).
constexpr int kOrderMax = 16384;
__global__ __launch_bounds__(1024) void order_kernel(const mtrpl_solve_desc *__restrict__ descs, int B, int n_pad,
                                                     int32_t *__restrict__ perm)
{
    extern __shared__ unsigned long long kv[];  // (sortable key << 32) | index
    for (int i = threadIdx.x; i < n_pad; i += blockDim.x) {
        unsigned key = 0u;  // padding sorts last
        if (i < B) {
            const unsigned u = __float_as_uint(cost_model(descs[i]));
            key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving map of float to unsigned
            key = key ? key : 1u;
        }
        kv[i] = ((unsigned long long)key << 32) | (unsigned)i;
    }
    __syncthreads();
    for (int k = 2; k <= n_pad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n_pad; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = kv[i], b = kv[l];
                    const bool desc = (i & k) == 0;  // descending blocks first: the final order is descending
                    if (desc ? (a < b) : (a > b)) { kv[i] = b; kv[l] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < B; i += blockDim.x) perm[i] = (int32_t)(kv[i] & 0xffffffffu);
}

}  // namespace mtrpl
