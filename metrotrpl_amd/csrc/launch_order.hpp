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

// perm[0..B) = solve indices by descending expected cost, in two passes of independent ONE-WAVE workgroups: the keys,
// then the rank of every key among all of them (ties broken by index: a permutation).  One-wave workgroups on
// purpose: with launches of the two-solves-per-SIMD kernel in flight every register file of the device is full, a
// new wave can start only where a solve has just finished, and a 1024-thread sorting workgroup would wait for
// sixteen such slots on ONE compute unit (measured: 0.44 ms on average in front of every launch).
constexpr int kOrderMax = 16384;
__global__ __launch_bounds__(64) void cost_key_kernel(const mtrpl_solve_desc *__restrict__ descs, int B, float *__restrict__ keys)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < B) keys[i] = cost_model(descs[i]);
}
__global__ __launch_bounds__(64) void cost_rank_kernel(const float *__restrict__ keys, int B, int32_t *__restrict__ perm)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const float mine = (i < B) ? keys[i] : 0.0f;
    int rank = 0;
    for (int j = 0; j < B; ++j) {  // keys[j]: a wave-uniform (scalar) load
        const float kj = keys[j];
        rank += (kj > mine || (kj == mine && j < i)) ? 1 : 0;
    }
    if (i < B) perm[rank] = i;
}

}  // namespace mtrpl
