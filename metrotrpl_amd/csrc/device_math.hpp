// device_math.hpp -- wave-level primitives and small fixed-size algebra shared by every solver
// backend (gfx950 only: wave64, DPP wave shifts, ds_bpermute, readlane scalarisation).
//
// The scalar type is a template parameter (`T` = double for the product kernels; float only for the
// precision sweep of BASELINE.json configs[4], instantiated through the team backend).  Memory that is
// shared with the host -- descriptors, sample times, output samples -- is always fp64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/metrotrpl_amd.h"

namespace mtrpl {

// forward_solver.py:21-24
static constexpr double kEps0 = 8.854 * 1e-12 * 1e-9;
static constexpr double kQC = 1.602e-19;
static constexpr double kKB = 8.61773e-5;

template <class T> struct Vec2;
template <> struct Vec2<double> { using type = double2; };
template <> struct Vec2<float> { using type = float2; };
template <class T> using vec2_t = typename Vec2<T>::type;

// "infinity" that survives a conversion to the working precision
template <class T> __device__ constexpr T big() { return sizeof(T) == 8 ? (T)1e300 : (T)1e30; }

// Pins a value to the point of the program where this stands: the optimiser may neither hoist the
// computations that consume the result above it nor sink its producers below it.  No instruction.
// (Used at the top of the factorisation: it sits behind `if (!have_fac)` inside the Newton loop, its
// inputs are invariant in that loop, and LICM otherwise evaluates half of the Jacobian -- four fp64
// reciprocals and a hundred multiplies -- speculatively on EVERY step instead of every eleventh.)
template <class T> __device__ inline T pinned(T x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// ---- lane exchange -------------------------------------------------------------------------------
template <class T> __device__ inline T shf(T v, int src) { return __shfl(v, src, 64); }

// value is identical in all lanes: move it to scalar registers
template <class T> __device__ inline T uniform(T v)
{
    union { T d; int i[sizeof(T) / 4]; } u;
    u.d = v;
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) u.i[k] = __builtin_amdgcn_readfirstlane(u.i[k]);
    return u.d;
}
__device__ inline int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// value of lane `src` as a wave-uniform scalar
template <class T> __device__ inline T lane_value(T v, int src)
{
    union { T d; int i[sizeof(T) / 4]; } u;
    u.d = v;
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) u.i[k] = __builtin_amdgcn_readlane(u.i[k], src);
    return u.d;
}

// DPP move with a partial row / bank mask: masked lanes (and lanes whose source is out of range) get 0.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, class T>
__device__ inline T dpp0(T v)
{
    union { T d; int i[sizeof(T) / 4]; } u, r;
    u.d = v;
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 4; ++k)
        r.i[k] = __builtin_amdgcn_update_dpp(0, u.i[k], CTRL, ROW_MASK, BANK_MASK, false);
    return r.d;
}
// DPP move over all rows and banks with bound_ctrl: lanes whose source is out of range get 0 and no
// lane keeps an old value, so the destination needs no initialisation (one v_mov_b32_dpp per dword).
template <int CTRL, class T>
__device__ inline T dppz(T v)
{
    union { T d; int i[sizeof(T) / 4]; } u, r;
    u.d = v;
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) r.i[k] = __builtin_amdgcn_mov_dpp(u.i[k], CTRL, 0xf, 0xf, true);
    return r.d;
}

// Nearest-neighbour exchange across the whole wave as a DPP wave shift (2 VALU moves per double
// instead of a ds_bpermute round trip).  The lane with no neighbour gets 0.
template <class T> __device__ inline T from_prev(T v) { return dppz<0x138>(v); }  // wave_shr:1: lane i <- lane i-1
template <class T> __device__ inline T from_next(T v) { return dppz<0x130>(v); }  // wave_shl:1: lane i <- lane i+1

// Stride-32 exchange without LDS: one v_permlane32_swap per dword (gfx950) yields BOTH neighbours --
// `dn` holds the value of lane-32 in lanes 32..63, `up` the value of lane+32 in lanes 0..31; the other
// halves keep the lane's own value (finite: where that is read it meets a zero block).
template <class T> __device__ inline void swap32(T v, T &dn, T &up)
{
    union { T d; unsigned i[sizeof(T) / 4]; } u, a, b;
    u.d = v;
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) {
        const auto r = __builtin_amdgcn_permlane32_swap(u.i[k], u.i[k], false, false);
        a.i[k] = r[0];
        b.i[k] = r[1];
    }
    dn = a.d;
    up = b.d;
}

// Sum over the 64 lanes, result wave-uniform (in SGPRs).  gfx9 DPP ladder: row_shr 1,2,4,8 inside
// each row of 16 (sources outside the row read as 0), then row_bcast15 / row_bcast31 carry the row
// totals forward; the grand total lands in lane 63.
template <class T> __device__ inline T wave_sum(T v)
{
    v += dppz<0x111>(v);        // row_shr:1
    v += dppz<0x112>(v);        // row_shr:2
    v += dppz<0x114>(v);        // row_shr:4
    v += dppz<0x118>(v);        // row_shr:8
    v += dpp0<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
    v += dpp0<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
    return lane_value(v, 63);
}
// The same in f32 for quantities that only steer the controller (error norms): the DPP move folds into
// the add (v_add_f32_dpp), a third of the instructions of the f64 ladder.
__device__ inline double wave_sum_ctl(double v) { return (double)wave_sum((float)v); }
__device__ inline float wave_sum_ctl(float v) { return wave_sum(v); }
template <class T> __device__ inline T wave_max(T v)
{
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v = fmax(v, __shfl_xor(v, s, 64));
    return v;
}
// inclusive prefix sum over the lanes of a wave
template <class T> __device__ inline T wave_scan_incl(T v, int lane)
{
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const T t = shf(v, (lane - s) & 63);
        v += (lane >= s) ? t : (T)0;
    }
    return v;
}

// ---- controller-grade transcendental shortcuts ---------------------------------------------------
// Norms, contraction rates and step factors only steer the controller: f32 hardware
// sqrt / rcp / log2 / exp2 (1 ulp of f32) replace the fp64 sequences on the scalar critical path.
template <class T> __device__ inline T norm_sqrt(T x) { return (T)__builtin_amdgcn_sqrtf((float)x); }
template <class T> __device__ inline T ctl_rcp(T x) { return (T)__builtin_amdgcn_rcpf((float)x); }
// x^e (e = -1/k as a compile-time constant)
template <class T> __device__ inline T pow_neg_inv(T x, float e)
{
    const float xf = (float)fmin(fmax(x, (T)1e-30), (T)1e30);
    return (T)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(xf) * e);
}
// x^(-1/k)
template <class T> __device__ inline T inv_root(T x, int k)
{
    const float xf = (float)fmin(fmax(x, (T)1e-30), (T)1e30);
    return (T)__builtin_amdgcn_exp2f(-__builtin_amdgcn_logf(xf) / (float)k);
}

// 1/x by the hardware seed + two Newton steps (~1 ulp; no div_scale/fixup ladder: operands here are
// always normal, finite and non-zero)
__device__ inline double frcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ inline float frcp(float x)
{
    const float q = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, q, 1.0f), q, q);
}

// ---- 2x2 blocks ----------------------------------------------------------------------------------
template <class T> struct M2T { T a, b, c, d; };
template <class T> struct V2T { T x, y; };
using M2 = M2T<double>;
using V2 = V2T<double>;

template <class T> __device__ inline M2T<T> mzero() { return M2T<T>{(T)0, (T)0, (T)0, (T)0}; }
template <class T> __device__ inline M2T<T> mident() { return M2T<T>{(T)1, (T)0, (T)0, (T)1}; }
template <class T> __device__ inline M2T<T> mmul(const M2T<T> &x, const M2T<T> &y)
{
    return M2T<T>{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}
template <class T> __device__ inline M2T<T> msub(const M2T<T> &x, const M2T<T> &y) { return M2T<T>{x.a - y.a, x.b - y.b, x.c - y.c, x.d - y.d}; }
template <class T> __device__ inline M2T<T> mneg(const M2T<T> &x) { return M2T<T>{-x.a, -x.b, -x.c, -x.d}; }
template <class T> __device__ inline M2T<T> minv(const M2T<T> &x)
{
    const T r = frcp(x.a * x.d - x.b * x.c);
    return M2T<T>{x.d * r, -x.b * r, -x.c * r, x.a * r};
}
// member-wise selects (a `c ? structA : structB` can end up as a select of two stack addresses)
template <class T> __device__ inline M2T<T> msel(bool c, const M2T<T> &x, const M2T<T> &y) { return M2T<T>{c ? x.a : y.a, c ? x.b : y.b, c ? x.c : y.c, c ? x.d : y.d}; }
template <class T> __device__ inline V2T<T> vsel(bool c, const V2T<T> &x, const V2T<T> &y) { return V2T<T>{c ? x.x : y.x, c ? x.y : y.y}; }
template <class T> __device__ inline V2T<T> mvec(const M2T<T> &m, const V2T<T> &v) { return V2T<T>{m.a * v.x + m.b * v.y, m.c * v.x + m.d * v.y}; }
// Shuffles wrap around the wave.  No masking is needed where these are used: every wrapped value is a
// finite number that only ever meets an exactly-zero coupling block (A of the first rows / C of the
// last rows vanish, and stay zero through every reduction level), so it contributes exactly 0.
template <class T> __device__ inline M2T<T> mshf(const M2T<T> &m, int src) { return M2T<T>{shf(m.a, src), shf(m.b, src), shf(m.c, src), shf(m.d, src)}; }
template <class T> __device__ inline V2T<T> vshf(const V2T<T> &v, int src) { return V2T<T>{shf(v.x, src), shf(v.y, src)}; }
template <class T> __device__ inline void mswap32(const M2T<T> &m, M2T<T> &dn, M2T<T> &up)
{
    swap32(m.a, dn.a, up.a); swap32(m.b, dn.b, up.b); swap32(m.c, dn.c, up.c); swap32(m.d, dn.d, up.d);
}
template <class T> __device__ inline void vswap32(const V2T<T> &v, V2T<T> &dn, V2T<T> &up)
{
    swap32(v.x, dn.x, up.x); swap32(v.y, dn.y, up.y);
}
template <class T> __device__ inline M2T<T> mnext(const M2T<T> &m) { return M2T<T>{from_next(m.a), from_next(m.b), from_next(m.c), from_next(m.d)}; }
template <class T> __device__ inline M2T<T> mprev(const M2T<T> &m) { return M2T<T>{from_prev(m.a), from_prev(m.b), from_prev(m.c), from_prev(m.d)}; }
template <class T> __device__ inline V2T<T> vnext(const V2T<T> &v) { return V2T<T>{from_next(v.x), from_next(v.y)}; }
template <class T> __device__ inline V2T<T> vprev(const V2T<T> &v) { return V2T<T>{from_prev(v.x), from_prev(v.y)}; }

}  // namespace mtrpl
