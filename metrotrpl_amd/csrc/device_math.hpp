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
// ... and when the compiler can see for itself that v is uniform it drops the readfirstlane and leaves v in the vector register
// its arithmetic produced it in (floating point is on the vector unit only).  The EMPTY asm with a scalar-register
// constraint is what keeps the value in a scalar register: whatever the compiler makes of the builtin, the operand of
// the asm has to be in an SGPR, and the only way there from a VGPR is a v_readfirstlane the COMPILER emits -- so the
// hazard recogniser sees the producing instruction and owns its wait states (round 5 had the v_readfirstlane inside
// the asm with hand-counted s_nops, valid only for the consumers of that day: ADVICE r5).
__device__ inline float to_scalar(float v)
{
    int r = __builtin_amdgcn_readfirstlane(__float_as_int(v));
    asm volatile("" : "+s"(r));
    return __int_as_float(r);
}

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
// Several fp32 sums over the 64 lanes in ONE DPP ladder (the sums only steer the controller).  The first
// level(s) add neighbours inside a quad and keep a different sum per lane of the quad (lane & 1 selects between
// two sums, lane & 3 between four), the row_shr levels then work on stride-2 / stride-4 lanes, and the rows are
// combined with the two permlane swaps of gfx950: about half the instructions of one ladder per sum, and no
// serial chain of six dependent DPP adds per sum.  Results are wave-uniform (SGPRs).
template <int CTRL> __device__ inline float add_dpp(float v)  // v + dpp(v), sources out of range read as 0
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ inline float rows_total(float v)  // every lane <- sum over the 4 rows of the same lane-in-row
{
    // (the two results are copied to scalars before the bit cast: __builtin_bit_cast applied to an element of the
    // returned vector reads element 0 for both with this compiler)
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // rows (0,0,2,2) | (1,1,3,3)
    const unsigned a0 = a[0], a1 = a[1];
    v = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
    const unsigned w = __builtin_bit_cast(unsigned, v);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);  // halves (lo,lo) | (hi,hi)
    const unsigned b0 = b[0], b1 = b[1];
    return __builtin_bit_cast(float, b0) + __builtin_bit_cast(float, b1);
}
template <int NS> __device__ inline void wave_sums_f(float (&s)[NS], int lane)
{
    static_assert(NS == 1 || NS == 2 || NS == 4, "1, 2 or 4 sums");
    if constexpr (NS == 1) {
        float v = s[0];
        v = add_dpp<0x111>(v); v = add_dpp<0x112>(v); v = add_dpp<0x114>(v); v = add_dpp<0x118>(v);  // row_shr 1,2,4,8
        s[0] = lane_value(rows_total(v), 15);
    } else if constexpr (NS == 2) {
        const float a = add_dpp<0xb1>(s[0]), b = add_dpp<0xb1>(s[1]);  // quad_perm [1,0,3,2]: pair sums
        float v = (lane & 1) ? b : a;
        v = add_dpp<0x112>(v); v = add_dpp<0x114>(v); v = add_dpp<0x118>(v);  // lanes 14 / 15 of a row: its two totals
        v = rows_total(v);
        s[0] = lane_value(v, 14);
        s[1] = lane_value(v, 15);
    } else {
        const float a = add_dpp<0xb1>(s[0]), b = add_dpp<0xb1>(s[1]), c = add_dpp<0xb1>(s[2]), d = add_dpp<0xb1>(s[3]);
        float ab = (lane & 1) ? b : a, cd = (lane & 1) ? d : c;
        ab = add_dpp<0x4e>(ab);  // quad_perm [2,3,0,1]: quad sums, lane & 1 still selects the sum
        cd = add_dpp<0x4e>(cd);
        float v = (lane & 2) ? cd : ab;
        v = add_dpp<0x114>(v); v = add_dpp<0x118>(v);  // lanes 12..15 of a row: its four totals
        v = rows_total(v);
        s[0] = lane_value(v, 12);
        s[1] = lane_value(v, 13);
        s[2] = lane_value(v, 14);
        s[3] = lane_value(v, 15);
    }
}
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
// Norms, contraction rates and step factors only steer the controller: they are fp32 values handled with the
// hardware sqrt / rcp / log2 / exp2 (1 ulp of fp32).
// x^e for x > 0
__device__ inline float pow_ctl(float x, float e)
{
    const float xf = fminf(fmaxf(x, 1e-30f), 1e30f);
    return __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(xf) * e);
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
// ... and with one Newton step: the seed of v_rcp_f64 carries about 2^-26, so one step leaves ~1e-15 (measured on
// gfx950: tests/test_gpu_solver.py::test_rhs_matches_reference).  For the RIGHT-HAND SIDE, evaluated 1300 times per
// solve; the Jacobian keeps frcp.
__device__ inline double frcp1(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ inline float frcp(float x)
{
    const float q = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, q, 1.0f), q, q);
}
__device__ inline float frcp1(float x) { return frcp(x); }

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
