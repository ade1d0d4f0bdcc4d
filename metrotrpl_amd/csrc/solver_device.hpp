// solver_device.hpp -- device code of the MI355X (gfx950) forward solver.
//
// One 64-lane wavefront integrates one forward simulation (forward_solver.py:41-203):
// lane l owns R consecutive grid nodes (N_i, P_i) and the edge field E_i on the
// LEFT face of each of them (i = l*R + r), so nx <= 64*R.  The whole solver state
// lives on the CU (backward-difference history and Newton vectors in this wave's
// VGPRs/AGPRs, Newton matrix factors in its LDS slab); HBM is touched for the
// per-solve descriptor, the initial condition, the output sample times and the nt
// output samples only.
//
// Time integration: variable-order (1..5), quasi-constant-step NDF
// (Shampine & Reichelt, "The MATLAB ODE Suite", SIAM J. Sci. Comput. 18, 1997;
// Byrne & Hindmarsh 1975) -- the same method class as the reference's ODEPACK
// LSODA in its stiff (BDF) mode (forward_solver.py:163-165), but with
//   * an ANALYTIC Jacobian instead of 3nx+1 finite-difference columns,
//   * the structure of that Jacobian used exactly: each dE_i row couples only to
//     its own E_i, so E is eliminated per edge and the Newton matrix I - c*J
//     becomes block tridiagonal with 2x2 blocks in (dN_i, dP_i);
//   * that system solved across the wave by one in-lane cyclic-reduction level
//     followed by 6 parallel-cyclic-reduction levels over the 64 lanes
//     (nearest-neighbour exchange by DPP wave shifts, longer strides by
//     ds_bpermute; one wave per workgroup, so no barriers); all 17 elimination
//     blocks per lane are kept in this wave's 34 KB LDS slab as 16-byte rows, so
//     each further Newton iteration is a cheap re-solve.
// (no #pragma once: mtrpl_fp32_pass.hpp compiles this file a second time with `double` read as `float`)
#if (!defined(MTRPL_FP32_PASS) && !defined(MTRPL_SOLVER_DEVICE_HPP_F64)) || (defined(MTRPL_FP32_PASS) && !defined(MTRPL_SOLVER_DEVICE_HPP_F32))
#ifdef MTRPL_FP32_PASS
#define MTRPL_SOLVER_DEVICE_HPP_F32
#else
#define MTRPL_SOLVER_DEVICE_HPP_F64
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/metrotrpl_amd.h"

#ifndef MTRPL_FP32_PASS
typedef double f64;  // memory that is fp64 in BOTH passes: descriptors, sample times, output samples
#endif

namespace mtrpl {

// forward_solver.py:21-24
static constexpr double kEps0 = 8.854 * 1e-12 * 1e-9;
static constexpr double kQC = 1.602e-19;
static constexpr double kKB = 8.61773e-5;

struct Par {
    double n0, p0, mun, mup, ks, Cn, Cp, Sf, Sb, tn, tp, Lam, Tm, kC, Nt, tE;
    double dx, idx, DnT, DpT, n0p0;  // idx = 1/dx
};

__device__ inline Par make_par(const f64 *p, double dx)
{
    Par q;
    q.n0 = p[MTRPL_P_N0]; q.p0 = p[MTRPL_P_P0]; q.mun = p[MTRPL_P_MU_N]; q.mup = p[MTRPL_P_MU_P];
    q.ks = p[MTRPL_P_KS]; q.Cn = p[MTRPL_P_CN]; q.Cp = p[MTRPL_P_CP]; q.Sf = p[MTRPL_P_SF];
    q.Sb = p[MTRPL_P_SB]; q.tn = p[MTRPL_P_TAUN]; q.tp = p[MTRPL_P_TAUP];
    q.Lam = kQC / (p[MTRPL_P_EPS] * kEps0);  // forward_solver.py:131
    q.Tm = p[MTRPL_P_TM]; q.kC = p[MTRPL_P_KC]; q.Nt = p[MTRPL_P_NT]; q.tE = p[MTRPL_P_TAUE];
    q.dx = dx;
    q.idx = 1.0 / dx;
    q.DnT = q.mun * kKB * q.Tm;  // mu_n*kB*Tm, left to right as forward_solver.py:357
    q.DpT = q.mup * kKB * q.Tm;
    q.n0p0 = q.n0 * q.p0;
    return q;
}

// ---- wave helpers ----------------------------------------------------------
__device__ inline double shf(double v, int src) { return __shfl(v, src, 64); }

__device__ inline double uniform(double v)
{
    // value is identical in all lanes: move it to scalar registers
    union { double d; int i[sizeof(double) / 4]; } u;
    u.d = v;
#pragma unroll
    for (unsigned k = 0; k < sizeof(double) / 4; ++k) u.i[k] = __builtin_amdgcn_readfirstlane(u.i[k]);
    return u.d;
}

// DPP move of a double: lanes whose source is out of range or masked get 0.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ inline double dpp0(double v)
{
    union { double d; int i[sizeof(double) / 4]; } u, r;
    u.d = v;
#pragma unroll
    for (unsigned k = 0; k < sizeof(double) / 4; ++k)
        r.i[k] = __builtin_amdgcn_update_dpp(0, u.i[k], CTRL, ROW_MASK, BANK_MASK, false);
    return r.d;
}

// Nearest-neighbour exchange across the whole wave as a DPP wave shift (2 VALU moves
// per double instead of a ds_bpermute round trip).  The lane with no neighbour gets
// 0.0, which -- like the wrapped value of a bpermute -- only ever meets a zero block.
__device__ inline double from_prev(double v) { return dpp0<0x138>(v); }  // wave_shr:1: lane i <- lane i-1
__device__ inline double from_next(double v) { return dpp0<0x130>(v); }  // wave_shl:1: lane i <- lane i+1

// Sum over the 64 lanes, result wave-uniform (in SGPRs).  gfx9 DPP ladder:
// row_shr 1,2,4,8 inside each row of 16, then row_bcast15 / row_bcast31 carry the
// row totals forward; the grand total lands in lane 63.
__device__ inline double wave_sum(double v)
{
    v += dpp0<0x111>(v);             // row_shr:1
    v += dpp0<0x112>(v);             // row_shr:2
    v += dpp0<0x114, 0xf, 0xe>(v);   // row_shr:4, banks 1-3
    v += dpp0<0x118, 0xf, 0xc>(v);   // row_shr:8, banks 2-3
    v += dpp0<0x142, 0xa>(v);        // row_bcast:15 -> rows 1,3
    v += dpp0<0x143, 0xc>(v);        // row_bcast:31 -> rows 2,3
    union { double d; int i[sizeof(double) / 4]; } u;
    u.d = v;
#pragma unroll
    for (unsigned k = 0; k < sizeof(double) / 4; ++k) u.i[k] = __builtin_amdgcn_readlane(u.i[k], 63);
    return u.d;
}

// x^(-1/k) for step-size factors: 1e-6 relative accuracy is plenty, so use the
// f32 hardware log2/exp2 instead of a software fp64 pow.
__device__ inline double inv_root(double x, int k)
{
    const float xf = (float)fmin(fmax(x, 1e-30), 1e30);
    return (double)__builtin_amdgcn_exp2f(-__builtin_amdgcn_logf(xf) / (float)k);
}

// 1/x by the hardware seed + two Newton steps (~1 ulp; no div_scale/fixup ladder,
// operands here are always normal, finite and non-zero)
__device__ inline double frcp(double x)
{
#ifdef MTRPL_FP32_PASS
    float q = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, q, 1.0f), q, q);
#endif
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// ---- 2x2 blocks -------------------------------------------------------------
struct M2 { double a, b, c, d; };
struct V2 { double x, y; };

__device__ inline M2 mzero() { return M2{0.0, 0.0, 0.0, 0.0}; }
__device__ inline M2 mmul(const M2 &x, const M2 &y)
{
    return M2{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}
__device__ inline M2 msub(const M2 &x, const M2 &y) { return M2{x.a - y.a, x.b - y.b, x.c - y.c, x.d - y.d}; }
__device__ inline M2 mneg(const M2 &x) { return M2{-x.a, -x.b, -x.c, -x.d}; }
__device__ inline M2 minv(const M2 &x)
{
    const double r = frcp(x.a * x.d - x.b * x.c);
    return M2{x.d * r, -x.b * r, -x.c * r, x.a * r};
}
__device__ inline V2 mvec(const M2 &m, const V2 &v) { return V2{m.a * v.x + m.b * v.y, m.c * v.x + m.d * v.y}; }
// Shuffles wrap around the wave.  No masking is needed anywhere below: every
// wrapped value is a finite number that only ever meets an exactly-zero
// coupling block (A of the first rows / C of the last rows vanish, and stay
// zero through every reduction level), so it contributes exactly 0.
__device__ inline M2 mshf(const M2 &m, int src) { return M2{shf(m.a, src), shf(m.b, src), shf(m.c, src), shf(m.d, src)}; }
__device__ inline V2 vshf(const V2 &v, int src) { return V2{shf(v.x, src), shf(v.y, src)}; }
__device__ inline M2 mnext(const M2 &m) { return M2{from_next(m.a), from_next(m.b), from_next(m.c), from_next(m.d)}; }
__device__ inline V2 vnext(const V2 &v) { return V2{from_next(v.x), from_next(v.y)}; }
__device__ inline V2 vprev(const V2 &v) { return V2{from_prev(v.x), from_prev(v.y)}; }

// ---- RHS: forward_solver.py:332-372 (dydt_numba) ------------------------------
// y[r], y[R+r], y[2R+r] = N_i, P_i, E_i (E on the left face of node i), i = lane*R+r.
// Padded nodes (i >= L) carry N=P=1, E=0 and get f=0.
template <int R>
__device__ inline void rhs_std(const Par &p, const double (&y)[3 * R], double (&f)[3 * R], int lane, int L)
{
    double Jn[R + 1], Jp[R + 1];
    const double Nl = from_prev(y[R - 1]);
    const double Pl = from_prev(y[2 * R - 1]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane * R + r;
        const double N = y[r], P = y[R + r], E = y[2 * R + r];
        const double Nm = (r == 0) ? Nl : y[(r == 0) ? 0 : r - 1];
        const double Pm = (r == 0) ? Pl : y[R + ((r == 0) ? 0 : r - 1)];
        // :356-359 (x/dx evaluated as x*(1/dx): 1 ulp from the reference's division)
        double jn = p.mun * ((Nm + N) / 2 * E) + p.DnT * ((N - Nm) * p.idx);
        double jp = p.mup * ((Pm + P) / 2 * E) - p.DpT * ((P - Pm) * p.idx);
        if (i == 0) {  // :346-351
            const double Sft = p.Sf * (N * P - p.n0p0) * frcp(N + P);
            jn = Sft;
            jp = -Sft;
        }
        if (i >= L) { jn = 0.0; jp = 0.0; }
        Jn[r] = jn;
        Jp[r] = jp;
    }
    Jn[R] = from_next(Jn[0]);
    Jp[R] = from_next(Jp[0]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane * R + r;
        const double N = y[r], P = y[R + r];
        const double NP = N * P - p.n0p0;
        double jnr = Jn[r + 1], jpr = Jp[r + 1];
        if (i == L - 1) {  // :347,:350,:352
            const double Sbt = p.Sb * NP * frcp(N + P);
            jnr = -Sbt;
            jpr = Sbt;
        }
        const double recomb = ((p.Cn * N + p.Cp * P) + p.ks + frcp((p.tn * P) + (p.tp * N))) * NP;  // :366
        const bool real = i < L;
        f[r] = real ? ((jnr - Jn[r]) * p.idx - recomb) : 0.0;       // :369
        f[R + r] = real ? (-(jpr - Jp[r]) * p.idx - recomb) : 0.0;  // :370
        f[2 * R + r] = (real && i > 0) ? (-(Jn[r] + Jp[r]) * p.Lam) : 0.0;  // :363 (edge 0: exactly 0)
    }
}

// ---- Newton matrix factors ---------------------------------------------------
// M = I - c*J(yJ).  See DESIGN.md "Newton linear algebra" for the derivation.
template <int R>
struct Factors;

template <>
struct Factors<2> {
    double w[2][2];    // dJ/dbE weights per own edge
    double eco[2][4];  // dE_i = bE_i*minv - eco.(dN_{i-1}, dN_i, dP_{i-1}, dP_i)
    double minv[2];
    double cd;         // c/dx
    M2 Bfin;
    // The 17 elimination blocks (in-lane CR: Binv_e, BiA_e, BiC_e, cr_al, cr_ga; PCR:
    // al[6], ga[6]) live in this wave's LDS slab: [17 blocks][2 halves][64 lanes] x 16 B,
    // i.e. every access is a conflict-free ds_read/write_b128 (lane stride 16 B).
    double2 *lds;
    enum { BINV_E = 0, BIA_E, BIC_E, CR_AL, CR_GA, PAL0, PGA0 = PAL0 + 6, NBLK = PGA0 + 6 };
    __device__ inline void put(int blk, int lane, const M2 &m) const
    {
        lds[(blk * 2 + 0) * 64 + lane] = double2{m.a, m.b};
        lds[(blk * 2 + 1) * 64 + lane] = double2{m.c, m.d};
    }
    __device__ inline M2 get(int blk, int lane) const
    {
        // Opaque lane index: keeps the block loads inside the Newton loop (16-B rows re-read
        // from LDS every iteration) instead of letting the compiler hoist 64+ VGPRs of
        // loop-invariant blocks and push the difference array out to AGPRs
        // (measured: AGPRs 256 -> 116, 12.8 -> 12.3 ms per 1024-solve batch).
        asm volatile("" : "+v"(lane));
        const double2 u = lds[(blk * 2 + 0) * 64 + lane], v = lds[(blk * 2 + 1) * 64 + lane];
        return M2{u.x, u.y, v.x, v.y};
    }
};
constexpr int kFactorLdsBytes = 17 * 2 * 64 * 16;  // 34816 B per wave

template <int R>
__device__ inline void edge_blocks(const Par &p, const double (&y)[3 * R], double c, int lane, int L,
                                   M2 (&Lm)[R], M2 (&Rm)[R], Factors<R> &F)
{
    const double cL = c * p.Lam;
    const double dnx = p.DnT * p.idx, dpx = p.DpT * p.idx;
    const double Nl = from_prev(y[R - 1]);
    const double Pl = from_prev(y[2 * R - 1]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane * R + r;
        const double N = y[r], P = y[R + r], E = y[2 * R + r];
        const double Nm = (r == 0) ? Nl : y[(r == 0) ? 0 : r - 1];
        const double Pm = (r == 0) ? Pl : y[R + ((r == 0) ? 0 : r - 1)];
        const double an = 0.5 * p.mun * E, ap = 0.5 * p.mup * E;
        const double jnm = an - dnx, jnp = an + dnx, jpm = ap + dpx, jpp = ap - dpx;
        const double jne = p.mun * 0.5 * (Nm + N), jpe = p.mup * 0.5 * (Pm + P);
        const double mi = frcp(1.0 + cL * (jne + jpe));
        const double g = jne * cL * mi, gp = jpe * cL * mi;
        M2 lm{(1.0 - g) * jnm, -g * jpm, -gp * jnm, (1.0 - gp) * jpm};
        M2 rm{(1.0 - g) * jnp, -g * jpp, -gp * jnp, (1.0 - gp) * jpp};
        double w0 = jne * mi, w1 = jpe * mi, mv = mi;
        double e0 = cL * mi * jnm, e1 = cL * mi * jnp, e2 = cL * mi * jpm, e3 = cL * mi * jpp;
        if (i == 0) {  // front surface: J_0 = (S0, -S0), no field
            const double NP = N * P - p.n0p0, s = N + P, is2 = frcp(s * s);
            const double s0n = p.Sf * (P * s - NP) * is2, s0p = p.Sf * (N * s - NP) * is2;
            lm = mzero();
            rm = M2{s0n, s0p, -s0n, -s0p};
            w0 = w1 = e0 = e1 = e2 = e3 = 0.0;
            mv = 1.0;
        }
        if (i >= L) {
            lm = mzero();
            rm = mzero();
            w0 = w1 = e0 = e1 = e2 = e3 = 0.0;
            mv = 1.0;
        }
        Lm[r] = lm;
        Rm[r] = rm;
        F.w[r][0] = w0; F.w[r][1] = w1;
        F.eco[r][0] = e0; F.eco[r][1] = e1; F.eco[r][2] = e2; F.eco[r][3] = e3;
        F.minv[r] = mv;
    }
}

template <int R>
__device__ inline void node_blocks(const Par &p, const double (&y)[3 * R], double c, int lane, int L,
                                   const M2 (&Lm)[R], const M2 (&Rm)[R], M2 (&A)[R], M2 (&B)[R], M2 (&C)[R])
{
    const double cd = c * p.idx;
    // lane 63 wraps to lane 0's edge 0: harmless, its last node is the back surface or padding
    const M2 LmNext = mnext(Lm[0]);
    const M2 RmNext = mnext(Rm[0]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane * R + r;
        const double N = y[r], P = y[R + r];
        M2 lmn = (r == R - 1) ? LmNext : Lm[(r == R - 1) ? r : r + 1];
        M2 rmn = (r == R - 1) ? RmNext : Rm[(r == R - 1) ? r : r + 1];
        const double NP = N * P - p.n0p0;
        if (i == L - 1) {  // back surface: J_L = (-SL, SL)
            const double s = N + P, is2 = frcp(s * s);
            const double sLn = p.Sb * (P * s - NP) * is2, sLp = p.Sb * (N * s - NP) * is2;
            lmn = M2{-sLn, -sLp, sLn, sLp};
            rmn = mzero();
        }
        const double den = p.tn * P + p.tp * N, iden = frcp(den);
        const double kk = (p.Cn * N + p.Cp * P) + p.ks + iden;
        const double RN = (p.Cn - p.tp * iden * iden) * NP + kk * P;
        const double RP = (p.Cp - p.tn * iden * iden) * NP + kk * N;
        M2 a{cd * Lm[r].a, cd * Lm[r].b, -cd * Lm[r].c, -cd * Lm[r].d};
        M2 cc{-cd * rmn.a, -cd * rmn.b, cd * rmn.c, cd * rmn.d};
        M2 b{1.0 + cd * (Rm[r].a - lmn.a) + c * RN, cd * (Rm[r].b - lmn.b) + c * RP,
             -cd * (Rm[r].c - lmn.c) + c * RN, 1.0 - cd * (Rm[r].d - lmn.d) + c * RP};
        if (i >= L) {
            a = mzero();
            cc = mzero();
            b = M2{1.0, 0.0, 0.0, 1.0};
        }
        A[r] = a;
        B[r] = b;
        C[r] = cc;
    }
}

__device__ inline void build_factors(const Par &p, const double (&yJ)[6], double c, int lane, int L, Factors<2> &F)
{
    M2 Lm[2], Rm[2], A[2], B[2], C[2];
    edge_blocks<2>(p, yJ, c, lane, L, Lm, Rm, F);
    node_blocks<2>(p, yJ, c, lane, L, Lm, Rm, A, B, C);
    F.cd = c * p.idx;
    // in-lane cyclic reduction: row e = 2*lane (r=0) eliminated, row k = 2*lane+1 (r=1) kept
    const M2 Binv_e = minv(B[0]);
    F.put(Factors<2>::BINV_E, lane, Binv_e);
    F.put(Factors<2>::BIA_E, lane, mmul(Binv_e, A[0]));
    F.put(Factors<2>::BIC_E, lane, mmul(Binv_e, C[0]));
    const int nx = (lane + 1) & 63;
    const M2 Binv_up = mnext(Binv_e);
    const M2 A_up = mnext(A[0]);
    const M2 C_up = mnext(C[0]);
    const M2 cr_al = mmul(A[1], Binv_e);
    const M2 cr_ga = mmul(C[1], Binv_up);
    F.put(Factors<2>::CR_AL, lane, cr_al);
    F.put(Factors<2>::CR_GA, lane, cr_ga);
    M2 Ak = mneg(mmul(cr_al, A[0]));
    M2 Ck = mneg(mmul(cr_ga, C_up));
    M2 Bk = msub(msub(B[1], mmul(cr_al, C[0])), mmul(cr_ga, A_up));
    // parallel cyclic reduction over the 64 kept rows
#pragma unroll
    for (int lv = 0; lv < 6; ++lv) {
        const int s = 1 << lv;
        const int lm_ = (lane - s) & 63, lp_ = (lane + s) & 63;
        const M2 Bi = minv(Bk);
        const M2 Bim = mshf(Bi, lm_), Bip = mshf(Bi, lp_);
        const M2 Akm = mshf(Ak, lm_), Akp = mshf(Ak, lp_);
        const M2 Ckm = mshf(Ck, lm_), Ckp = mshf(Ck, lp_);
        const M2 al = mmul(Ak, Bim), ga = mmul(Ck, Bip);
        F.put(Factors<2>::PAL0 + lv, lane, al);
        F.put(Factors<2>::PGA0 + lv, lane, ga);
        Bk = msub(msub(Bk, mmul(al, Ckm)), mmul(ga, Akp));
        Ak = mneg(mmul(al, Akm));
        Ck = mneg(mmul(ga, Ckp));
    }
    F.Bfin = minv(Bk);
}

// x = M^{-1} b; b, x in lane layout [N0,N1,P0,P1,E0,E1]
__device__ inline void apply_factors(const Factors<2> &F, const double (&b)[6], double (&x)[6], int lane, int L)
{
    const int lnext = (lane + 1) & 63, lprev = (lane + 63) & 63;
    // E-elimination right-hand side
    V2 wb[3];
    wb[0] = V2{F.w[0][0] * b[4], F.w[0][1] * b[4]};
    wb[1] = V2{F.w[1][0] * b[5], F.w[1][1] * b[5]};
    wb[2] = vnext(wb[0]);  // lane 63 gets 0: its last node is the back surface or padding
    V2 r[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const V2 nxt = wb[q + 1];  // edge L and padded edges carry zero weights
        r[q] = V2{b[q] + F.cd * (nxt.x - wb[q].x), b[2 + q] - F.cd * (nxt.y - wb[q].y)};
    }
    // cyclic reduction of the right-hand side
    const V2 re_up = vnext(r[0]);
    const V2 t1 = mvec(F.get(Factors<2>::CR_AL, lane), r[0]), t2 = mvec(F.get(Factors<2>::CR_GA, lane), re_up);
    V2 rk{r[1].x - t1.x - t2.x, r[1].y - t1.y - t2.y};
#pragma unroll
    for (int lv = 0; lv < 6; ++lv) {
        const int s = 1 << lv;
        const V2 rm = vshf(rk, (lane - s) & 63);
        const V2 rp = vshf(rk, (lane + s) & 63);
        const V2 u1 = mvec(F.get(Factors<2>::PAL0 + lv, lane), rm), u2 = mvec(F.get(Factors<2>::PGA0 + lv, lane), rp);
        rk = V2{rk.x - u1.x - u2.x, rk.y - u1.y - u2.y};
    }
    const V2 xk = mvec(F.Bfin, rk);
    const V2 xk_prev = vprev(xk);
    const V2 s0 = mvec(F.get(Factors<2>::BINV_E, lane), r[0]), s1 = mvec(F.get(Factors<2>::BIA_E, lane), xk_prev),
             s2 = mvec(F.get(Factors<2>::BIC_E, lane), xk);
    const V2 xe{s0.x - s1.x - s2.x, s0.y - s1.y - s2.y};
    x[0] = xe.x; x[1] = xk.x; x[2] = xe.y; x[3] = xk.y;
    // back-substitute the eliminated field unknowns
    x[4] = b[4] * F.minv[0] - (F.eco[0][0] * xk_prev.x + F.eco[0][1] * xe.x + F.eco[0][2] * xk_prev.y + F.eco[0][3] * xe.y);
    x[5] = b[5] * F.minv[1] - (F.eco[1][0] * xe.x + F.eco[1][1] * xk.x + F.eco[1][2] * xe.y + F.eco[1][3] * xk.y);
}

// ---- NDF constants (Shampine & Reichelt 1997, table of kappa) ---------------
__device__ inline double ndf_gamma(int k)
{
    // sum_{j<=k} 1/j
    const double g[6] = {0.0, 1.0, 1.5, 1.5 + 1.0 / 3.0, 1.5 + 1.0 / 3.0 + 0.25, 1.5 + 1.0 / 3.0 + 0.25 + 0.2};
    double v = g[0];
#pragma unroll
    for (int j = 1; j < 6; ++j) v = (k == j) ? g[j] : v;
    return v;
}
__device__ inline double ndf_kappa(int k)
{
    const double kp[6] = {0.0, -0.1850, -1.0 / 9.0, -0.0823, -0.0415, 0.0};
    double v = kp[0];
#pragma unroll
    for (int j = 1; j < 6; ++j) v = (k == j) ? kp[j] : v;
    return v;
}
__device__ inline double ndf_alpha(int k) { return (1.0 - ndf_kappa(k)) * ndf_gamma(k); }
__device__ inline double ndf_errc(int k) { return ndf_kappa(k) * ndf_gamma(k) + 1.0 / (double)(k + 1); }

}  // namespace mtrpl

#endif  // MTRPL_SOLVER_DEVICE_HPP
