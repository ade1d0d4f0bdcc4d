// model_physics.hpp -- the drift-diffusion / recombination model of forward_solver.py:332-418 as
// per-edge and per-node formulas: right-hand side terms and their analytic derivatives.  Every solver
// backend (one wave, a team of waves) evaluates exactly these expressions; what differs
// between backends is only where the neighbouring node's values come from.
//
// Discretisation (forward_solver.py:332-372): nodes i = 0..L-1 carry N_i, P_i; edge i is the LEFT face
// of node i and carries the field E_i and the fluxes Jn_i, Jp_i; edge 0 and edge L are the surfaces.
#pragma once
#include "device_math.hpp"

namespace mtrpl {

template <class T>
struct ParT {
    T n0, p0, mun, mup, ks, Cn, Cp, Sf, Sb, tn, tp, Lam, Tm, kC, Nt, tE;
    T dx, idx, DnT, DpT, n0p0;  // idx = 1/dx
    T hmun, hmup, dnx, dpx;     // mu_n / 2, mu_p / 2, mu_n kB Tm / dx, mu_p kB Tm / dx: the coefficients of an edge flux
};
using Par = ParT<double>;

template <class T>
__device__ inline ParT<T> make_par(const double *p, double dx)
{
    ParT<T> q;
    q.n0 = (T)p[MTRPL_P_N0]; q.p0 = (T)p[MTRPL_P_P0]; q.mun = (T)p[MTRPL_P_MU_N]; q.mup = (T)p[MTRPL_P_MU_P];
    q.ks = (T)p[MTRPL_P_KS]; q.Cn = (T)p[MTRPL_P_CN]; q.Cp = (T)p[MTRPL_P_CP]; q.Sf = (T)p[MTRPL_P_SF];
    q.Sb = (T)p[MTRPL_P_SB]; q.tn = (T)p[MTRPL_P_TAUN]; q.tp = (T)p[MTRPL_P_TAUP];
    q.Lam = (T)(kQC / (p[MTRPL_P_EPS] * kEps0));  // forward_solver.py:131
    q.Tm = (T)p[MTRPL_P_TM]; q.kC = (T)p[MTRPL_P_KC]; q.Nt = (T)p[MTRPL_P_NT]; q.tE = (T)p[MTRPL_P_TAUE];
    q.dx = (T)dx;
    q.idx = (T)(1.0 / dx);
    q.DnT = (T)(p[MTRPL_P_MU_N] * kKB * p[MTRPL_P_TM]);  // mu_n*kB*Tm, left to right as forward_solver.py:357
    q.DpT = (T)(p[MTRPL_P_MU_P] * kKB * p[MTRPL_P_TM]);
    q.n0p0 = (T)(p[MTRPL_P_N0] * p[MTRPL_P_P0]);
    q.hmun = (T)(0.5 * p[MTRPL_P_MU_N]);
    q.hmup = (T)(0.5 * p[MTRPL_P_MU_P]);
    q.dnx = (T)(p[MTRPL_P_MU_N] * kKB * p[MTRPL_P_TM] / dx);
    q.dpx = (T)(p[MTRPL_P_MU_P] * kKB * p[MTRPL_P_TM] / dx);
    return q;
}

// ---- right-hand side terms ------------------------------------------------------------------------
// interior edge between node i-1 (Nm, Pm) and node i (N, P) with field E: forward_solver.py:355-359,
//   Jn = mu_n (Nm + N)/2 E + mu_n kB Tm (N - Nm)/dx,   Jp = mu_p (Pm + P)/2 E - mu_p kB Tm (P - Pm)/dx,
// with the constant factors of each term folded once per solve (mu/2, mu kB Tm / dx): five operations per flux instead
// of seven, a rounding or two away from the reference's order of operations (tests/test_gpu_solver.py::test_rhs_matches_reference
// holds the fluxes to 1e-12 of their scale)
template <class T>
__device__ inline void edge_flux(const ParT<T> &p, T Nm, T N, T Pm, T P, T E, T &jn, T &jp)
{
    jn = p.hmun * ((Nm + N) * E) + p.dnx * (N - Nm);
    jp = p.hmup * ((Pm + P) * E) - p.dpx * (P - Pm);
}
// surface recombination velocity term S*(NP - n0p0)/(N+P): forward_solver.py:346-347
template <class T>
__device__ inline T surface_term(const ParT<T> &p, T S, T N, T P)
{
    return S * (N * P - p.n0p0) * frcp1(N + P);
}
// bulk recombination ((Cn N + Cp P) + ks + 1/(tn P + tp N)) * (NP - n0p0): forward_solver.py:366
template <class T>
__device__ inline T bulk_recomb(const ParT<T> &p, T N, T P, T NP)
{
    return ((p.Cn * N + p.Cp * P) + p.ks + frcp1((p.tn * P) + (p.tp * N))) * NP;
}
// signal density of one node: ks (N P - n0 p0) (forward_solver.py:228-236,:267-269) or
// q_C (mu_n (N-n0) + mu_p (P-p0)) (:239-247,:272-274).
// The photoconductivity is emitted from the ELECTRON excess: sum_i (P_i - p0) = sum_i (N_i - n0) [+ sum_i Ntrap_i] is an
// exact invariant of the model (both carriers lose the same recombination, the surface terms enter Jn and Jp with
// opposite signs, :346-352, :366-370, :410-416, and solve() starts from dP = dN, :120-121, :133), so over the grid
// q (mu_n (N-n0) + mu_p (P-p0)) and q ((mu_n + mu_p)(N-n0) + mu_p Ntrap) have the same sum -- but P - p0 is a difference
// of nearly equal numbers once the excess has fallen below the doping (p0 / dP = 1e5 six decades into a TRTS trace:
// an error of 2.5e-11 of P is 2.7e-6 of the signal, measured on the held-out set), N - n0 is not.
template <class T>
__device__ inline T signal_density(const ParT<T> &p, int meas, T N, T P, T Ntrap)
{
    const T dN = N - p.n0;
    return (meas == MTRPL_MEAS_TRPL) ? p.ks * (N * P - p.n0p0) : (T)kQC * (p.mun * dN + p.mup * (dN + Ntrap));
}
// the std model (no trapped density: dN + 0 is dN, written out so that no addition of a zero is left in the sample loop)
template <class T>
__device__ inline T signal_density(const ParT<T> &p, int meas, T N, T P)
{
    const T dN = N - p.n0;
    return (meas == MTRPL_MEAS_TRPL) ? p.ks * (N * P - p.n0p0) : (T)kQC * (p.mun * dN + p.mup * dN);
}
// ... and literally as the reference writes it: for the zeroed state behind an early stop (:168), which is not a
// state of the model
template <class T>
__device__ inline T signal_density_literal(const ParT<T> &p, int meas, T N, T P)
{
    return (meas == MTRPL_MEAS_TRPL) ? p.ks * (N * P - p.n0p0) : (T)kQC * (p.mun * (N - p.n0) + p.mup * (P - p.p0));
}

// ---- analytic Jacobian, with the field eliminated per edge ------------------------------------------
// M = I - c*J.  Each dE_i row couples only to its own E_i and to N, P of the two adjacent nodes, so
// dE_i is eliminated exactly; what remains is block tridiagonal with 2x2 blocks in (dN_i, dP_i).
// Per edge i:  Lm, Rm   effective flux derivatives w.r.t. (N,P) of the left / right node,
//              w0, w1   dJn/dbE, dJp/dbE weights of the edge's own right-hand side entry bE_i,
//              eco[4]   dE_i = bE_i*mv - eco . (dN_{i-1}, dN_i, dP_{i-1}, dP_i),
//              mv       1/(1 + c Lam (jne + jpe)).
template <class T>
struct EdgeJac {
    M2T<T> lm, rm;
    T w0, w1, mv, e0, e1, e2, e3;
};
template <class T>
__device__ inline EdgeJac<T> edge_jac_interior(const ParT<T> &p, T cL, T dnx, T dpx, T Nm, T N, T Pm, T P, T E)
{
    EdgeJac<T> j;
    const T an = (T)0.5 * p.mun * E, ap = (T)0.5 * p.mup * E;
    const T jnm = an - dnx, jnp = an + dnx, jpm = ap + dpx, jpp = ap - dpx;
    const T jne = p.mun * (T)0.5 * (Nm + N), jpe = p.mup * (T)0.5 * (Pm + P);
    const T mi = frcp((T)1.0 + cL * (jne + jpe));
    const T g = jne * cL * mi, gp = jpe * cL * mi;
    j.lm = M2T<T>{((T)1.0 - g) * jnm, -g * jpm, -gp * jnm, ((T)1.0 - gp) * jpm};
    j.rm = M2T<T>{((T)1.0 - g) * jnp, -g * jpp, -gp * jnp, ((T)1.0 - gp) * jpp};
    j.w0 = jne * mi; j.w1 = jpe * mi; j.mv = mi;
    j.e0 = cL * mi * jnm; j.e1 = cL * mi * jnp; j.e2 = cL * mi * jpm; j.e3 = cL * mi * jpp;
    return j;
}
// derivatives of the surface term S*(NP - n0p0)/(N+P) w.r.t. N and P
template <class T>
__device__ inline void surface_jac(const ParT<T> &p, T S, T N, T P, T &sn, T &sp)
{
    const T NP = N * P - p.n0p0, s = N + P, is2 = frcp(s * s);
    sn = S * (P * s - NP) * is2;
    sp = S * (N * s - NP) * is2;
}
// edge 0: J_0 = (S0, -S0), no field
template <class T>
__device__ inline EdgeJac<T> edge_jac_front(const ParT<T> &p, T N, T P)
{
    EdgeJac<T> j;
    T s0n, s0p;
    surface_jac(p, p.Sf, N, P, s0n, s0p);
    j.lm = mzero<T>();
    j.rm = M2T<T>{s0n, s0p, -s0n, -s0p};
    j.w0 = j.w1 = j.e0 = j.e1 = j.e2 = j.e3 = (T)0;
    j.mv = (T)1;
    return j;
}
// padding beyond the film
template <class T>
__device__ inline EdgeJac<T> edge_jac_none()
{
    EdgeJac<T> j;
    j.lm = mzero<T>();
    j.rm = mzero<T>();
    j.w0 = j.w1 = j.e0 = j.e1 = j.e2 = j.e3 = (T)0;
    j.mv = (T)1;
    return j;
}
template <class T>
__device__ inline EdgeJac<T> ejsel(bool c, const EdgeJac<T> &x, const EdgeJac<T> &y)
{
    EdgeJac<T> j;
    j.lm = msel(c, x.lm, y.lm);
    j.rm = msel(c, x.rm, y.rm);
    j.w0 = c ? x.w0 : y.w0; j.w1 = c ? x.w1 : y.w1; j.mv = c ? x.mv : y.mv;
    j.e0 = c ? x.e0 : y.e0; j.e1 = c ? x.e1 : y.e1; j.e2 = c ? x.e2 : y.e2; j.e3 = c ? x.e3 : y.e3;
    return j;
}
// derivatives of the bulk recombination w.r.t. N and P
template <class T>
__device__ inline void bulk_jac(const ParT<T> &p, T N, T P, T &RN, T &RP)
{
    const T NP = N * P - p.n0p0;
    const T den = p.tn * P + p.tp * N, iden = frcp(den);
    const T kk = (p.Cn * N + p.Cp * P) + p.ks + iden;
    RN = (p.Cn - p.tp * iden * iden) * NP + kk * P;
    RP = (p.Cp - p.tn * iden * iden) * NP + kk * N;
}
// block row of node i from the effective edge matrices of its left edge (Lm, Rm) and right edge
// (lmn, rmn): A x_{i-1} + B x_i + C x_{i+1}
template <class T>
__device__ inline void node_row(T c, T cd, const M2T<T> &Lm, const M2T<T> &Rm, const M2T<T> &lmn, const M2T<T> &rmn,
                                T RN, T RP, M2T<T> &A, M2T<T> &B, M2T<T> &C)
{
    A = M2T<T>{cd * Lm.a, cd * Lm.b, -cd * Lm.c, -cd * Lm.d};
    C = M2T<T>{-cd * rmn.a, -cd * rmn.b, cd * rmn.c, cd * rmn.d};
    B = M2T<T>{(T)1.0 + cd * (Rm.a - lmn.a) + c * RN, cd * (Rm.b - lmn.b) + c * RP,
               -cd * (Rm.c - lmn.c) + c * RN, (T)1.0 - cd * (Rm.d - lmn.d) + c * RP};
}

}  // namespace mtrpl
