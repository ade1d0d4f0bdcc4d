"""
Prototype (numpy, vectorised over 'lanes') of the GPU linear solver for the
Newton matrix M = I - c*J of the std model:
  1. eliminate dE_i (each E row couples only to its own E)  -> 2x2 block
     tridiagonal system in u_i = (dN_i, dP_i)
  2. one in-lane cyclic-reduction level (2 nodes per lane) + 6 PCR levels
     across the 64 lanes; all factors are kept so later solves are cheap.
Development tool only.
"""
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
from tools.proto_ndf import Problem, kB  # noqa


def mm(a, b):
    return np.einsum("...ij,...jk->...ik", a, b)


def mv(a, x):
    return np.einsum("...ij,...j->...i", a, x)


def inv2(B):
    det = B[..., 0, 0] * B[..., 1, 1] - B[..., 0, 1] * B[..., 1, 0]
    r = 1.0 / det
    out = np.empty_like(B)
    out[..., 0, 0] = B[..., 1, 1] * r
    out[..., 0, 1] = -B[..., 0, 1] * r
    out[..., 1, 0] = -B[..., 1, 0] * r
    out[..., 1, 1] = B[..., 0, 0] * r
    return out


def shift(x, s, fill=0.0):
    """x[i+s] with out-of-range -> fill (s may be negative)."""
    out = np.full_like(x, fill)
    n = x.shape[0]
    if s > 0:
        out[:n - s] = x[s:]
    elif s < 0:
        out[-s:] = x[:n + s]
    else:
        out[:] = x
    return out


def factor(prob, y, c, npad):
    """Blocks for npad >= nx rows (identity padding)."""
    L, dx = prob.nx, prob.dx
    N, P, E = y[:L], y[L:2 * L], y[2 * L:]
    Dn, Dp = prob.mun * kB * prob.Tm, prob.mup * kB * prob.Tm
    # edge i = 0..L : left matrix Lm (acts on u_{i-1}), right matrix Rm (on u_i), w (on bE_i)
    Lm = np.zeros((L + 1, 2, 2))
    Rm = np.zeros((L + 1, 2, 2))
    w = np.zeros((L + 1, 2))
    eco = np.zeros((L + 1, 4))  # dE_i = bE_i/m - eco . (dN_{i-1}, dN_i, dP_{i-1}, dP_i)
    minv = np.zeros(L + 1)
    i = np.arange(1, L)
    an, ap = 0.5 * prob.mun * E[i], 0.5 * prob.mup * E[i]
    jnm, jnp_ = an - Dn / dx, an + Dn / dx
    jpm, jpp = ap + Dp / dx, ap - Dp / dx
    jne, jpe = prob.mun * 0.5 * (N[i - 1] + N[i]), prob.mup * 0.5 * (P[i - 1] + P[i])
    cL = c * prob.Lam
    m = 1 + cL * (jne + jpe)
    g, gp = jne * cL / m, jpe * cL / m
    Lm[i, 0, 0], Lm[i, 0, 1], Lm[i, 1, 0], Lm[i, 1, 1] = (1 - g) * jnm, -g * jpm, -gp * jnm, (1 - gp) * jpm
    Rm[i, 0, 0], Rm[i, 0, 1], Rm[i, 1, 0], Rm[i, 1, 1] = (1 - g) * jnp_, -g * jpp, -gp * jnp_, (1 - gp) * jpp
    w[i, 0], w[i, 1] = jne / m, jpe / m
    minv[i] = 1 / m
    eco[i] = (cL / m)[:, None] * np.stack([jnm, jnp_, jpm, jpp], axis=1)
    # boundaries
    NP0 = N[0] * P[0] - prob.n0 * prob.p0
    s0 = N[0] + P[0]
    s0n, s0p = prob.Sf * (P[0] * s0 - NP0) / s0 ** 2, prob.Sf * (N[0] * s0 - NP0) / s0 ** 2
    Rm[0] = [[s0n, s0p], [-s0n, -s0p]]
    NPL = N[-1] * P[-1] - prob.n0 * prob.p0
    sL = N[-1] + P[-1]
    sLn, sLp = prob.Sb * (P[-1] * sL - NPL) / sL ** 2, prob.Sb * (N[-1] * sL - NPL) / sL ** 2
    Lm[L] = [[-sLn, -sLp], [sLn, sLp]]
    # recombination derivative
    NP = N * P - prob.n0 * prob.p0
    den = prob.tn * P + prob.tp * N
    k_ = (prob.Cn * N + prob.Cp * P) + prob.ks + 1 / den
    RN = (prob.Cn - prob.tp / den ** 2) * NP + k_ * P
    RP = (prob.Cp - prob.tn / den ** 2) * NP + k_ * N
    Sg = np.array([1.0, -1.0])[:, None]
    cd = c / dx
    A = np.zeros((npad, 2, 2))
    B = np.tile(np.eye(2), (npad, 1, 1))
    Cb = np.zeros((npad, 2, 2))
    A[:L] = cd * Sg * Lm[:L]
    B[:L] = np.eye(2) + cd * Sg * (Rm[:L] - Lm[1:]) + c * np.stack([np.stack([RN, RP], 1)] * 2, 1)
    Cb[:L] = -cd * Sg * Rm[1:]
    fac = dict(L=L, npad=npad, cd=cd, w=w, eco=eco, minv=minv, A=A, B=B, C=Cb)
    # ---- CR level (rows 2l eliminated, rows 2l+1 kept) ----
    ev, od = slice(0, npad, 2), slice(1, npad, 2)
    Binv_e = inv2(B[ev])
    fac["BiA_e"] = mm(Binv_e, A[ev])
    fac["BiC_e"] = mm(Binv_e, Cb[ev])
    fac["Binv_e"] = Binv_e
    # kept row k=2l+1: neighbours k-1 = 2l (same lane), k+1 = 2l+2 (lane l+1)
    Binv_up = shift(Binv_e, 1)  # lane l+1's even row; beyond the end: zero (C is zero there anyway)
    A_up, C_up = shift(A[ev], 1), shift(Cb[ev], 1)
    al = mm(A[od], Binv_e)
    ga = mm(Cb[od], Binv_up)
    fac["cr_al"], fac["cr_ga"] = al, ga
    Ak = -mm(al, A[ev])
    Ck = -mm(ga, C_up)
    Bk = B[od] - mm(al, Cb[ev]) - mm(ga, A_up)
    # ---- PCR over the 64 kept rows ----
    nl = npad // 2
    fac["pcr"] = []
    s = 1
    while s < nl:
        Bi = inv2(Bk)
        al = mm(Ak, shift(Bi, -s))
        ga = mm(Ck, shift(Bi, s))
        fac["pcr"].append((al, ga, s))
        Bn = Bk - mm(al, shift(Ck, -s)) - mm(ga, shift(Ak, s))
        An = -mm(al, shift(Ak, -s))
        Cn = -mm(ga, shift(Ck, s))
        Ak, Bk, Ck = An, Bn, Cn
        s *= 2
    fac["Binv_final"] = inv2(Bk)
    return fac


def solve(fac, b):
    """b = full residual vector [bN(L), bP(L), bE(L+1)] -> delta, same layout."""
    L, npad, cd = fac["L"], fac["npad"], fac["cd"]
    bN, bP, bE = b[:L], b[L:2 * L], b[2 * L:]
    wb = fac["w"] * bE[:, None]  # [L+1, 2]; zero at edges 0 and L
    Sg = np.array([1.0, -1.0])
    r = np.zeros((npad, 2))
    r[:L] = np.stack([bN, bP], 1) + cd * Sg * (wb[1:] - wb[:L])
    ev, od = slice(0, npad, 2), slice(1, npad, 2)
    re = r[ev]
    rk = r[od] - mv(fac["cr_al"], re) - mv(fac["cr_ga"], shift(re, 1))
    for al, ga, s in fac["pcr"]:
        rk = rk - mv(al, shift(rk, -s)) - mv(ga, shift(rk, s))
    xk = mv(fac["Binv_final"], rk)
    xe = mv(fac["Binv_e"], re) - mv(fac["BiA_e"], shift(xk, -1)) - mv(fac["BiC_e"], xk)
    u = np.zeros((npad, 2))
    u[ev], u[od] = xe, xk
    u = u[:L]
    dE = np.zeros(L + 1)
    i = np.arange(1, L)
    e = fac["eco"][i]
    dE[i] = bE[i] * fac["minv"][i] - (e[:, 0] * u[i - 1, 0] + e[:, 1] * u[i, 0] + e[:, 2] * u[i - 1, 1] + e[:, 3] * u[i, 1])
    dE[0], dE[L] = bE[0], bE[L]
    return np.concatenate([u[:, 0], u[:, 1], dE])


if __name__ == "__main__":
    from metrotrpl_amd import workloads as W
    ix, un = W.indexes(), W.units()
    c2 = W.config2(8)
    rng = np.random.default_rng(0)
    for nx in (128, 100):
        for k in range(8):
            prob = Problem(c2["states"][k], un, ix, c2["iniPar"][k], 2000.0 if k % 2 else 311.0, nx)
            y = prob.y0.copy()
            # perturb into a generic state
            y[:2 * nx] *= 10 ** rng.uniform(-0.3, 0.3, 2 * nx)
            y[2 * nx:] += rng.normal(0, 1e-5, nx + 1)
            for c in (1e-6, 1e-2, 1.0, 4 / 2.28):
                M = np.eye(y.size) - c * prob.jac(y)
                b = rng.normal(0, 1, y.size) * np.abs(y)
                x_ref = np.linalg.solve(M, b)
                fac = factor(prob, y, c, 128)
                x = solve(fac, b)
                res = np.linalg.norm(M @ x - b) / np.linalg.norm(b)
                err = np.max(np.abs(x - x_ref) / (np.abs(x_ref) + 1e-30 + 1e-9 * np.max(np.abs(x_ref))))
                print(f"nx={nx} k={k} c={c:g}: cond={np.linalg.cond(M):.2e} resid={res:.2e} maxrelerr={err:.2e}")
