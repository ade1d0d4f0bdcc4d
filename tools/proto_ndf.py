"""
Algorithm prototype (numpy, dense linear algebra) for the GPU integrator:
variable-order (1-5) quasi-constant-step NDF (Shampine & Reichelt, "The MATLAB
ODE suite", 1997; Byrne & Hindmarsh 1975) with an analytic Jacobian.
Development tool only -- not shipped, not imported by the package.
"""
import sys
import time

import numpy as np

sys.path.insert(0, "/root/repo")
kB = 8.61773e-5
q_C = 1.602e-19
eps0 = 8.854 * 1e-12 * 1e-9


from tests.dense_model import Problem, kB, q_C, eps0  # noqa: E402,F401


KAPPA = np.array([0, -0.1850, -1 / 9, -0.0823, -0.0415, 0])
GAMMA = np.hstack((0, np.cumsum(1 / np.arange(1, 6))))
ALPHA = (1 - KAPPA) * GAMMA
ERRC = KAPPA * GAMMA + 1 / np.arange(1, 7)


def change_D(D, order, factor):
    I = np.arange(1, order + 1)[:, None]
    Jm = np.arange(1, order + 1)

    def compute_R(fac):
        M = np.zeros((order + 1, order + 1))
        M[1:, 1:] = (I - 1 - fac * Jm) / I
        M[0] = 1
        return np.cumprod(M, axis=0)
    R = compute_R(factor)
    U = compute_R(1)
    RU = R.dot(U)
    D[:order + 1] = np.dot(RU.T, D[:order + 1])


def ndf(prob, t_out, rtol, atol, hmax=np.inf, h0=None, newton_tol_frac=None, max_order=5, verbose=False, banded=None):
    y = prob.y0.copy()
    S = y.size
    t = 0.0
    t_end = t_out[-1]
    f0 = prob.f(y)
    scale = atol + rtol * np.abs(y)
    if h0 is None:
        d0 = np.sqrt(np.mean((y / scale) ** 2))
        d1 = np.sqrt(np.mean((f0 / scale) ** 2))
        h0 = 0.01 * d0 / d1 if d0 > 1e-5 and d1 > 1e-5 else 1e-6
        y1 = y + h0 * f0
        f1 = prob.f(y1)
        d2 = np.sqrt(np.mean(((f1 - f0) / scale) ** 2)) / h0
        h1 = (0.01 / max(d1, d2)) ** 0.5 if max(d1, d2) > 1e-15 else max(1e-6, h0 * 1e-3)
        h0 = min(100 * h0, h1)
    h = min(h0, hmax)
    D = np.zeros((8, S))
    D[0] = y
    D[1] = f0 * h
    order = 1
    n_equal = 0
    newton_tol = max(10 * np.finfo(float).eps / rtol, min(0.03, rtol ** 0.5)) if newton_tol_frac is None else newton_tol_frac
    J = prob.jac(y) if banded is None else y.copy()
    LU = None
    st = dict(nsteps=0, nfev=2, njev=1, nlu=0, nrej=0, nnewtfail=0, niter=0)
    out = np.zeros(len(t_out))
    out[0] = prob.signal(y)
    io = 1
    I = np.eye(S)
    while io < len(t_out):
        # one step
        current_jac = False if st["nsteps"] else True
        accepted = False
        while not accepted:
            t_new = t + h
            y_pred = np.sum(D[:order + 1], axis=0)
            scale = atol + rtol * np.abs(y_pred)
            psi = np.dot(D[1:order + 1].T, GAMMA[1:order + 1]) / ALPHA[order]
            c = h / ALPHA[order]
            converged = False
            while not converged:
                if LU is None:
                    LU = np.linalg.inv(I - c * J) if banded is None else banded[0](prob, J, c, 2 * ((prob.nx + 127) // 128) * 64)
                    st["nlu"] += 1
                d = np.zeros(S)
                yk = y_pred.copy()
                dn_old = None
                rate = None
                for k in range(4):
                    fk = prob.f(yk)
                    st["nfev"] += 1
                    st["niter"] += 1
                    if not np.all(np.isfinite(fk)):
                        break
                    dy = LU.dot(c * fk - psi - d) if banded is None else banded[1](LU, c * fk - psi - d)
                    dn = np.sqrt(np.mean((dy / scale) ** 2))
                    if dn_old is not None:
                        rate = dn / dn_old
                    if rate is not None and (rate >= 1 or rate ** (4 - k) / (1 - rate) * dn > newton_tol):
                        break
                    yk += dy
                    d += dy
                    if dn == 0 or (rate is not None and rate / (1 - rate) * dn < newton_tol):
                        converged = True
                        break
                    dn_old = dn
                if not converged:
                    if current_jac:
                        break
                    J = prob.jac(y_pred) if banded is None else y_pred.copy()
                    st["njev"] += 1
                    LU = None
                    current_jac = True
            if not converged:
                st["nnewtfail"] += 1
                h *= 0.5
                change_D(D, order, 0.5)
                n_equal = 0
                LU = None
                continue
            n_iter = k + 1
            safety = 0.9 * (2 * 4 + 1) / (2 * 4 + n_iter)
            scale = atol + rtol * np.abs(yk)
            err = ERRC[order] * d
            en = np.sqrt(np.mean((err / scale) ** 2))
            if en > 1:
                st["nrej"] += 1
                fac = max(0.2, safety * en ** (-1 / (order + 1)))
                h *= fac
                change_D(D, order, fac)
                n_equal = 0
                LU = None
            else:
                accepted = True
        st["nsteps"] += 1
        n_equal += 1
        t_old = t
        t = t_new
        y = yk
        D[order + 2] = d - D[order + 1]
        D[order + 1] = d
        for i in reversed(range(order + 1)):
            D[i] += D[i + 1]
        # dense output
        while io < len(t_out) and t_out[io] <= t * (1 + 1e-15):
            tq = t_out[io]
            yq = D[0].copy()
            p = 1.0
            for j in range(1, order + 1):
                p *= (tq - (t - (j - 1) * h)) / (h * j)
                yq += D[j] * p
            out[io] = prob.signal(yq)
            io += 1
        if n_equal < order + 1:
            continue
        if order > 1:
            em = np.sqrt(np.mean((ERRC[order - 1] * D[order] / scale) ** 2))
        else:
            em = np.inf
        if order < max_order:
            ep = np.sqrt(np.mean((ERRC[order + 1] * D[order + 2] / scale) ** 2))
        else:
            ep = np.inf
        ens = np.array([em, en, ep])
        with np.errstate(divide="ignore"):
            facs = ens ** (-1 / np.arange(order, order + 3))
        dorder = int(np.argmax(facs)) - 1
        fac = min(10.0, safety * np.max(facs))
        if h * fac > hmax:
            fac = hmax / h
        # deadband: keep h (and the LU) unless the change is worthwhile
        if dorder == 0 and 0.9 <= fac < 1.2:
            n_equal = order + 1 - 1  # re-test next step
            continue
        order += dorder
        h *= fac
        change_D(D, order, fac)
        n_equal = 0
        LU = None
    return out, st


if __name__ == "__main__":
    from metrotrpl_amd import workloads as W
    g = np.load("/root/repo/tests/golden/solve.npz", allow_pickle=True)
    cv = np.load("/root/repo/tests/golden/solve_converged.npz", allow_pickle=True)
    ix = W.indexes()
    for nm in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["cfg2_0", "cfg3_m0"]):
        st, un, ini = g[nm + "__tight__state"], g[nm + "__tight__units"], g[nm + "__tight__iniPar"]
        geom, t, ref = g[nm + "__tight__geom"], g[nm + "__tight__tSteps"], cv[nm + "__conv"]
        prob = Problem(st, un, ix, ini, geom[0], int(geom[1]))
        for rtol in (1e-7, 1e-8, 3e-9):
            t0 = time.time()
            sig, s = ndf(prob, t, rtol, rtol * 1e-4 * np.max(prob.y0[:2 * prob.nx]), hmax=4.0)
            m = ref > 1e-4 * ref.max()
            rel = np.abs(sig - ref) / np.abs(ref)
            print(f"{nm} rtol={rtol:g}: max4dec={rel[m].max():.2e} {s} ({time.time() - t0:.1f}s)")
