"""
Algorithm prototype (numpy, dense linear algebra) of the integrator planned for the next round:
fixed-coefficient BDF in Nordsieck form (Hindmarsh's LSODE: `cfode` coefficients, `stode` step),
orders 1-5, driven by the same controller as tools/proto_ndf.py (SciPy-style factors, dead band,
carried Newton rate left out).  The history is z[j] = h^j y^(j)/j!, j = 0..q: six rows at order 5
instead of the eight difference rows of the NDF form, the predictor is an in-place Pascal
multiplication, a step-size change is a scaling by eta^j.  Development tool only.
"""
import sys
import time

import numpy as np

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tools")
import proto_ndf as PN  # noqa: E402


def cfode_bdf():
    """el[q][j], tesco[q] = (t1, t2, t3) for q = 1..5 (ODEPACK cfode, meth = 2)."""
    el, tes = {}, {}
    pc = np.zeros(8)
    pc[0] = 1.0
    rq1fac = 1.0
    for nq in range(1, 6):
        fnq = float(nq)
        # p(x) <- p(x) * (x + nq)
        pc[nq] = 0.0
        for i in range(nq, 0, -1):
            pc[i] = pc[i - 1] + fnq * pc[i]
        pc[0] = fnq * pc[0]
        e = pc[:nq + 1] / pc[1]
        e[1] = 1.0
        el[nq] = e.copy()
        tes[nq] = (rq1fac, (nq + 1) / e[0], (nq + 2) / e[0])
        rq1fac /= fnq
    return el, tes


EL, TES = cfode_bdf()


def bdf_nordsieck(prob, t_out, rtol, atol, hmax=np.inf, max_order=5):
    y = prob.y0.copy()
    S = y.size
    f0 = prob.f(y)
    scale = atol + rtol * np.abs(y)
    d0 = np.sqrt(np.mean((y / scale) ** 2))
    d1 = np.sqrt(np.mean((f0 / scale) ** 2))
    h0 = 0.01 * d0 / d1 if d0 > 1e-5 and d1 > 1e-5 else 1e-6
    f1 = prob.f(y + h0 * f0)
    d2 = np.sqrt(np.mean(((f1 - f0) / scale) ** 2)) / h0
    h1 = (0.01 / max(d1, d2)) ** 0.5 if max(d1, d2) > 1e-15 else max(1e-6, h0 * 1e-3)
    h = min(100 * h0, h1, hmax)
    z = np.zeros((7, S))
    z[0] = y
    z[1] = h * f0
    q = 1
    n_equal = 0
    acor_prev = np.zeros(S)
    newton_tol = 0.03
    J = prob.jac(y)
    LU = None
    t = 0.0
    st = dict(nsteps=0, niter=0, nlu=0, nrej=0, nnewtfail=0, njev=1)
    out = np.zeros(len(t_out))
    out[0] = prob.signal(y)
    io = 1
    I = np.eye(S)

    def rescale(fac):
        nonlocal h
        r = 1.0
        for j in range(1, q + 1):
            r *= fac
            z[j] *= r
        h *= fac

    while io < len(t_out):
        accepted = False
        while not accepted:
            zs = z.copy()
            for k in range(1, q + 1):          # Pascal predictor, in place
                for j in range(q, k - 1, -1):
                    z[j - 1] += z[j]
            el = EL[q]
            ypred = z[0].copy()
            scale = atol + rtol * np.abs(ypred)
            c = h * el[0]
            converged = False
            fresh = False
            while not converged:
                if LU is None:
                    J = prob.jac(ypred)        # always at the current predictor (free inside the factorisation)
                    st["njev"] += 1
                    LU = np.linalg.inv(I - c * J)
                    st["nlu"] += 1
                    fresh = True
                acor = np.zeros(S)
                yk = ypred.copy()
                dn_old = None
                for k in range(4):
                    fk = prob.f(yk)
                    st["niter"] += 1
                    dx = LU.dot(h * fk - z[1] - acor)
                    dn = np.sqrt(np.mean((el[0] * dx / scale) ** 2))
                    rate = None if dn_old is None else dn / dn_old
                    if rate is not None and (rate >= 1 or rate ** (4 - k) / (1 - rate) * dn > newton_tol):
                        break
                    acor += dx
                    yk = ypred + el[0] * acor
                    if dn < 1e-2 * newton_tol or (rate is not None and rate / (1 - rate) * dn < newton_tol):
                        converged = True
                        break
                    dn_old = dn
                if not converged:
                    if fresh:
                        break
                    LU = None
            if not converged:
                st["nnewtfail"] += 1
                z[:] = zs
                rescale(0.5)
                n_equal = 0
                LU = None
                continue
            n_iter = k + 1
            safety = 0.9 * (2 * 4 + 1) / (2 * 4 + n_iter)
            scale = atol + rtol * np.abs(yk)
            dsm = np.sqrt(np.mean((acor / scale) ** 2)) / TES[q][1]
            if dsm > 1:
                st["nrej"] += 1
                z[:] = zs
                rescale(max(0.2, safety * dsm ** (-1 / (q + 1))))
                n_equal = 0
                LU = None
            else:
                accepted = True
        for j in range(q + 1):
            z[j] += el[j] * acor
        st["nsteps"] += 1
        n_equal += 1
        t += h
        while io < len(t_out) and t_out[io] <= t * (1 + 1e-15):
            s = (t_out[io] - t) / h
            yq = z[q].copy()
            for j in range(q - 1, -1, -1):
                yq = yq * s + z[j]
            out[io] = prob.signal(yq)
            io += 1
        if n_equal < q + 1:
            acor_prev = acor
            continue
        ddn = np.sqrt(np.mean((z[q] / scale) ** 2)) / TES[q][0] if q > 1 else np.inf
        dup = np.sqrt(np.mean(((acor - acor_prev) / scale) ** 2)) / TES[q][2] if q < max_order else np.inf
        acor_prev = acor
        with np.errstate(divide="ignore"):
            facs = np.array([ddn ** (-1 / q) if q > 1 else 0.0, dsm ** (-1 / (q + 1)), dup ** (-1 / (q + 2)) if q < max_order else 0.0])
        dq = int(np.argmax(facs)) - 1
        fac = min(10.0, safety * facs.max())
        if h * fac > hmax:
            fac = hmax / h
        if dq == 0 and 0.9 <= fac < 1.2:
            n_equal = q - 2
            continue
        if dq == 1:
            z[q + 1] = acor * el[q] / (q + 1)
        q += dq
        rescale(fac)
        n_equal = 0
        LU = None
    return out, st


if __name__ == "__main__":
    from metrotrpl_amd import workloads as W
    g = np.load("/root/repo/tests/golden/solve.npz", allow_pickle=True)
    cv = np.load("/root/repo/tests/golden/solve_converged.npz", allow_pickle=True)
    ix = W.indexes()
    for nm in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["cfg2_0", "cfg2_11", "cfg3_m0", "cfg2_3"]):
        stt = g[nm + "__tight__state"]; un = g[nm + "__tight__units"]; ini = g[nm + "__tight__iniPar"]
        geom = g[nm + "__tight__geom"]; t = g[nm + "__tight__tSteps"]; ref = cv[nm + "__conv"]
        prob = PN.Problem(stt, un, ix, ini, geom[0], int(geom[1]))
        rtol = 3e-9
        atol = rtol * 1e-4 * np.max(prob.y0[:2 * prob.nx])
        for label, fn in (("NDF (differences)", PN.ndf), ("BDF (Nordsieck)", bdf_nordsieck)):
            t0 = time.time()
            sig, s = fn(prob, t, rtol, atol, hmax=4.0)
            m = ref > 1e-4 * ref.max()
            with np.errstate(all="ignore"):
                rel = np.abs(sig - ref) / np.abs(ref)
            print(f"{nm:8s} {label:18s}: max4dec={rel[m].max():.2e} steps={s['nsteps']} iter={s['niter']} lu={s['nlu']} rej={s['nrej']} nf={s['nnewtfail']} ({time.time() - t0:.1f}s)", flush=True)
