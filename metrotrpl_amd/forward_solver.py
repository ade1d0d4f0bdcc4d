"""Drop-in for the reference's `forward_solver` module on an MI355X.

Same names, argument order, `state` semantics and exception classes as
forward_solver.py:26-274; the arithmetic runs in the HIP kernels behind
include/metrotrpl_amd.h.  `solve()` is the B = 1 case of batch.solve_batch.
"""
import numpy as np

from . import _lib as L
from . import batch

# forward_solver.py:18-24
DEFAULT_RTOL = 1e-7
DEFAULT_ATOL = 1e-10
eps0 = 8.854 * 1e-12 * 1e-9  # [C / V m] to {C / V nm}
q = 1.0  # [e]
q_C = 1.602e-19  # [C]
kB = 8.61773e-5  # [eV / K]


def E_field(N, P, n0, p0, eps, dx, corner_E=0):
    """forward_solver.py:26-38"""
    N = np.asarray(N, dtype=np.float64)
    P = np.asarray(P, dtype=np.float64)
    if N.ndim not in (1, 2):
        raise NotImplementedError(f"Unsupported number of dimensions: {N.ndim}")
    ctx = L.default_context()
    N2, P2 = L.f64(np.atleast_2d(N)), L.f64(np.atleast_2d(P))
    E = np.empty((N2.shape[0], N2.shape[1] + 1))
    ctx.check(L.lib().mtrpl_E_field(ctx.handle, N2.shape[0], N2.shape[1], L.dptr(N2), L.dptr(P2), n0, p0, eps,
                                    dx, corner_E, L.dptr(E)))
    return E if N.ndim == 2 else E[0]


def _integral(which, dx, N, P, *a):
    N = np.asarray(N, dtype=np.float64)
    P = np.asarray(P, dtype=np.float64)
    if N.ndim not in (1, 2):
        raise ValueError(f"Invalid number of dims (got {N.ndim} dims) in Solution")
    ctx = L.default_context()
    N2, P2 = L.f64(np.atleast_2d(N)), L.f64(np.atleast_2d(P))
    out = np.empty(N2.shape[0])
    fn = L.lib().mtrpl_calculate_PL if which == "PL" else L.lib().mtrpl_calculate_TRTS
    ctx.check(fn(ctx.handle, N2.shape[0], N2.shape[1], dx, L.dptr(N2), L.dptr(P2), *a, L.dptr(out)))
    return out if N.ndim == 2 else out[0]


def calculate_PL(dx, N, P, ks, n0, p0):
    """forward_solver.py:228-236"""
    return _integral("PL", dx, N, P, ks, n0, p0)


def calculate_TRTS(dx, N, P, mu_n, mu_p, n0, p0):
    """forward_solver.py:239-247"""
    return _integral("TRTS", dx, N, P, mu_n, mu_p, n0, p0)


def _density(which, N, P, *a):
    N = np.asarray(N, dtype=np.float64)
    P = np.asarray(P, dtype=np.float64)
    N, P = np.broadcast_arrays(N, P)
    ctx = L.default_context()
    Nf, Pf = L.f64(N.reshape(-1)), L.f64(P.reshape(-1))
    out = np.empty(Nf.shape[0])
    fn = L.lib().mtrpl_calculate_RR if which == "RR" else L.lib().mtrpl_calculate_photoc
    ctx.check(fn(ctx.handle, Nf.shape[0], L.dptr(Nf), L.dptr(Pf), *a, L.dptr(out)))
    return out.reshape(N.shape) if N.ndim else out[0]


def calculate_RR(N, P, ks, n0, p0):
    """forward_solver.py:267-269"""
    return _density("RR", N, P, ks, n0, p0)


def calculate_photoc(N, P, mu_n, mu_p, n0, p0):
    """forward_solver.py:272-274"""
    return _density("photoc", N, P, mu_n, mu_p, n0, p0)


def integrate_2D(dx, y):
    """forward_solver.py:250-255: integrate_1D of every row."""
    y = np.asarray(y, dtype=np.float64)
    if y.ndim != 2:
        raise ValueError(f"Invalid number of dims (got {y.ndim} dims) in Solution")
    return calculate_PL(dx, y, np.ones_like(y), 1.0, 0.0, 0.0)


def integrate_1D(dx, y):
    """forward_solver.py:258-264 == dx * sum(y); evaluated by the PL kernel with N=y, P=1, ks=1."""
    y = np.asarray(y, dtype=np.float64)
    return calculate_PL(dx, y, np.ones_like(y), 1.0, 0.0, 0.0)


def dydt_numba(t, y, L_, dx, N0, P0, mu_n, mu_p, r_rad, CN, CP, sr0, srL, tauN, tauP, Lambda, Tm):
    """forward_solver.py:332-372 on the device (one state)."""
    par = np.array([[N0, P0, mu_n, mu_p, r_rad, CN, CP, sr0, srL, tauN, tauP, q_C / (Lambda * eps0), Tm,
                     0, 0, 1.0]])
    return batch.rhs_batch(par, [dx], np.asarray(y, dtype=np.float64)[None, :], L_)[0]


def dydt(t, y, L_, dx, N0, P0, mu_n, mu_p, r_rad, CN, CP, sr0, srL, tauN, tauP, Lambda, Tm):
    """forward_solver.py:277-329: the reference's numpy twin of dydt_numba -- the same right-hand side."""
    return dydt_numba(t, y, L_, dx, N0, P0, mu_n, mu_p, r_rad, CN, CP, sr0, srL, tauN, tauP, Lambda, Tm)


def solve(iniPar, g, state, indexes, meas="TRPL", units=None, solver=("solveivp",), model="std",
          ini_mode="density", RTOL=None, ATOL=None):
    """Calculate one simulation (forward_solver.py:41-203).

    `state` is scaled by `units` for the simulation; like the reference an ndarray
    `state` is in user units again on return (the reference multiplies in place and
    divides back, :119/:188; this shim works on a scaled copy, so the caller's
    array is bit-unchanged instead of 1-ulp perturbed).  `solver[0]` selects the
    reference's SciPy front end ("solveivp" / "odeint" / "diagnostic"); all three
    map to the same GPU integrator, "NN" is out of scope."""
    if units is None:
        units = np.ones_like(np.asarray(state, dtype=float))
    if solver[0] == "NN":
        raise ImportError("Neural network library not loaded (Reason): not part of metrotrpl_amd")
    if solver[0] not in ("solveivp", "odeint", "diagnostic"):
        raise NotImplementedError
    if ini_mode == "density":
        if len(iniPar) != g.nx:
            raise ValueError(f"Expected {g.nx} initial densities but initial condition file has {len(iniPar)}")
    elif ini_mode == "fluence":
        if len(iniPar) > 3:
            raise ValueError("Expected only fluence, absorption coef, and direction but initial "
                             f"condition file has {len(iniPar)} values")
    else:
        raise ValueError("Invalid ini_mode - must be 'density' or 'fluence'")
    if model not in L.MODEL:
        raise ValueError(f"Invalid model {model}")
    ini = np.asarray(iniPar, dtype=np.float64)
    meas_dev = meas if meas in L.MEAS else "TRPL"  # the reference integrates first and raises afterwards (:203)
    sig = batch.solve_batch(np.asarray(state, dtype=np.float64)[None, :], indexes, units, ini[None, :],
                            g.thickness, g.nx, g.tSteps, hmax=g.hmax, meas=meas_dev, model=model,
                            ini_mode=ini_mode, RTOL=RTOL, ATOL=ATOL, min_y=g.min_y)[0]
    if meas not in L.MEAS:
        raise NotImplementedError("TRTS or TRPL only")
    return sig


def dydt_numba_traps(t, y, L_, dx, N0, P0, mu_n, mu_p, r_rad, CN, CP, sr0, srL, tauN, tauP, Lambda, Tm, kC, Nt,
                     tauE):
    """forward_solver.py:374-418 on the device (one state, y = [N, N_trap, P, E])."""
    par = np.array([[N0, P0, mu_n, mu_p, r_rad, CN, CP, sr0, srL, tauN, tauP, q_C / (Lambda * eps0), Tm,
                     kC, Nt, tauE]])
    return batch.rhs_batch(par, [dx], np.asarray(y, dtype=np.float64)[None, :], L_, model="traps")[0]


MODELS = {"std": dydt_numba, "traps": dydt_numba_traps}
