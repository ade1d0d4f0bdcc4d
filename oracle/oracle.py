"""
oracle.py -- TEST INFRASTRUCTURE ONLY (the checker, never the product).

CPU restatement of MetroTRPL's likelihood hot path: the reference's own
call structure (SciPy LSODA driving a compiled RHS through a Python
trampoline, forward_solver.py:142-168) with Numba's JIT replaced by the gcc
build of oracle/mtrpl_oracle.c.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; metrotrpl_amd/ never does.

The integrator is third-party (SciPy ODEPACK LSODA, pinned scipy==1.12.0 in the
reference's Requirements/requirements_win_basic.txt:14); this container and the
GPU image ship SciPy 1.15.3.  It is the same Fortran ODEPACK either way.

Parity pin: tests/test_oracle_golden.py compares every function here with
vectors captured from the reference itself (tests/golden/make_golden.py).
"""
import ctypes as C
import os
import subprocess
from sys import float_info

import numpy as np
from scipy.integrate import odeint, solve_ivp

try:
    from threadpoolctl import threadpool_limits

    def _one_blas_thread():
        return threadpool_limits(limits=1, user_api="blas")
except ImportError:  # pragma: no cover
    import contextlib

    def _one_blas_thread():
        return contextlib.nullcontext()

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmtrpl_oracle.so")

# forward_solver.py:18-24
DEFAULT_RTOL = 1e-7
DEFAULT_ATOL = 1e-10
eps0 = 8.854 * 1e-12 * 1e-9
q_C = 1.602e-19
kB = 8.61773e-5
# sim_utils.py:23
NEGATIVE_FRAC_TOL = 0.2

_dp = C.POINTER(C.c_double)


def build(force=False):
    """Compile the C restatement with gcc (build the checker; not use it)."""
    src = os.path.join(_HERE, "mtrpl_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        d, i = C.c_double, C.c_int
        L.mtrpl_oracle_dydt_std.argtypes = [d, _dp, _dp, i] + [d] * 14
        L.mtrpl_oracle_dydt_std.restype = None
        L.mtrpl_oracle_dydt_traps.argtypes = [d, _dp, _dp, i] + [d] * 17
        L.mtrpl_oracle_dydt_traps.restype = None
        L.mtrpl_oracle_E_field.argtypes = [_dp, _dp, i, d, d, d, d, d, _dp]
        L.mtrpl_oracle_E_field.restype = None
        L.mtrpl_oracle_integrate_1D.argtypes = [d, _dp, i]
        L.mtrpl_oracle_integrate_1D.restype = d
        L.mtrpl_oracle_calculate_PL.argtypes = [d, _dp, _dp, i, i, d, d, d, _dp]
        L.mtrpl_oracle_calculate_PL.restype = None
        L.mtrpl_oracle_calculate_TRTS.argtypes = [d, _dp, _dp, i, i, d, d, d, d, _dp]
        L.mtrpl_oracle_calculate_TRTS.restype = None
        L.mtrpl_oracle_scale_clamp.argtypes = [_dp, i, d, d]
        L.mtrpl_oracle_scale_clamp.restype = None
        L.mtrpl_oracle_convolve.argtypes = [_dp, i, _dp, i, _dp]
        L.mtrpl_oracle_convolve.restype = None
        L.mtrpl_oracle_I_moment.argtypes = [_dp, _dp, i, i, i]
        L.mtrpl_oracle_I_moment.restype = d
        L.mtrpl_oracle_make_I_table.argtypes = [_dp, _dp, i, _dp]
        L.mtrpl_oracle_make_I_table.restype = None
        L.mtrpl_oracle_interp_linear.argtypes = [_dp, _dp, i, _dp, i, _dp]
        L.mtrpl_oracle_interp_linear.restype = None
        L.mtrpl_oracle_loglik.argtypes = [_dp, _dp, _dp, i, d, d, _dp, i, d, _dp]
        L.mtrpl_oracle_loglik.restype = i
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# --------------------------------------------------------------------------
# sim_utils.py:246-277
class Grid:
    def __init__(self, thickness, nx, tSteps, hmax):
        self.thickness = thickness
        self.nx = nx
        self.dx = self.thickness / self.nx
        self.xSteps = np.linspace(self.dx / 2, self.thickness - self.dx / 2, self.nx)
        if tSteps[0] != 0:
            raise ValueError("Grid error - times must start at t=0")
        self.tSteps = tSteps
        self.start_time = 0
        self.nt = len(tSteps) - 1
        self.hmax = hmax
        self.final_time = self.tSteps[-1]
        self.min_y = float_info.min


# --------------------------------------------------------------------------
# RHS (forward_solver.py:332-418)
def dydt_std(t, y, L, dx, N0, P0, mu_n, mu_p, r_rad, CN, CP, sr0, srL, tauN, tauP, Lambda, Tm):
    y = _c(y)
    dy = np.empty(3 * L + 1)
    lib().mtrpl_oracle_dydt_std(t, _p(y), _p(dy), L, dx, N0, P0, mu_n, mu_p, r_rad, CN, CP,
                                sr0, srL, tauN, tauP, Lambda, Tm)
    return dy


def dydt_traps(t, y, L, dx, N0, P0, mu_n, mu_p, r_rad, CN, CP, sr0, srL, tauN, tauP, Lambda, Tm,
               kC, Nt, tauE):
    y = _c(y)
    dy = np.empty(4 * L + 1)
    lib().mtrpl_oracle_dydt_traps(t, _p(y), _p(dy), L, dx, N0, P0, mu_n, mu_p, r_rad, CN, CP,
                                  sr0, srL, tauN, tauP, Lambda, Tm, kC, Nt, tauE)
    return dy


MODELS = {"std": dydt_std, "traps": dydt_traps}


# forward_solver.py:26-38
def E_field(N, P, n0, p0, eps, dx, corner_E=0):
    N = _c(N)
    P = _c(P)
    if N.ndim == 1:
        E = np.empty(N.shape[0] + 1)
        lib().mtrpl_oracle_E_field(_p(N), _p(P), N.shape[0], n0, p0, eps, dx, corner_E, _p(E))
        return E
    if N.ndim == 2:
        E = np.empty((N.shape[0], N.shape[1] + 1))
        for k in range(N.shape[0]):
            lib().mtrpl_oracle_E_field(_p(N[k]), _p(P[k]), N.shape[1], n0, p0, eps, dx, corner_E,
                                       _p(E[k]))
        return E
    raise NotImplementedError(f"Unsupported number of dimensions: {N.ndim}")


# forward_solver.py:258-264
def integrate_1D(dx, y):
    y = _c(y)
    return lib().mtrpl_oracle_integrate_1D(dx, _p(y), y.shape[0])


# forward_solver.py:228-236
def calculate_PL(dx, N, P, ks, n0, p0):
    N = _c(N)
    P = _c(P)
    if N.ndim not in (1, 2):
        raise ValueError(f"Invalid number of dims (got {N.ndim} dims) in Solution")
    N2 = np.atleast_2d(N)
    P2 = np.atleast_2d(P)
    out = np.empty(N2.shape[0])
    lib().mtrpl_oracle_calculate_PL(dx, _p(N2), _p(P2), N2.shape[0], N2.shape[1], ks, n0, p0, _p(out))
    return out if N.ndim == 2 else out[0]


# forward_solver.py:239-247
def calculate_TRTS(dx, N, P, mu_n, mu_p, n0, p0):
    N = _c(N)
    P = _c(P)
    if N.ndim not in (1, 2):
        raise ValueError(f"Invalid number of dims (got {N.ndim} dims) in Solution")
    N2 = np.atleast_2d(N)
    P2 = np.atleast_2d(P)
    out = np.empty(N2.shape[0])
    lib().mtrpl_oracle_calculate_TRTS(dx, _p(N2), _p(P2), N2.shape[0], N2.shape[1], mu_n, mu_p,
                                      n0, p0, _p(out))
    return out if N.ndim == 2 else out[0]


# --------------------------------------------------------------------------
# forward_solver.py:41-225 (solve); NN branch out of scope.
def solve(iniPar, g, state, indexes, meas="TRPL", units=None, solver=("solveivp",), model="std",
          ini_mode="density", RTOL=None, ATOL=None, stats=None):
    if units is None:
        units = np.ones_like(state)
    if RTOL is None:
        RTOL = DEFAULT_RTOL
    if ATOL is None:
        ATOL = DEFAULT_ATOL
    if solver[0] not in ("solveivp", "odeint", "diagnostic"):
        raise NotImplementedError
    # :100-117
    if ini_mode == "density":
        if len(iniPar) == g.nx:
            init_dN = np.asarray(iniPar, dtype=float) * 1e-21
        else:
            raise ValueError(f"Expected {g.nx} initial densities but initial condition file has {len(iniPar)}")
    elif ini_mode == "fluence":
        if len(iniPar) <= 3:
            fluence = iniPar[0] * 1e-14
            alpha = iniPar[1] * 1e-7
            init_dN = fluence * alpha * np.exp(-alpha * g.xSteps)
            try:
                init_dN = init_dN[::np.sign(int(iniPar[2]))]
            except (IndexError, ValueError):
                pass
        else:
            raise ValueError("Expected only fluence, absorption coef, and direction but initial "
                             f"condition file has {len(iniPar)} values")
    else:
        raise ValueError("Invalid ini_mode - must be 'density' or 'fluence'")

    # :119 -- the reference scales the caller's array in place and divides it
    # back (:188/:196); the oracle works on a scaled copy s and leaves `state`
    # untouched (documented 1-ulp divergence on the caller's array only).
    s = np.asarray(state, dtype=float) * np.asarray(units, dtype=float)
    ix = indexes
    N = init_dN + s[ix["n0"]]
    P = init_dN + s[ix["p0"]]
    E_f = E_field(N, P, s[ix["n0"]], s[ix["p0"]], s[ix["eps"]], g.dx)

    base = (g.nx, g.dx, s[ix["n0"]], s[ix["p0"]], s[ix["mu_n"]], s[ix["mu_p"]], s[ix["ks"]],
            s[ix["Cn"]], s[ix["Cp"]], s[ix["Sf"]], s[ix["Sb"]], s[ix["tauN"]], s[ix["tauP"]],
            ((q_C) / (s[ix["eps"]] * eps0)), s[ix["Tm"]])
    if model == "std":
        init_condition = np.concatenate([N, P, E_f], axis=None)
        args = base
    elif model == "traps":
        init_condition = np.concatenate([N, np.zeros_like(N), P, E_f], axis=None)
        args = base + (s[ix["kC"]], s[ix["Nt"]], s[ix["tauE"]])
    else:
        raise ValueError(f"Invalid model {model}")

    f = MODELS[model]
    # LSODA factorises its dense Jacobian with LAPACK: one BLAS thread, so that the golden traces
    # (generated single-threaded) are reproduced bit for bit whatever the host's core count
    with _one_blas_thread():
        return _integrate_and_reduce(f, args, solver, model, g, init_condition, s, ix, meas, RTOL, ATOL, stats)


def _integrate_and_reduce(f, args, solver, model, g, init_condition, s, ix, meas, RTOL, ATOL, stats):
    if solver[0] in ("solveivp", "diagnostic"):
        sol = solve_ivp(lambda t, y: f(t, y, *args), [g.start_time, g.final_time], init_condition,
                        method="LSODA", dense_output=True, max_step=g.hmax, rtol=RTOL, atol=ATOL)
        data = sol.sol(g.tSteps).T
        data[g.tSteps > sol.t[-1]] = 0  # :168
        if stats is not None:
            stats.update(nfev=sol.nfev, njev=sol.njev, nsteps=len(sol.t) - 1, t_last=sol.t[-1])
    else:
        data = odeint(f, init_condition, g.tSteps, args=args, hmax=g.hmax, rtol=RTOL, atol=ATOL,
                      tfirst=True)

    if model == "std":
        Nt_, Pt_, _ = np.split(data, [g.nx, 2 * g.nx], axis=1)
    else:
        Nt_, _, Pt_, _ = np.split(data, [g.nx, 2 * g.nx, 3 * g.nx], axis=1)

    if meas == "TRPL":
        sig = calculate_PL(g.dx, Nt_, Pt_, s[ix["ks"]], s[ix["n0"]], s[ix["p0"]])
        scale = 1e23
    elif meas == "TRTS":
        sig = calculate_TRTS(g.dx, Nt_, Pt_, s[ix["mu_n"]], s[ix["mu_p"]], s[ix["n0"]], s[ix["p0"]])
        scale = 1e9
    else:
        raise NotImplementedError("TRTS or TRPL only")
    sig = _c(sig)
    lib().mtrpl_oracle_scale_clamp(_p(sig), sig.shape[0], scale, g.min_y)  # :189-192
    return sig


# --------------------------------------------------------------------------
# laplace.py
def I_moment(t, y, m, n, u_spacing=100):
    t = _c(t)
    y = _c(y)
    return lib().mtrpl_oracle_I_moment(_p(t), _p(y), m, n, u_spacing)


def make_I_tables(irfs):
    """laplace.py:13-41"""
    out = {}
    for w in irfs:
        irf = np.asarray(irfs[w], dtype=float)
        w = int(w)
        t_irf = _c(irf[:, 0])
        f_irf = _c(irf[:, 1])
        table = np.zeros((len(f_irf), 3))
        lib().mtrpl_oracle_make_I_table(_p(t_irf), _p(f_irf), len(f_irf), _p(table))
        out[w] = (table, t_irf)
    return out


def interp_linear(x, y, xq):
    x = _c(x)
    y = _c(y)
    xq = _c(xq)
    yq = np.empty(xq.shape[0])
    lib().mtrpl_oracle_interp_linear(_p(x), _p(y), x.shape[0], _p(xq), xq.shape[0], _p(yq))
    return yq


def convolve(resampled_y, I_table):
    """laplace.py:170-212"""
    ry = _c(resampled_y)
    T = _c(I_table)
    h = np.empty((ry.shape[0] - 1) // 2 + 1)
    lib().mtrpl_oracle_convolve(_p(ry), ry.shape[0], _p(T), T.shape[0], _p(h))
    return h


def do_irf_convolution(t, y, IRF_table, time_max_shift=False):
    """laplace.py:44-85"""
    success = True
    t_irf = IRF_table[1]
    dt_irf = np.mean(np.diff(t_irf))
    resampled_t = np.arange(0, t[-1] + dt_irf / 4, dt_irf / 2)
    if resampled_t[-1] > t[-1]:
        resampled_t[-1] = t[-1]
    resampled_y = interp_linear(t, y, resampled_t)
    if any(np.isnan(resampled_y)):
        success = False
    convolved_y = convolve(resampled_y, IRF_table[0])
    convolved_t = resampled_t[::2]
    if time_max_shift:
        convolved_t -= convolved_t[np.argmax(convolved_y)]
        if convolved_t[-1] == 0:
            success = False
    return convolved_t, convolved_y, success


def post_conv_trim(conv_t, conv_y, exp_t, exp_y, exp_u):
    """laplace.py:88-126"""
    conv_cutoff = np.where(exp_t < np.nanmax(conv_t))[0][-1]
    conv_y = interp_linear(conv_t, conv_y, exp_t[:conv_cutoff + 1])
    return conv_y, exp_t[:conv_cutoff + 1], exp_y[:conv_cutoff + 1], exp_u[:conv_cutoff + 1]


# --------------------------------------------------------------------------
# utils.py:16-31
def set_min_y(sol, vals, scale_shift):
    min_y = 10 ** min(vals - scale_shift)
    i_final = np.searchsorted(-sol, -min_y)
    sol[i_final:] = min_y
    return sol, min_y, len(sol[i_final:])


def loglik_tail(sol, vals_c, uncs_c, scale_shift, sigma, temperatures):
    """trial_move_evaluation.py:117-130,147-165 for one measurement.
    Returns ll at every temperature (array) or -inf array on failure."""
    sol = _c(sol).copy()
    vals_c = _c(vals_c)
    uncs_c = _c(uncs_c)
    T = _c(np.atleast_1d(temperatures))
    out = np.empty(T.shape[0])
    rc = lib().mtrpl_oracle_loglik(_p(sol), _p(vals_c), _p(uncs_c), sol.shape[0], scale_shift,
                                   sigma, _p(T), T.shape[0], NEGATIVE_FRAC_TOL, _p(out))
    if rc != 0:
        out[:] = -np.inf
    out[np.isnan(out)] = -np.inf
    return out


def one_sim_likelihood(meas_index, state, unique_fields, shared_fields, temperatures=None):
    """trial_move_evaluation.py:30-166 (fittable fluence/absorption act on a
    COPY of the init-params row: the reference mutates the shared row
    cumulatively, :44/:51 -- a reference bug the oracle does not reproduce)."""
    iniPar = np.array(shared_fields["_init_params"][meas_index], dtype=float, copy=True)
    meas_type = shared_fields["_sim_info"]["meas_types"][meas_index]
    irf_convolution = shared_fields.get("irf_convolution", None)
    pidx = shared_fields["_param_indexes"]
    if temperatures is None:
        temperatures = [unique_fields.get("_T", 1)]
    nT = len(np.atleast_1d(temperatures))
    fail = np.full(nT, -np.inf)

    def _grp(spec, prefix):
        if spec[2] is not None and len(spec[2]) > 0:
            g0 = meas_index
            for c_grp in spec[2]:
                if meas_index in c_grp:
                    g0 = c_grp[0]
                    break
            return f"{prefix}{g0}"
        return f"{prefix}{meas_index}"

    ff = shared_fields.get("fittable_fluences", None)
    if ff is not None and meas_index in ff[1]:
        iniPar[0] *= state[pidx[_grp(ff, "_f")]]
    fa = shared_fields.get("fittable_absps", None)
    if fa is not None and meas_index in fa[1]:
        iniPar[1] *= state[pidx[_grp(fa, "_a")]]
    fs = shared_fields.get("scale_factor", None)
    if fs is not None and meas_index in fs[1]:
        scale_shift = np.log10(state[pidx[_grp(fs, "_s")]])
    else:
        scale_shift = 0

    times = shared_fields["_times"][meas_index]
    try:
        g = Grid(shared_fields["_sim_info"]["lengths"][meas_index],
                 shared_fields["_sim_info"]["nx"][meas_index], times, shared_fields["hmax"])
        sol = solve(iniPar, g, state, pidx, meas=meas_type, units=shared_fields["units"],
                    solver=shared_fields["solver"], model=shared_fields["model"],
                    ini_mode=shared_fields["ini_mode"], RTOL=shared_fields["rtol"],
                    ATOL=shared_fields["atol"])
        tSteps = g.tSteps
    except ValueError:
        return fail, None

    vals_c = shared_fields["_vals"][meas_index]
    uncs_c = shared_fields["_uncs"][meas_index]
    try:
        if irf_convolution is not None and irf_convolution[meas_index] != 0:
            wave = int(irf_convolution[meas_index])
            tSteps, sol, success = do_irf_convolution(tSteps, sol, shared_fields["_IRF_tables"][wave],
                                                      time_max_shift=True)
            if not success:
                raise ValueError("Conv failed")
            sol, _, vals_c, uncs_c = post_conv_trim(tSteps, sol, times, vals_c, uncs_c)
        else:
            sol = sol[-len(times):]
    except ValueError:
        return fail, None

    sol = np.array(sol, dtype=float, copy=True)
    n_fails = np.sum(sol < 0)
    if not n_fails < NEGATIVE_FRAC_TOL * len(sol):
        return fail, sol
    sol[sol < 0] *= -1
    if shared_fields.get("force_min_y", False):
        sol, _, _ = set_min_y(sol, vals_c, scale_shift)
    sigma = unique_fields["model_uncertainty"][meas_type]
    return loglik_tail(sol, vals_c, uncs_c, scale_shift, sigma, temperatures), sol


def eval_trial_move(state, unique_fields, shared_fields, temperatures=None):
    """trial_move_evaluation.py:9-28.  Returns (logll at T=_T, ll table
    [num_meas, nT])."""
    num_meas = shared_fields["_sim_info"]["num_meas"]
    rows = []
    for i in range(num_meas):
        ll, _ = one_sim_likelihood(i, state, unique_fields, shared_fields, temperatures)
        rows.append(ll)
    table = np.array(rows)
    return np.sum(table[:, 0]), table
