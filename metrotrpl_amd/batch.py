"""Batched host-side API above the C ABI: many forward_solver.solve calls
(forward_solver.py:41-203) in one launch."""
import ctypes as C
from sys import float_info

import numpy as np

from . import _lib as L

DEFAULT_RTOL = 1e-7   # forward_solver.py:18
DEFAULT_ATOL = 1e-10  # forward_solver.py:19


def pack_params(states_solver_units, indexes):
    """[B, n_params] states already multiplied by `units` -> [B,16] MTRPL_P_* slots."""
    s = np.atleast_2d(np.asarray(states_solver_units, dtype=np.float64))
    out = np.zeros((s.shape[0], 16))
    out[:, 15] = 1.0  # tauE (unused by the std model; keeps 1/tauE finite)
    for k, name in enumerate(L.PAR_SLOTS):
        if name in indexes:
            out[:, k] = s[:, indexes[name]]
        elif k < 13:
            raise KeyError(name)
    return out


def make_descs(par, thickness, nx, nt, hmax, rtol, atol, min_y, meas, model, ini_mode, ini_par,
               ini_off, t_off, out_off):
    """Structured array of mtrpl_solve_desc; scalars broadcast over the batch."""
    B = par.shape[0]
    d = np.zeros(B, dtype=L.SOLVE_DESC_DTYPE)
    d["par"] = par
    for key, val in (("thickness", thickness), ("hmax", hmax), ("rtol", rtol), ("atol", atol),
                     ("min_y", min_y), ("nx", nx), ("nt", nt), ("meas", meas), ("model", model),
                     ("ini_mode", ini_mode), ("ini_off", ini_off), ("t_off", t_off), ("out_off", out_off)):
        d[key] = val
    if ini_par is not None:
        d["ini"] = ini_par
    return d


def build_batch(states, indexes, units, iniPar, thickness, nx, nt, hmax=4.0, meas="TRPL", model="std",
                ini_mode="fluence", RTOL=None, ATOL=None, min_y=None):
    """Validate like forward_solver.solve (:100-117,:140,:203) and pack B solves that
    share one time grid.  Returns (descs, ini array)."""
    states = np.atleast_2d(np.asarray(states, dtype=np.float64))
    B = states.shape[0]
    units = np.ones(states.shape[1]) if units is None else np.asarray(units, dtype=np.float64)
    par = pack_params(states * units, indexes)
    iniPar = np.atleast_2d(np.asarray(iniPar, dtype=np.float64))
    if ini_mode == "density":
        if iniPar.shape[1] != nx:
            raise ValueError(f"Expected {nx} initial densities but initial condition file has {iniPar.shape[1]}")
        ini = L.f64(iniPar).reshape(-1)
        ini_par3, ini_off = None, np.arange(B, dtype=np.int64) * nx
    elif ini_mode == "fluence":
        if iniPar.shape[1] > 3:
            raise ValueError("Expected only fluence, absorption coef, and direction but initial "
                             f"condition file has {iniPar.shape[1]} values")
        ini_par3 = np.ones((B, 3))
        ini_par3[:, :iniPar.shape[1]] = iniPar
        ini = np.zeros(1)
        ini_off = 0
    else:
        raise ValueError("Invalid ini_mode - must be 'density' or 'fluence'")
    if model not in L.MODEL:
        raise ValueError(f"Invalid model {model}")
    meas_list = [meas] * B if isinstance(meas, str) else list(meas)
    for m in meas_list:
        if m not in L.MEAS:
            raise NotImplementedError("TRTS or TRPL only")
    d = make_descs(par, thickness, nx, nt, hmax, DEFAULT_RTOL if RTOL is None else RTOL,
                   DEFAULT_ATOL if ATOL is None else ATOL, float_info.min if min_y is None else min_y,
                   np.array([L.MEAS[m] for m in meas_list]), L.MODEL[model], L.INI[ini_mode], ini_par3, ini_off,
                   0, np.arange(B, dtype=np.int64) * nt)
    return d, ini


def solve_batch(states, indexes, units, iniPar, thickness, nx, tSteps, hmax=4.0, meas="TRPL", model="std",
                ini_mode="fluence", RTOL=None, ATOL=None, min_y=None, ctx=None, return_stats=False, pinned=False):
    """B forward simulations sharing one time grid (host arrays in, host arrays out).

    states [B, n_params] in user units (NOT modified); iniPar [B,3] (fluence mode)
    or [B,nx] (density mode); thickness scalar or [B]; meas str or list of str.
    Returns sig [B, nt] (and status [B], stats [B,8] if return_stats).

    The result is an ordinary numpy array the caller owns.  pinned=True (big batches only: B * nt >= 2^20 samples): it is
    a view of page-locked memory of the context instead -- the library copies into it asynchronously, without its
    staging pass (about 9 % faster at B = 8192) -- which stays page-locked, and keeps the native context alive, for as
    long as the caller holds it (`_lib.PinnedBuffer`): for results that are consumed and dropped, not for a history."""
    ctx = ctx or L.default_context()
    t = L.f64(tSteps)
    nt = t.shape[0]
    if t[0] != 0:
        raise ValueError("Grid error - times must start at t=0")  # sim_utils.py:269-270
    d, ini = build_batch(states, indexes, units, iniPar, thickness, nx, nt, hmax, meas, model, ini_mode,
                         RTOL, ATOL, min_y)
    B = d.shape[0]
    # big batches are pipelined inside the call (sub-launches on the context's streams): their samples come back under
    # the later sub-launches' compute -- through the library's pinned staging ring into this ordinary array, or
    # (pinned=True) straight into pinned memory the caller then holds
    out = L.pinned_f64(ctx, B * nt) if (pinned and B * nt >= (1 << 20)) else np.empty(B * nt)
    status = np.zeros(B, dtype=np.int32)
    stats = np.zeros((B, 8), dtype=np.int32)
    ctx.check(L.lib().mtrpl_solve_batch(ctx.handle, B, d.ctypes.data_as(C.c_void_p), L.dptr(ini), ini.shape[0],
                                        L.dptr(t), nt, L.dptr(out), out.shape[0], L.iptr(status), L.iptr(stats)))
    out = out.reshape(B, nt)
    if return_stats:
        return out, status, stats
    return out


def rhs_batch(par, dx, y, nx, model="std", ctx=None):
    """dydt_numba / dydt_numba_traps (forward_solver.py:332-418) for B states at once."""
    ctx = ctx or L.default_context()
    par, dx, y = L.f64(np.atleast_2d(par)), L.f64(np.atleast_1d(dx)), L.f64(np.atleast_2d(y))
    dy = np.empty_like(y)
    ctx.check(L.lib().mtrpl_rhs_batch(ctx.handle, y.shape[0], nx, L.MODEL[model], L.dptr(par), L.dptr(dx),
                                      L.dptr(y), L.dptr(dy)))
    return dy


def newton_probe(par, dx, c, y, b, nx, ctx=None):
    """x = (I - c J(y))^{-1} b with the kernel's own linear solver (test entry)."""
    ctx = ctx or L.default_context()
    par, dx, c = L.f64(np.atleast_2d(par)), L.f64(np.atleast_1d(dx)), L.f64(np.atleast_1d(c))
    y, b = L.f64(np.atleast_2d(y)), L.f64(np.atleast_2d(b))
    x = np.empty_like(y)
    ctx.check(L.lib().mtrpl_newton_probe(ctx.handle, y.shape[0], nx, L.dptr(par), L.dptr(dx), L.dptr(c),
                                         L.dptr(y), L.dptr(b), L.dptr(x)))
    return x
