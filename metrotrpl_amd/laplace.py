"""Drop-in for the reference's `laplace` module (IRF convolution) on an MI355X.
Same names / argument order / return tuples as laplace.py:13-212."""
import ctypes as C

import numpy as np

from . import _lib as L


def make_I_tables(irfs):
    """laplace.py:13-41: {wavelength: [k,2] (t, IRF(t))} -> {int(w): ([k,3] moments, t_irf)}"""
    ctx = L.default_context()
    I_tables = {}
    for w in irfs:
        irf = np.asarray(irfs[w], dtype=np.float64)
        w = int(w)
        t_irf = L.f64(irf[:, 0])
        f_t_irf = L.f64(irf[:, 1])
        table = np.zeros((len(f_t_irf), 3))
        ctx.check(L.lib().mtrpl_make_I_table(ctx.handle, len(f_t_irf), L.dptr(t_irf), L.dptr(f_t_irf),
                                             L.dptr(table)))
        I_tables[w] = (table, t_irf)
    return I_tables


def I_moment(t, y, m, n, u_lower=0, u_upper=1, u_spacing=100):
    """laplace.py:129-167: moment integral I_m^n of an IRF y(t), any u_spacing."""
    ctx = L.default_context()
    t, y = L.f64(t), L.f64(y)
    out = C.c_double(0.0)
    ctx.check(L.lib().mtrpl_I_moment(ctx.handle, len(y), L.dptr(t), L.dptr(y), int(m), int(n), int(u_lower),
                                     int(u_upper), int(u_spacing), C.byref(out)))
    return out.value


def convolve(resampled_y, I_table):
    """laplace.py:170-212"""
    ctx = L.default_context()
    ry, T = L.f64(resampled_y), L.f64(I_table)
    h_k = np.zeros((len(ry) - 1) // 2 + 1)
    ctx.check(L.lib().mtrpl_convolve(ctx.handle, len(ry), L.dptr(ry), T.shape[0], L.dptr(T), L.dptr(h_k)))
    return h_k


n_convolve = convolve  # laplace.py:215-246 is the same arithmetic


def do_irf_convolution(t, y, IRF_table, time_max_shift=False):
    """laplace.py:44-85 -> (convolved_t, convolved_y, success)"""
    ctx = L.default_context()
    t, y = L.f64(t), L.f64(y)
    T = L.f64(IRF_table[0])
    dt_irf = float(np.mean(np.diff(IRF_table[1])))  # laplace.py:66
    n_conv = L.lib().mtrpl_irf_conv_len(float(t[-1]), dt_irf)
    conv_t, conv_y = np.empty(n_conv), np.empty(n_conv)
    ok = C.c_int32(0)
    ctx.check(L.lib().mtrpl_irf_convolve(ctx.handle, len(t), L.dptr(t), L.dptr(y), T.shape[0], L.dptr(T), dt_irf,
                                         int(bool(time_max_shift)), n_conv, L.dptr(conv_t), L.dptr(conv_y),
                                         C.byref(ok)))
    return conv_t, conv_y, bool(ok.value)


def post_conv_trim(conv_t, conv_y, exp_t, exp_y, exp_u):
    """laplace.py:88-126"""
    ctx = L.default_context()
    conv_cutoff = np.where(exp_t < np.nanmax(conv_t))[0][-1]
    ct, cy, xq = L.f64(conv_t), L.f64(conv_y), L.f64(exp_t[:conv_cutoff + 1])
    out = np.empty(len(xq))
    ctx.check(L.lib().mtrpl_interp_linear(ctx.handle, len(ct), L.dptr(ct), L.dptr(cy), len(xq), L.dptr(xq),
                                          L.dptr(out)))
    return out, exp_t[:conv_cutoff + 1], exp_y[:conv_cutoff + 1], exp_u[:conv_cutoff + 1]
