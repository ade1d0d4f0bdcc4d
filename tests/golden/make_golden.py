"""
make_golden.py -- generate golden input/output vectors by IMPORTING the
reference (HagesLab/MetroTRPL @ /root/reference) in the build container.

Run here only (the reference cannot travel to the GPU box):

    python tests/golden/make_golden.py [--jobs 8] [--only rhs,irf,...]

Only DATA is committed (tests/golden/*.npz): inputs and the reference's
outputs.  No reference source enters the repo.  `numba` is not installed in
this image, so a 6-line no-op `njit` stand-in is written to a temp dir at run
time (`@njit(cache=True)` is used without fastmath /
parallel, so the stand-in changes speed, not arithmetic).
"""
import argparse
import os
import sys
import tempfile
import time
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)


def _import_reference():
    shim = os.path.join(tempfile.gettempdir(), "mtrpl_numba_shim")
    os.makedirs(os.path.join(shim, "numba"), exist_ok=True)
    with open(os.path.join(shim, "numba", "__init__.py"), "w") as f:
        f.write("def njit(*a, **k):\n"
                "    if len(a) == 1 and callable(a[0]) and not k:\n"
                "        return a[0]\n"
                "    return lambda fn: fn\n")
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    import forward_solver  # noqa
    import laplace  # noqa
    import sim_utils  # noqa
    import trial_move_evaluation  # noqa
    return forward_solver, laplace, sim_utils, trial_move_evaluation


# ---------------------------------------------------------------------------
def gen_rhs(out):
    fs, _, _, _ = _import_reference()
    rng = np.random.default_rng(1)
    cases = {}
    k = 0
    for L in (8, 100, 128, 512):
        for rep in range(2):
            dx = rng.uniform(1.0, 20.0)
            N = 10 ** rng.uniform(-9, -3, L)
            P = 10 ** rng.uniform(-9, -3, L)
            E = rng.normal(0, 1e-4, L + 1)
            args = (L, dx, 10 ** rng.uniform(-14, -12), 10 ** rng.uniform(-7, -5),
                    10 ** rng.uniform(4, 7), 10 ** rng.uniform(4, 7), 10 ** rng.uniform(1, 3),
                    10 ** rng.uniform(12, 14), 10 ** rng.uniform(12, 14), 10 ** rng.uniform(-3, 2),
                    10 ** rng.uniform(-3, 2), 10 ** rng.uniform(0, 3.3), 10 ** rng.uniform(0, 4),
                    fs.q_C / (rng.uniform(5, 30) * fs.eps0), 300.0)
            y = np.concatenate([N, P, E])
            cases[f"std{k}_y"] = y
            cases[f"std{k}_args"] = np.array(args, dtype=float)
            cases[f"std{k}_dy"] = fs.dydt_numba(0.0, y, *args)
            cases[f"std{k}_dy_np"] = fs.dydt(0.0, y, *args)
            Ntr = 10 ** rng.uniform(-9, -5, L)
            targs = args + (10 ** rng.uniform(0, 3), 10 ** rng.uniform(-6, -4), 10 ** rng.uniform(0, 3))
            yt = np.concatenate([N, Ntr, P, E])
            cases[f"traps{k}_y"] = yt
            cases[f"traps{k}_args"] = np.array(targs, dtype=float)
            cases[f"traps{k}_dy"] = fs.dydt_numba_traps(0.0, yt, *targs)
            k += 1
    cases["n_cases"] = np.array(k)
    # E_field (forward_solver.py:26-38) 1-D, 2-D, corner_E
    N = 10 ** rng.uniform(-9, -3, 128)
    P = 10 ** rng.uniform(-9, -3, 128)
    cases["ef_N"], cases["ef_P"] = N, P
    cases["ef_par"] = np.array([1e-13, 3e-6, 10.0, 2.4296875])
    cases["ef_E"] = fs.E_field(N, P, 1e-13, 3e-6, 10.0, 2.4296875)
    cases["ef_E_corner"] = fs.E_field(N, P, 1e-13, 3e-6, 10.0, 2.4296875, corner_E=24)
    N2 = 10 ** rng.uniform(-9, -3, (5, 32))
    P2 = 10 ** rng.uniform(-9, -3, (5, 32))
    cases["ef_N2"], cases["ef_P2"] = N2, P2
    cases["ef_E2"] = fs.E_field(N2, P2, 1e-13, 3e-6, 10.0, 2.4296875)
    # integrals (forward_solver.py:228-274)
    cases["int_y"] = rng.uniform(0, 1, 100)
    cases["int_1d"] = np.array(fs.integrate_1D(0.37, cases["int_y"]))
    cases["pl_2d"] = fs.calculate_PL(2.4296875, N2, P2, 4.8e1, 1e-13, 3e-6)
    cases["trts_2d"] = fs.calculate_TRTS(2.4296875, N2, P2, 2e6, 1.5e6, 1e-13, 3e-6)
    cases["pl_1d"] = np.array(fs.calculate_PL(2.4296875, N2[0], P2[0], 4.8e1, 1e-13, 3e-6))
    cases["trts_1d"] = np.array(fs.calculate_TRTS(2.4296875, N2[0], P2[0], 2e6, 1.5e6, 1e-13, 3e-6))
    np.savez_compressed(os.path.join(out, "rhs.npz"), **cases)
    print("rhs.npz", k, "cases")


# ---------------------------------------------------------------------------
def _solve_case(job):
    """One reference solve() call; counts RHS/Jacobian calls for information."""
    fs, _, su, _ = _import_reference()
    (name, iniPar, thickness, nx, tSteps, hmax, state, names, units, meas, solver, model, ini_mode,
     rtol, atol, min_y) = job
    g = su.Grid(thickness, nx, np.array(tSteps), hmax)
    if min_y is not None:
        g.min_y = min_y
    idx = {n: i for i, n in enumerate(names)}
    st = np.array(state, dtype=float)
    t0 = time.time()
    stats = {}
    orig = fs.solve_ivp

    def spy(*a, **k):
        r = orig(*a, **k)
        stats.update(nfev=r.nfev, njev=r.njev, nsteps=len(r.t) - 1, t_last=r.t[-1])
        return r
    fs.solve_ivp = spy
    try:
        sig = fs.solve(np.array(iniPar, dtype=float), g, st, idx, meas=meas, units=np.array(units),
                       solver=solver, model=model, ini_mode=ini_mode, RTOL=rtol, ATOL=atol)
    finally:
        fs.solve_ivp = orig
    return name, sig, stats, time.time() - t0, st


def gen_solve(out, jobs):
    from metrotrpl_amd import workloads as W
    names = W.PARAM_NAMES
    units = W.units()
    t = W.default_times()
    cases = []

    def add(name, iniPar, thickness, nx, state, meas="TRPL", model="std", ini_mode="fluence",
            tS=t, hmax=4.0, names_=names, units_=units, min_y=None, solver=("solveivp",),
            tols=((1e-10, 1e-14, "tight"), (1e-7, 1e-10, "default"))):
        for rtol, atol, tag in tols:
            cases.append((f"{name}|{tag}", list(iniPar), thickness, nx, list(tS), hmax, list(state),
                          list(names_), list(units_), meas, solver, model, ini_mode, rtol, atol, min_y))

    # config 1: staub power scan at the un-randomised initial guesses, nx=100
    s0 = W.initial_state()
    for m in range(6):
        add(f"cfg1_m{m}", W.STAUB_FLUENCES[m], W.STAUB_LENGTHS[m], 100, s0)
    # config 3 geometry (nx=128) at the initial guesses, thin + thick, low + high power
    for m in (0, 1, 4, 5):
        add(f"cfg3_m{m}", W.STAUB_FLUENCES[m], W.STAUB_LENGTHS[m], 128, s0)
    # config 2: first 16 draws
    c2 = W.config2(16)
    for k in range(16):
        add(f"cfg2_{k}", c2["iniPar"][k], c2["thickness"], c2["nx"], c2["states"][k])
    # TRTS + reversed illumination + nx=64 short cases
    add("trts_thin", W.STAUB_FLUENCES[2], 311, 128, s0, meas="TRTS")
    add("trts_thick", W.STAUB_FLUENCES[5], 2000, 128, s0, meas="TRTS")
    add("rev_thick", [1.92e12, 6e4, -1], 2000, 128, s0)
    add("nx64", W.STAUB_FLUENCES[3], 2000, 64, s0, tS=np.linspace(0, 500, 251))
    # config 5: nx=512, thin + thick, TRPL + TRTS (tight only: slow)
    add("cfg5_thick_pl", W.STAUB_FLUENCES[3], 2000, 512, s0, tols=((1e-10, 1e-14, "tight"),))
    add("cfg5_thin_trts", W.STAUB_FLUENCES[2], 311, 512, s0, meas="TRTS", tols=((1e-10, 1e-14, "tight"),))
    # traps model with real trap parameters
    tnames = names + ["kC", "Nt", "tauE"]
    tunits = list(units) + [((1e7) ** 3) / (1e9), ((1e-7) ** 3), 1]
    add("traps_thick", W.STAUB_FLUENCES[3], 2000, 128, list(s0) + [1e-10, 1e15, 50.0], model="traps",
        names_=tnames, units_=tunits)
    # Tests/test_metropolis.py:94-184 set-ups (density mode, analytic radiative decay)
    tn = ["n0", "p0", "mu_n", "mu_p", "ks", "tauN", "tauP", "Cn", "Cp", "Sf", "Sb", "eps", "Tm"]
    tu = {"n0": ((1e-7) ** 3), "p0": ((1e-7) ** 3), "mu_n": ((1e7) ** 2) / (1e9),
          "mu_p": ((1e7) ** 2) / (1e9), "ks": ((1e7) ** 3) / (1e9), "Sf": 1e-2, "Sb": 1e-2}
    tun = [tu.get(n, 1) for n in tn]
    v = {"n0": 0, "p0": 0, "mu_n": 0, "mu_p": 0, "ks": 1e-11, "Cn": 0, "Cp": 0, "tauN": 1e99,
         "tauP": 1e99, "Sf": 0, "Sb": 0, "Tm": 300, "eps": 1}
    t100 = np.linspace(0, 100, 1001)
    add("tm_rad", 1e20 * np.ones(100), 1000, 100, [v[n] for n in tn], ini_mode="density", tS=t100,
        names_=tn, units_=tun, tols=((1e-10, 1e-14, "tight"),))
    add("tm_rad_odeint", 1e20 * np.ones(100), 1000, 100, [v[n] for n in tn], ini_mode="density", tS=t100,
        names_=tn, units_=tun, solver=("odeint",), tols=((1e-10, 1e-14, "tight"),))
    v2 = dict(v, mu_n=10, mu_p=10)
    add("tm_trts", 1e20 * np.ones(100), 1000, 100, [v2[n] for n in tn], meas="TRTS", ini_mode="density",
        tS=t100, names_=tn, units_=tun, tols=((1e-7, 1e-10, "default"),))
    # Tests/test_metropolis.py:199-266 (min_y clamp)
    v3 = dict(v, mu_n=1, mu_p=1, ks=2e-10)
    pl0 = 1000 * 1e-7 * 2e-10 * 1e18 * 1e18  # calculate_PL(dx*1e-7, ...) over 100 nodes
    add("tm_depl", 1e18 * np.ones(100), 1000, 100, [v3[n] for n in tn], ini_mode="density", tS=t100,
        names_=tn, units_=tun, min_y=pl0 * 1e-2, tols=((1e-10, 1e-14, "tight"),))
    # Tests/test_metropolis.py:330-379 (fluence mode == density mode)
    add("tm_flu", [1e15, 6e4], 1000, 100, [v[n] for n in tn], ini_mode="fluence", tS=t100,
        names_=tn, units_=tun, tols=((1e-10, 1e-14, "tight"),))

    t0 = time.time()
    res = {}
    with Pool(jobs) as pool:
        for name, sig, stats, dt, st_after in pool.imap_unordered(_solve_case, cases):
            res[name] = (sig, stats, dt, st_after)
            print(f"  {name}: {dt:.1f}s {stats}", flush=True)
    print("solve cases done in", time.time() - t0)
    store = {}
    order = []
    for c in cases:
        name = c[0]
        sig, stats, dt, st_after = res[name]
        key = name.replace("|", "__")
        order.append(key)
        store[key + "__sig"] = sig
        store[key + "__iniPar"] = np.array(c[1], dtype=float)
        store[key + "__geom"] = np.array([c[2], c[3], c[5], c[13], c[14],
                                          -1.0 if c[15] is None else c[15]], dtype=float)
        store[key + "__tSteps"] = np.array(c[4])
        store[key + "__state"] = np.array(c[6], dtype=float)
        store[key + "__state_after"] = st_after
        store[key + "__units"] = np.array(c[8], dtype=float)
        store[key + "__meta"] = np.array([",".join(c[7]), c[9], c[10][0], c[11], c[12]])
        store[key + "__stats"] = np.array([stats.get("nfev", -1), stats.get("njev", -1),
                                           stats.get("nsteps", -1), stats.get("t_last", -1), dt])
    store["order"] = np.array(order)
    np.savez_compressed(os.path.join(out, "solve.npz"), **store)
    print("solve.npz", len(order), "cases")


# ---------------------------------------------------------------------------
def gen_irf(out):
    _, lp, _, _ = _import_reference()
    store = {}
    irfs = {}
    for w in (520, 745):
        irfs[w] = np.loadtxt(os.path.join(REF, "IRFs", f"irf_{w}nm.csv"), delimiter=",", skiprows=1)
    tables = lp.make_I_tables(irfs)
    for w in (520, 745):
        store[f"irf{w}_raw"] = irfs[w]
        store[f"irf{w}_table"] = tables[w][0]
    # Tests/test_convolution.py set-up: f=exp(-t), g=sin(t)
    t = np.linspace(0, 10, 1001)
    g_t = np.sin(t)
    tab = lp.make_I_tables({1: np.vstack((t, g_t)).T})[1][0]
    store["sin_table"] = tab
    dbl = np.arange(0, 10 + (t[1] - t[0]) / 4, (t[1] - t[0]) / 2)
    store["sin_f"] = np.exp(-dbl)
    store["sin_conv"] = lp.convolve(np.exp(-dbl), tab)
    ct, cy, ok = lp.do_irf_convolution(t / 2, np.exp(-t / 2), (tab, t), time_max_shift=True)
    store["sin_full_t"], store["sin_full_y"], store["sin_full_ok"] = ct, cy, np.array(ok)
    # synthetic bi-exponential PL traces through the shipped IRFs
    for w, t_end, nt in ((745, 2000.0, 1000), (520, 200.0, 400)):
        ts = np.linspace(0, t_end, nt + 1)
        y = 3e21 * np.exp(-ts / 3.0) + 4e20 * np.exp(-ts / 150.0) + 1e15
        for shift in (False, True):
            ct, cy, ok = lp.do_irf_convolution(ts, y, tables[w], time_max_shift=shift)
            tag = f"pl{w}_{int(shift)}"
            exp_y = np.log10(y) + 0.01
            exp_u = np.full_like(ts, 0.02)
            ty, tt, tv, tu = lp.post_conv_trim(ct, cy, ts, exp_y, exp_u)
            store[tag + "_t"], store[tag + "_y"] = ts, y
            store[tag + "_ok"] = np.array(ok)
            store[tag + "_conv_len"] = np.array(len(cy))
            store[tag + "_conv_t_first_last"] = np.array([ct[0], ct[-1]])
            store[tag + "_conv_argmax"] = np.array(int(np.argmax(cy)))
            store[tag + "_conv_head"] = cy[:400].copy()
            store[tag + "_conv_stride"] = cy[::97].copy()
            store[tag + "_conv_sum"] = np.array(np.sum(cy))
            store[tag + "_trim_y"], store[tag + "_trim_t"] = ty, tt
            store[tag + "_trim_len"] = np.array(len(tv))
    # post_conv_trim alone (Tests/test_convolution.py:55-79)
    exp_t = t
    conv_t = dbl[:801]
    y_c, times_c, vals_c, uncs_c = lp.post_conv_trim(conv_t, np.arange(len(conv_t), dtype=float), exp_t,
                                                     np.arange(len(exp_t), dtype=float),
                                                     np.arange(len(exp_t)) * 0.1)
    store["trim_yc"], store["trim_tc"] = y_c, times_c
    np.savez_compressed(os.path.join(out, "irf.npz"), **store)
    print("irf.npz")


# ---------------------------------------------------------------------------
def _eval_case(which):
    """The four golden set-ups of Tests/test_eval_trial_move.py (values only;
    the dict layout is the reference's public calling convention)."""
    import logging
    _, lp, _, tme = _import_reference()
    names = ["n0", "p0", "mu_n", "mu_p", "ks", "Cn", "Cp", "Tm", "Sf", "Sb", "tauN", "tauP", "eps", "m"]
    uc = {"n0": ((1e-7) ** 3), "p0": ((1e-7) ** 3), "mu_n": ((1e7) ** 2) / (1e9),
          "mu_p": ((1e7) ** 2) / (1e9), "ks": ((1e7) ** 3) / (1e9), "Sf": 1e-2, "Sb": 1e-2}
    guess = {"n0": 0, "p0": 0, "mu_n": 0, "mu_p": 0, "ks": 1e-11, "Sf": 0, "Sb": 0, "Cn": 0, "Cp": 0,
             "Tm": 300, "tauN": 1e99, "tauP": 1e99, "eps": 10, "m": 1}
    nt, t_end = 1000, 100
    mtype = ["TRPL", "TRPL"]
    ini = np.array([1e15 * np.ones(128), 1e16 * np.ones(128)])
    vals = [np.ones(nt + 1) * 23, np.ones(nt + 1) * 23]
    sigma = {"TRPL": 1}
    extra = {}
    if which == "cutoff":
        nt, t_end = 500, 50
        vals = [np.ones(nt + 1) * 23, np.ones(nt + 1) * 23]
    elif which == "scale":
        names = names[:-1] + ["_s0", "_s1"]
        guess = dict(guess, ks=1e-20, _s0=2e-17 ** -1, _s1=2e-15 ** -1)
        extra["scale_factor"] = (0.02, [0, 1, 2, 3, 4, 5], [(0, 2, 4), (1, 3, 5)])
    elif which == "mixed":
        mtype = ["TRPL", "TRTS"]
        ini = np.array([1e15 * np.ones(128), 1e15 * np.ones(128)])
        guess = dict(guess, mu_n=0.01, mu_p=0.01)
        vals = [np.ones(nt + 1) * 23, np.ones(nt + 1) * -2]
        sigma = {"TRPL": 1, "TRTS": 10}
    elif which == "irf":
        # beyond the reference's own tests: the IRF branch (:75-94) on the shipped 745 nm IRF
        irf = np.loadtxt(os.path.join(REF, "IRFs", "irf_745nm.csv"), delimiter=",", skiprows=1)
        extra["irf_convolution"] = [745, 745]
        extra["_IRF_tables"] = lp.make_I_tables({745: irf})
        guess = dict(guess, ks=2e-10)
        vals = [23 - np.linspace(0, 1, nt + 1), 24 - np.linspace(0, 2, nt + 1)]
    idx = {n: names.index(n) for n in names}
    state = [guess[n] for n in names]
    units = np.array([uc.get(n, 1) for n in names], dtype=float)
    times = [np.linspace(0, t_end, nt + 1), np.linspace(0, t_end, nt + 1)]
    uncs = [np.ones(nt + 1) * 1e-99, np.ones(nt + 1) * 1e-99]
    if which == "irf":
        uncs = [np.ones(nt + 1) * 0.02, np.ones(nt + 1) * 0.02]
    shared = {"units": units, "solver": ("solveivp",), "model": "std", "hmax": 4, "rtol": 1e-5,
              "atol": 1e-8, "_sim_info": {"lengths": [2000, 2000], "nx": [128, 128],
                                          "meas_types": mtype, "num_meas": 2},
              "_param_indexes": idx, "_init_params": ini, "ini_mode": "density",
              "_times": times, "_vals": vals, "_uncs": uncs}
    shared.update(extra)
    out = {}
    for T in (1.0, 7.5):
        uniq = {"model_uncertainty": sigma, "_T": T}
        logll, ll_funcs = tme.eval_trial_move(list(state), uniq, shared, logging.getLogger())
        out[T] = (logll, [f(T) for f in ll_funcs], [f(3.0) for f in ll_funcs])
    return which, names, state, units, mtype, ini, times, vals, uncs, sigma, extra, out


def gen_eval(out, jobs):
    store = {}
    with Pool(min(jobs, 5)) as pool:
        for r in pool.imap_unordered(_eval_case, ["base", "cutoff", "scale", "mixed", "irf"]):
            which, names, state, units, mtype, ini, times, vals, uncs, sigma, extra, o = r
            store[which + "__names"] = np.array(names)
            store[which + "__state"] = np.array(state, dtype=float)
            store[which + "__units"] = units
            store[which + "__mtype"] = np.array(mtype)
            store[which + "__ini"] = ini
            store[which + "__t_end_nt"] = np.array([times[0][-1], len(times[0]) - 1])
            store[which + "__vals"] = np.array(vals)
            store[which + "__uncs"] = np.array(uncs)
            store[which + "__sigma"] = np.array([sigma.get("TRPL", 0), sigma.get("TRTS", 0)], dtype=float)
            store[which + "__has_scale"] = np.array("scale_factor" in extra)
            store[which + "__has_irf"] = np.array("irf_convolution" in extra)
            store[which + "__logll_T1"] = np.array(o[1.0][0])
            store[which + "__ll_meas_T1"] = np.array(o[1.0][1])
            store[which + "__ll_meas_T3"] = np.array(o[1.0][2])
            store[which + "__logll_T7p5"] = np.array(o[7.5][0])
            print(which, o[1.0][0], o[7.5][0], flush=True)
    np.savez_compressed(os.path.join(out, "eval.npz"), **store)
    print("eval.npz")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--only", default="rhs,irf,eval,solve")
    a = ap.parse_args()
    only = a.only.split(",")
    if "rhs" in only:
        gen_rhs(HERE)
    if "irf" in only:
        gen_irf(HERE)
    if "eval" in only:
        gen_eval(HERE, a.jobs)
    if "solve" in only:
        gen_solve(HERE, a.jobs)
