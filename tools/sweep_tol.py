"""GPU sweep of the internal tolerance policy against the golden tight-tolerance vectors (dev tool)."""
import sys, time, os
import numpy as np
sys.path.insert(0, "/root/repo")
from metrotrpl_amd import batch, _lib as L
gold = np.load("/root/repo/tests/golden/solve.npz")
ctx = L.default_context()
cases = []
for key in [str(k) for k in gold["order"]]:
    if not key.endswith("__tight"): continue
    names, meas, solver, model, ini_mode = [str(x) for x in gold[key + "__meta"]]
    th, nx, hmax, rtol, atol, min_y = gold[key + "__geom"]
    if model != "std" or nx > 128 or min_y >= 0: continue
    cases.append((key, names.split(","), meas, ini_mode, th, int(nx), hmax, gold[key+"__tSteps"], gold[key+"__state"], gold[key+"__units"], gold[key+"__iniPar"], gold[key+"__sig"]))
def run(verbose=False):
    w4 = w6 = 0; steps = 0; who = ""
    for (key, names, meas, ini_mode, th, nx, hmax, t, st, un, ini, ref) in cases:
        idx = {n: i for i, n in enumerate(names)}
        sig, status, stats = batch.solve_batch(st[None,:], idx, un, ini[None,:], th, nx, t, hmax=hmax, meas=meas, ini_mode=ini_mode, return_stats=True)
        e = np.abs(sig[0]-ref)/np.abs(ref)
        m4 = np.abs(ref) >= 1e-4*abs(ref[0]); m6 = np.abs(ref) >= 1e-6*abs(ref[0])
        if e[m4].max() > w4: w4 = e[m4].max(); who = key
        w6 = max(w6, e[m6].max()); steps += stats[0,0]
        if verbose: print(f"   {key}: e4={e[m4].max():.2e} e6={e[m6].max():.2e} steps={stats[0,0]}")
    return w4, w6, steps/len(cases), who
import itertools
from metrotrpl_amd import workloads as W
c2 = W.config2(1024); ix, un = W.indexes(), W.units()
for ntol, age in [(0.0, 0), (0.0, 8), (1e-3, 8), (1e-2, 8), (3e-2, 8), (1e-1, 8), (3e-2, 50), (1e-1, 50)]:
    ctx.set_option("newton_tol", ntol); ctx.set_option("rate_max_age", age)
    w4, w6, s, who = run()
    t0 = time.time()
    sig, status, stats = batch.solve_batch(c2["states"], ix, un, c2["iniPar"], c2["thickness"], c2["nx"], c2["tSteps"], return_stats=True)
    dt = time.time() - t0
    print(f"newton_tol={ntol:g} age={age}: worst4={w4:.2e} ({who}) worst6={w6:.2e} golden steps={s:.0f} | cfg2x1024: {dt*1e3:.1f} ms steps={stats[:,0].mean():.0f} rhs={stats[:,1].mean():.0f} fac={stats[:,2].mean():.0f} nfail={stats[:,5].mean():.1f} bad={int((status!=0).sum())}", flush=True)
