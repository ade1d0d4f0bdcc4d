"""Accuracy census on PT-like solves (thin + thick films x 3 fluences, random states) vs the converged oracle (dev tool)."""
import sys, time, os
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np
from multiprocessing import Pool
sys.path.insert(0, "/root/repo")
from metrotrpl_amd import workloads as W
NCH = int(sys.argv[1]) if len(sys.argv) > 1 else 48
e = W.pt_ensemble(NCH); ix, un = W.indexes(), W.units()
def one(k):
    from oracle import oracle as O
    g = O.Grid(e["thickness"][k], 128, e["tSteps"], 4.0)
    return O.solve(e["iniPar"][k], g, e["states"][k], ix, units=un, ini_mode="fluence", RTOL=1e-12, ATOL=1e-17)
if __name__ == "__main__":
    n = NCH * 6
    with Pool(16) as p:
        refs = np.array(p.map(one, range(n), chunksize=4))
    from metrotrpl_amd import batch, _lib as L
    import time
    ctx = L.default_context()
    c2 = W.config2(1024)
    for ar, cap, ntol, fs in ((1e-4, 5e-9, 0.03, 4.0), (1e-4, 5e-9, 0.01, 4.0), (1e-4, 5e-9, 0.003, 4.0), (1e-4, 5e-9, 0.01, 0.0), (1e-4, 4e-9, 0.03, 4.0), (1e-4, 3e-9, 0.03, 4.0)):
        ctx.set_option("atol_rel", ar); ctx.set_option("rtol_cap", cap); ctx.set_option("rtol_floor", min(cap, 5e-9)); ctx.set_option("newton_tol", ntol); ctx.set_option("rate_first_safety", fs)
        sig, status, stats = batch.solve_batch(e["states"], ix, un, e["iniPar"], e["thickness"], 128, e["tSteps"], return_stats=True)
        errs = np.zeros(n)
        for k in range(n):
            m = np.abs(refs[k]) >= 1e-4 * abs(refs[k][0])
            errs[k] = np.max(np.abs(sig[k][m] - refs[k][m]) / np.abs(refs[k][m]))
        thin = e["thickness"] < 1000
        t0 = time.time(); s2, st2, stat2 = batch.solve_batch(c2["states"], ix, un, c2["iniPar"], c2["thickness"], 128, c2["tSteps"], return_stats=True); dt = time.time() - t0
        print(f"ntol={ntol:g} fs={fs:g} rtol_cap={cap:g}: thin max={errs[thin].max():.2e} p99={np.percentile(errs[thin],99):.2e} med={np.median(errs[thin]):.2e} | thick max={errs[~thin].max():.2e} med={np.median(errs[~thin]):.2e} | steps {stats[:,0].mean():.0f}/{stats[:,0].max()} bad={(status!=0).sum()} | cfg2x1024: {dt*1e3:.1f} ms steps {stat2[:,0].mean():.0f}/{stat2[:,0].max()} rhs {stat2[:,1].mean():.0f} bad={(st2!=0).sum()}", flush=True)
