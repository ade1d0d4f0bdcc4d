"""GPU sweep of the Newton rate carry-over policy: speed on config-2 x1024 and accuracy census vs converged oracle (dev tool)."""
import sys, time, os
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np
from multiprocessing import Pool
sys.path.insert(0, "/root/repo")
from metrotrpl_amd import workloads as W
N = 256
c = W.config2(1024); ix, un = W.indexes(), W.units()
def one(k):
    from oracle import oracle as O
    g = O.Grid(c["thickness"], c["nx"], c["tSteps"], c["hmax"])
    return O.solve(c["iniPar"][k], g, c["states"][k], ix, units=un, ini_mode="fluence", RTOL=1e-12, ATOL=1e-17)
if __name__ == "__main__":
    with Pool(16) as p:
        refs = np.array(p.map(one, range(N), chunksize=4))
    from metrotrpl_amd import batch, _lib as L
    ctx = L.default_context()
    for rfs, age, ntol in [(4.0, 8, 0.03), (2.0, 20, 0.1), (2.0, 20, 0.2), (2.0, 20, 0.3), (1.0, 20, 0.3), (2.0, 20, 0.5), (1.0, 50, 0.5)]:
        ctx.set_option("rate_first_safety", rfs); ctx.set_option("rate_max_age", age); ctx.set_option("newton_tol", ntol)
        for rep in range(2):
            t0 = time.time()
            sig, status, stats = batch.solve_batch(c["states"], ix, un, c["iniPar"], c["thickness"], c["nx"], c["tSteps"], return_stats=True)
            dt = time.time() - t0
        errs = []
        for k in range(N):
            m = np.abs(refs[k]) >= 1e-4 * abs(refs[k][0])
            errs.append(np.max(np.abs(sig[k][m] - refs[k][m]) / np.abs(refs[k][m])))
        errs = np.array(errs)
        print(f"first_safety={rfs:g} age={age} ntol={ntol:g}: {dt*1e3:.1f} ms steps={stats[:,0].mean():.0f}/{stats[:,0].max()} rhs={stats[:,1].mean():.0f} fac={stats[:,2].mean():.0f} rej={stats[:,4].mean():.1f} nfail={stats[:,5].mean():.1f} bad={(status!=0).sum()} | err max={errs.max():.2e} p99={np.percentile(errs,99):.2e} med={np.median(errs):.2e}", flush=True)
