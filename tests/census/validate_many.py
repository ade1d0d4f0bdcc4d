"""Accuracy census on the GPU box: N config-2 draws vs the converged oracle (1e-12/1e-17), several tolerance settings (dev tool)."""
import sys, time, os
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np
from multiprocessing import Pool
sys.path.insert(0, "/root/repo")
from metrotrpl_amd import workloads as W
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
c = W.config2(N); ix, un = W.indexes(), W.units()
def one(k):
    from oracle import oracle as O
    g = O.Grid(c["thickness"], c["nx"], c["tSteps"], c["hmax"])
    return O.solve(c["iniPar"][k], g, c["states"][k], ix, units=un, ini_mode="fluence", RTOL=1e-12, ATOL=1e-17)
if __name__ == "__main__":
    t0 = time.time()
    with Pool(16) as p:
        refs = np.array(p.map(one, range(N), chunksize=4))
    print("oracle done", time.time() - t0, flush=True)
    from metrotrpl_amd import batch, _lib as L
    ctx = L.default_context()
    for rc in (5e-9, 3e-9, 2e-9, 1.5e-9):
        ctx.set_option("rtol_force", rc)
        sig, status, stats = batch.solve_batch(c["states"], ix, un, c["iniPar"], c["thickness"], c["nx"], c["tSteps"], return_stats=True)
        errs = []
        for k in range(N):
            m = np.abs(refs[k]) >= 1e-4 * abs(refs[k][0])
            errs.append(np.max(np.abs(sig[k][m] - refs[k][m]) / np.abs(refs[k][m])))
        errs = np.array(errs)
        print(f"rtol_cap={rc:g}: max={errs.max():.2e} p99={np.percentile(errs,99):.2e} p90={np.percentile(errs,90):.2e} median={np.median(errs):.2e} argmax={errs.argmax()} steps mean={stats[:,0].mean():.0f} max={stats[:,0].max()} bad={(status!=0).sum()}", flush=True)
