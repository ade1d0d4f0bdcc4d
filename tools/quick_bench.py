"""Quick GPU timing of the config-2 batch through the host API (dev tool)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from metrotrpl_amd import batch, workloads as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
c = W.config2(B)
ix, un = W.indexes(), W.units()
for rep in range(3):
    t0 = time.time()
    sig, status, stats = batch.solve_batch(c["states"], ix, un, c["iniPar"], c["thickness"], c["nx"], c["tSteps"],
                                           return_stats=True)
    dt = time.time() - t0
    print(f"B={B}: {dt*1e3:.1f} ms -> {B/dt:.0f} solves/s; status!=0: {(status!=0).sum()}; steps mean/max "
          f"{stats[:,0].mean():.0f}/{stats[:,0].max()}; rhs {stats[:,1].mean():.0f}; fac {stats[:,2].mean():.0f}; "
          f"jac {stats[:,3].mean():.0f}; rej {stats[:,4].mean():.1f}; nfail {stats[:,5].mean():.1f}", flush=True)
