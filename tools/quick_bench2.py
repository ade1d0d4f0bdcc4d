"""Wave kernel vs team kernel timing on the config-2 batch (dev tool)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from metrotrpl_amd import batch, _lib as L, workloads as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
c = W.config2(B); ix, un = W.indexes(), W.units(); ctx = L.default_context()
for variant in (0, 1):
    ctx.set_option("kernel_variant", variant)
    for rep in range(2):
        t0 = time.time()
        sig, status, stats = batch.solve_batch(c["states"], ix, un, c["iniPar"], c["thickness"], c["nx"], c["tSteps"], return_stats=True)
        dt = time.time() - t0
    print(f"variant {variant} B={B}: {dt*1e3:.1f} ms -> {B/dt:.0f} solves/s; bad {(status!=0).sum()}; steps {stats[:,0].mean():.0f}/{stats[:,0].max()} rhs {stats[:,1].mean():.0f}", flush=True)
