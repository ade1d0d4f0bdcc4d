"""dev: the SPIKE backend against the team backend, solve by solve: status, steps, rejections, Newton failures, agreement"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import _lib as L, batch, workloads as W
ctx = L.default_context()
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, float(v))
ix, un = W.indexes(), W.units()
c = W.config5(4)
for nx in (512, 256, 384, 200):
    for t_end, n in ((1e-3, 3), (1.0, 11), (100.0, 51), (2000.0, 1001)):
        t = np.linspace(0, t_end, n)
        res = {}
        for variant in (0, 1):
            ctx.set_option("kernel_variant", variant)
            sig, st, stats = batch.solve_batch(c["states"][:8], ix, un, c["iniPar"][:8], c["thickness"][:8], nx, t, meas=c["meas"][:8], return_stats=True)
            res[variant] = (sig, st, stats)
        ctx.set_option("kernel_variant", 0)
        a, b = res[0], res[1]
        with np.errstate(all="ignore"):
            d = np.max(np.abs(a[0] - b[0]) / np.abs(b[0]), axis=1)
        print(f"nx {nx} t_end {t_end:g}: spike status {a[1].tolist()} steps {a[2][:,0].tolist()} rej {a[2][:,4].tolist()} nfail {a[2][:,5].tolist()} rhs/step {a[2][:,1].sum()/max(1,a[2][:,0].sum()):.2f}")
        print(f"                  team  status {b[1].tolist()} steps {b[2][:,0].tolist()} rej {b[2][:,4].tolist()} nfail {b[2][:,5].tolist()} rhs/step {b[2][:,1].sum()/max(1,b[2][:,0].sum()):.2f} | max rel diff {np.array2string(d, precision=1)}", flush=True)
