"""Per-solve residency inside one 1024-solve launch: how far is the slowest solve above the mean (dev tool)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metrotrpl_amd import workloads as W, batch, _lib as L
c = W.config2(1024); ix, un = W.indexes(), W.units()
ctx = L.default_context()
for rep in range(2):
    t0 = time.time()
    sig, st, stats = batch.solve_batch(c["states"], ix, un, c["iniPar"], c["thickness"], 128, c["tSteps"], return_stats=True)
    dt = time.time() - t0
tk = stats[:, 7].astype(float)
print(f"host {dt*1e3:.1f} ms; residency ticks: mean {tk.mean():.0f} max {tk.max():.0f} p99 {np.percentile(tk,99):.0f} p90 {np.percentile(tk,90):.0f} min {tk.min():.0f}; max/mean {tk.max()/tk.mean():.2f}")
print("corr(ticks, steps) %.3f corr(ticks, rhs) %.3f" % (np.corrcoef(tk, stats[:,0])[0,1], np.corrcoef(tk, stats[:,1])[0,1]))
o = np.argsort(-tk)[:10]
for k in o:
    s = c["states"][k]
    print(f"  #{k}: ticks {tk[k]:.0f} steps {stats[k,0]} rhs {stats[k,1]} fac {stats[k,2]} rej {stats[k,4]} tauN {s[ix['tauN']]:.3g} tauP {s[ix['tauP']]:.3g} mu_n {s[ix['mu_n']]:.3g} mu_p {s[ix['mu_p']]:.3g} p0 {s[ix['p0']]:.2g} Sf {s[ix['Sf']]:.3g} flu {c['iniPar'][k][0]:.2g}")
# what would perfect backfilling give?
print("sum(ticks)/1024 SIMDs vs max(ticks): %.0f vs %.0f -> a launch-bound batch wastes %.0f%% of the SIMD time" % (tk.sum()/1024, tk.max(), 100*(1-tk.mean()/tk.max())))
