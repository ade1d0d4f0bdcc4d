"""Integrator statistics of every solve of the config-2 batch, with its parameters (dev tool): which
solves are the slow ones, and why.  Writes gpurun_out/step_census.npz.
usage: python tools/step_census.py [n] [key=value ...]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import _lib as L, batch, workloads as W
n = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else 1024
ctx = L.default_context()
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        ctx.set_option(k, float(v))
c = W.config2(n)
sig, status, stats = batch.solve_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], c["nx"], c["tSteps"],
                                       return_stats=True, ctx=ctx)
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/step_census.npz", states=c["states"], iniPar=c["iniPar"], stats=stats, status=status, sig0=sig[:, 0],
         sig_end=sig[:, -1], names=np.array(",".join(W.PARAM_NAMES)))
st = stats[:, 0]
print("steps mean %.0f median %.0f p90 %.0f p99 %.0f max %d; rej mean %.1f max %d; fac mean %.0f; nfail mean %.2f" % (
    st.mean(), np.median(st), np.percentile(st, 90), np.percentile(st, 99), st.max(), stats[:, 4].mean(), stats[:, 4].max(),
    stats[:, 2].mean(), stats[:, 5].mean()))
ix = W.indexes()
order = np.argsort(-st)[:12]
for k in order:
    s = c["states"][k]
    print(f"#{k}: steps {st[k]} rhs {stats[k,1]} fac {stats[k,2]} rej {stats[k,4]} nfail {stats[k,5]} | tauN {s[ix['tauN']]:.1f} tauP {s[ix['tauP']]:.0f} "
          f"Sf {s[ix['Sf']]:.1f} Sb {s[ix['Sb']]:.1f} mu_n {s[ix['mu_n']]:.1f} mu_p {s[ix['mu_p']]:.1f} ks {s[ix['ks']]:.1e} p0 {s[ix['p0']]:.1e} "
          f"flu {c['iniPar'][k,0]:.1e} decades {np.log10(sig[k,0]/max(sig[k,-1],1e-300)):.1f}")
