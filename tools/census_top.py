"""Which draws are the worst of the census, and how do they correlate with the lifetime ratio (dev tool)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metrotrpl_amd import workloads as W, batch, _lib as L
ix, un = W.indexes(), W.units()
ctx = L.default_context()
for MEAS in ("TRPL", "TRTS"):
    z = np.load(os.path.join(ROOT, "tools", "_cache", "census_refs.npz" if MEAS == "TRPL" else "census_refs_trts.npz"))
    N, NCH = int(z["n"]), int(z["nch"])
    c = W.config2(N); e = W.pt_ensemble(NCH)
    for name, st, ini, th, refs in (("cfg2", c["states"], c["iniPar"], np.full(N, 2000.0), z["cfg2"]), ("pt", e["states"], e["iniPar"], e["thickness"], z["pt"])):
        sig, status, stats = batch.solve_batch(st, ix, un, ini, th, 128, c["tSteps"], meas=MEAS, return_stats=True)
        errs = np.zeros(len(refs))
        for k in range(len(refs)):
            m = np.abs(refs[k]) >= 1e-4 * abs(refs[k][0])
            errs[k] = np.max(np.abs(sig[k][m] - refs[k][m]) / np.abs(refs[k][m]))
        tn, tp = st[:, ix["tauN"]], st[:, ix["tauP"]]
        A = (tn + tp) / np.minimum(tn, tp)
        An = (tn + tp) / tn   # p-type: low-injection lifetime is tauN
        order = np.argsort(-errs)[:8]
        print(MEAS, name, "corr(log err, log A) = %.2f, with (tn+tp)/tn: %.2f" % (np.corrcoef(np.log(errs), np.log(A))[0, 1], np.corrcoef(np.log(errs), np.log(An))[0, 1]))
        for k in order:
            print(f"   #{k}: err {errs[k]:.2e} A={A[k]:.1f} (tn+tp)/tn={An[k]:.1f} tauN={tn[k]:.3g} tauP={tp[k]:.3g} p0={st[k, ix['p0']]:.2g} L={th[k]:.0f} flu={ini[k][0]:.2g} steps={stats[k,0]}")
