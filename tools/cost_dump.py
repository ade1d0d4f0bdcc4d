"""dev: states + measured residency / steps of the config-2 draws (and the config-3 ensemble) -> gpurun_out/cost_dump.npz,
to fit the a-priori cost model the launcher orders the solves of a batch by"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import _lib as L, batch, workloads as W
c = W.config2(4096)
ix, un = W.indexes(), W.units()
out = {}
for name, (st, ini, th) in {"cfg2": (c["states"], c["iniPar"], np.full(4096, 2000.0))}.items():
    s, status, stats = batch.solve_batch(st, ix, un, ini, th, 128, c["tSteps"], return_stats=True)
    out[name + "_states"], out[name + "_ini"], out[name + "_th"], out[name + "_stats"] = st, ini, th, stats
e = W.pt_ensemble(256)
s, status, stats = batch.solve_batch(e["states"], ix, un, e["iniPar"], e["thickness"], 128, e["tSteps"], return_stats=True)
out["pt_states"], out["pt_ini"], out["pt_th"], out["pt_stats"] = e["states"], e["iniPar"], e["thickness"], stats
np.savez_compressed("gpurun_out/cost_dump.npz", **out)
print("ok", stats[:, 0].mean())
