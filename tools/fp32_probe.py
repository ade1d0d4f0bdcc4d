"""fp32 build probe (dev tool): status / statistics of a few golden cases at several tolerances."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metrotrpl_amd import batch, _lib as L
gold = np.load(os.path.join(ROOT, "tests", "golden", "solve.npz"))
ctx = L.default_context()
ctx.set_option("max_steps", 100000)
for prec in (32, 64):
    ctx.set_option("precision", prec)
    ctx.set_option("kernel_variant", 1)
    for key in ("cfg2_0__tight", "cfg3_m0__tight", "cfg5_thick_pl__tight", "cfg5_thin_trts__tight"):
        names, meas, solver, model, ini_mode = [str(x) for x in gold[key + "__meta"]]
        th, nx, hmax, _, _, min_y = gold[key + "__geom"]
        idx = {n: i for i, n in enumerate(names.split(","))}
        for rtol in (1e-3, 1e-4, 1e-5, 1e-6):
            ctx.set_option("rtol_force", rtol)
            for aw in (1e-4, 1e-2):
                ctx.set_option("atol_rel", aw)
                sig, status, stats = batch.solve_batch(gold[key + "__state"][None, :], idx, gold[key + "__units"], gold[key + "__iniPar"][None, :],
                                                       float(th), int(nx), gold[key + "__tSteps"], hmax=float(hmax), meas=meas, return_stats=True)
                ref = gold[key + "__sig"]
                m = np.abs(ref) >= 1e-2 * np.abs(ref[0])
                nz = np.nonzero(sig[0] == 0)[0]
                err = np.max(np.abs(sig[0][m] - ref[m]) / np.abs(ref[m]))
                print(f"fp{prec} {key} rtol={rtol:g} atol_rel={aw:g}: status={status[0]} stats={stats[0].tolist()} err2dec={err:.2e} first_zero={nz[0] if len(nz) else -1}", flush=True)
