"""BASELINE.json configs[4] / SURVEY.md 8d config 5: fp64 vs fp32 tolerance sweep at nx=512 (thick +
thin film, TRPL + TRTS).  For every (precision, internal rtol) the 64-solve batch is integrated on the
GPU and compared with the converged oracle (tests/census/config5_refs.py), per decade of signal decay.
    python tools/config5_sweep.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metrotrpl_amd import workloads as W, batch, _lib as L

z = np.load(os.path.join(ROOT, "tools", "_cache", "config5_refs.npz"))
refs, NS = z["refs"], int(z["n_states"])
c = W.config5(NS)
ix, un = W.indexes(), W.units()
ctx = L.default_context()
defaults = {k: ctx.get_option(k) for k in ("rtol_force", "rtol_cap", "rtol_floor", "atol_rel", "atol_rel_np", "max_steps")}
ctx.set_option("max_steps", 100000)  # bounds a non-converging fp32 run: it stops early and is counted as failed
meas = np.array(c["meas"])
rows = []
for precision, atol_rel in ((64, 1e-4), (32, 1e-4), (32, 1e-2), (64, None)):
    ctx.set_option("precision", precision)
    ctx.set_option("atol_rel", atol_rel if atol_rel is not None else defaults["atol_rel"])
    # uniform-tolerance rows: one absolute floor for densities and field alike (the shipped policy resolves the
    # densities 8 decades down, which fp32 cannot follow at all)
    ctx.set_option("atol_rel_np", atol_rel if atol_rel is not None else defaults["atol_rel_np"])
    for rtol in ((1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 3e-9) if atol_rel == 1e-4 and precision == 64 else
                 (1e-3, 1e-4, 1e-5, 1e-6) if precision == 32 else ("shipped policy",)):
        if isinstance(rtol, str):  # the product's own tolerance policy (per-solve rtol from meas type and tauP/tauN)
            for k, v in defaults.items():
                ctx.set_option(k, v)
            atol_rel = defaults["atol_rel"]
        else:
            ctx.set_option("rtol_force", rtol)
        t0 = time.time()
        sig, status, stats = batch.solve_batch(c["states"], ix, un, c["iniPar"], c["thickness"], c["nx"], c["tSteps"],
                                               meas=c["meas"], return_stats=True)
        dt = time.time() - t0
        ok = status == 0
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.abs(sig - refs) / np.abs(refs)
            # decade of decay of each sample: 0 = within the first decade below the initial value
            dec = np.floor(np.log10(np.abs(refs[:, :1])) - np.log10(np.maximum(np.abs(refs), 1e-300))).astype(int)
        per_dec = {}
        for d in range(6):
            m = (dec == d) & ok[:, None] & np.isfinite(rel)
            if m.any():
                per_dec[d] = (float(rel[m].max()), float(np.median(rel[m])))
        row = {"precision": precision, "rtol_int": rtol, "atol_rel": atol_rel, "failed": int((~ok).sum()), "mean_steps": float(stats[ok, 0].mean()) if ok.any() else None,
               "ms_per_batch": dt * 1e3, "per_decade_max_median": per_dec,
               "max_4dec_TRPL": float(max((rel[k][(dec[k] < 4) & np.isfinite(rel[k])].max() for k in range(len(rel)) if ok[k] and meas[k] == "TRPL"), default=np.nan)),
               "max_4dec_TRTS": float(max((rel[k][(dec[k] < 4) & np.isfinite(rel[k])].max() for k in range(len(rel)) if ok[k] and meas[k] == "TRTS"), default=np.nan))}
        rows.append(row)
        print(f"fp{precision} rtol={rtol} atol_rel={atol_rel:g}: failed {row['failed']}/{len(ok)} steps {row['mean_steps']} {dt*1e3:.0f} ms | "
              + " ".join(f"d{d}:{v[0]:.1e}/{v[1]:.1e}" for d, v in per_dec.items())
              + f" | 4dec TRPL {row['max_4dec_TRPL']:.1e} TRTS {row['max_4dec_TRTS']:.1e}", flush=True)
ctx.set_option("precision", 64)
for k, v in defaults.items():
    ctx.set_option(k, v)
if len(sys.argv) > 1:
    json.dump({"workload": "config5: 16 seeded states x (2000 nm, 311 nm) x (TRPL, TRTS), nx=512, 1001 samples over 2000 ns",
               "reference": "oracle (SciPy LSODA + gcc RHS) at RTOL=1e-12 / ATOL=1e-17", "rows": rows}, open(sys.argv[1], "w"), indent=1)
