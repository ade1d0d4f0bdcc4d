"""Per-decade accuracy of the GPU solver against the converged oracle traces of tests/golden/tail_decades.npz:
for every decade of signal decay (0 = within the first decade below the first sample) the max / p99 / median
relative error over ALL samples (none is clamped: min_y = the smallest double), for the config-2 draws and for the
config-3 geometry, at the shipped policy and at tighter settings (which also tells how far down the oracle
itself is converged: two independent integrators agreeing).
usage: python tools/tail_table.py [--json out.json] [--opt key=value ...]"""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def per_decade(sig, ref, max_dec=12):
    with np.errstate(divide="ignore", invalid="ignore"):
        dec = np.floor(np.log10(np.abs(ref[:, :1])) - np.log10(np.maximum(np.abs(ref), 1e-300))).astype(int)
        rel = np.abs(sig - ref) / np.abs(ref)
    dec = np.clip(dec, 0, max_dec)
    out = {}
    for d in range(max_dec + 1):
        m = dec == d
        if m.any():
            out[d] = {"n": int(m.sum()), "max": float(rel[m].max()), "p99": float(np.percentile(rel[m], 99)),
                      "median": float(np.median(rel[m]))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("--settings", default="shipped;rtol_force=3e-10;rtol_force=3e-11,atol_rel=1e-7")
    a = ap.parse_args()
    from metrotrpl_amd import _lib as L, batch
    z = np.load(os.path.join(REPO, "tests", "golden", "tail_decades.npz"))
    idx = {n: i for i, n in enumerate(str(z["names"]).split(","))}
    n2 = int(z["n_cfg2"])
    ctx = L.default_context()
    res = {}
    for setting in a.settings.split(";"):
        saved = {}
        if setting != "shipped":
            for kv in setting.split(","):
                k, v = kv.split("=")
                saved[k] = ctx.get_option(k)
                ctx.set_option(k, float(v))
        sig, status, stats = batch.solve_batch(z["states"], idx, z["units"], z["iniPar"], z["thickness"], 128, z["tSteps"],
                                               return_stats=True)
        for k, v in saved.items():
            ctx.set_option(k, v)
        res[setting] = {"ok": int((status == 0).sum()), "n": len(status), "mean_steps": float(stats[:, 0].mean()),
                        "config2": per_decade(sig[:n2], z["sig"][:n2]), "config3": per_decade(sig[n2:], z["sig"][n2:])}
        print(f"== {setting}: {res[setting]['ok']}/{len(status)} ok, mean steps {stats[:, 0].mean():.0f}")
        for part in ("config2", "config3"):
            print(f"  {part}: decade: n, max, p99, median")
            for d, r in res[setting][part].items():
                print(f"    {d:2d}: {r['n']:6d}  {r['max']:.2e}  {r['p99']:.2e}  {r['median']:.2e}")
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
