"""One config-5 batch (nx = 512, 512 solves) through the library MTRPL_LIB_PATH names (default: the shipped build): status
histogram, integrator statistics, and the solves that did not end with status 0.
usage: python tools/c5_status.py [opt=value ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import _lib as L, batch, workloads as W  # noqa: E402


def main():
    ctx = L.Context(0)
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        ctx.set_option(k, float(v))
    c = W.config5(128)
    sig, status, stats = batch.solve_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], c["nx"], c["tSteps"],
                                           hmax=c["hmax"], meas=c["meas"], model=c["model"], ini_mode=c["ini_mode"], return_stats=True,
                                           ctx=ctx)
    print("lib", os.environ.get("MTRPL_LIB_PATH", "shipped"), "| status", dict(zip(*np.unique(status, return_counts=True))),
          "| steps mean %.1f max %d | rhs %.1f | fac %.1f | rejected %.1f | Newton failures max %d" % (
              stats[:, 0].mean(), stats[:, 0].max(), stats[:, 1].mean(), stats[:, 2].mean(), stats[:, 4].mean(), stats[:, 5].max()),
          "| finite", bool(np.isfinite(sig).all()), "| checksum %.12e" % float(np.sum(np.log10(np.maximum(sig, 1e-300)))))
    for k in np.flatnonzero(status != 0)[:12]:
        print("  solve", k, "status", status[k], "stats", stats[k].tolist(), "meas", c["meas"][k], "thickness", c["thickness"][k])


if __name__ == "__main__":
    main()
