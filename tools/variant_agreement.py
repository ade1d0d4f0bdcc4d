"""Agreement of the default wave kernel (fp32-stored elimination blocks, packed-fp32 re-solve; kernel_variant 3: its
two-per-SIMD build) with the fp64-block wave kernel (2) over many config-2 draws: worst relative difference per
decade of signal decay (dev tool).
usage: python tools/variant_agreement.py [n_solves]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import _lib as L, batch, workloads as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
c = W.config2(n)
ctx = L.default_context()
out = {}
for v in (2, 3):
    ctx.set_option("kernel_variant", v)
    sig, status, stats = batch.solve_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], c["nx"],
                                           c["tSteps"], return_stats=True)
    out[v] = sig
    print(f"variant {v}: {int((status == 0).sum())}/{n} ok, steps mean {stats[:, 0].mean():.0f} max {stats[:, 0].max()}, "
          f"rejected {stats[:, 4].mean():.1f}, newton failures {stats[:, 5].mean():.2f}")
ctx.set_option("kernel_variant", 0)
a, b = out[2], out[3]
dec = np.floor(np.log10(a[:, :1]) - np.log10(np.maximum(a, 1e-300))).astype(int)
rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-300)
for d in range(12):
    m = dec == d
    if m.any():
        print(f"  decade {d:2d}: n={int(m.sum()):8d}  max {rel[m].max():.2e}  p99.9 {np.quantile(rel[m], 0.999):.2e}  median {np.median(rel[m]):.2e}")
