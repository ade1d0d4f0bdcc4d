"""tools/heldout_both.py ["k=v,k2=v2" ...] -- the nx <= 128 held-out sets (the frozen one and round 5's second and third draws) through
the current build (or MTRPL_LIB_PATH), per option setting: steps and the worst error minus u_oracle per decade 0-8."""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from metrotrpl_amd import _lib as L  # noqa: E402
from test_gpu_heldout import decade_table, solve_all  # noqa: E402

ctx = L.default_context()
for setting in ["shipped"] + sys.argv[1:]:
    saved = {}
    if setting != "shipped":
        for kv in setting.split(","):
            k, val = kv.split("=")
            saved[k] = ctx.get_option(k)
            ctx.set_option(k, float(val))
    for fn in ("heldout.npz", "heldout2.npz", "heldout3.npz"):
        z = np.load(os.path.join("tests", "golden", fn))
        sig, status, steps = solve_all(z)
        v = z["valid"]
        tab = decade_table(z, sig)
        print(f"{setting:34s} {fn:13s} steps {steps[v].mean():5.0f}/{steps[v].max():5d} bad {int((status[v] != 0).sum())} | err-u decades 0..8: "
              + " ".join("%.1e" % ex for (_, _, ex, _) in tab[:9]) + " | worst " + ",".join(str(w) for (_, _, _, w) in tab[:9]), flush=True)
    for k, val in saved.items():
        ctx.set_option(k, val)
