"""Per-decade errors and step counts on the held-out converged traces (tests/golden/heldout.npz; --large: the nx > 128 set
tests/golden/heldout_large.npz, through the SPIKE backend, or with kernel_variant=1 through the team backend) for one or more
integrator-option settings (dev tool):  python tools/heldout_table.py [--large] "honor_hmax=0" "shape_factor=100,fun_weight=4" """
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metrotrpl_amd import _lib as L
from tests.test_gpu_heldout import solve_all, decade_table
LARGE = "--large" in sys.argv
if LARGE:
    sys.argv.remove("--large")
z = np.load(os.path.join(ROOT, "tests", "golden", "heldout_large.npz" if LARGE else "heldout.npz"))
ctx = L.default_context()
v = z["valid"]
for setting in ["shipped"] + sys.argv[1:]:
    saved = {}
    if setting != "shipped":
        for kv in setting.split(","):
            k, val = kv.split("=")
            saved[k] = ctx.get_option(k)
            ctx.set_option(k, float(val))
    sig, status, steps = solve_all(z)
    for k, val in saved.items():
        ctx.set_option(k, val)
    tab = decade_table(z, sig)
    bad = int((status[v] != 0).sum())
    print(f"{setting:40s} steps {steps[v].mean():6.0f}/{steps[v].max():6d} bad {bad} | err-u per decade 0..8: " +
          " ".join(f"{ex:.1e}" for (_, _, ex, _) in tab[:9]) + " | worst " + ",".join(f"{who}" for (_, _, _, who) in tab[:9]), flush=True)
    if setting == "shipped" or os.environ.get("GROUPS"):
        for g_ in sorted(set(str(x) for x in z["group"])):
            m = v & (z["group"] == g_)
            print(f"     {g_:8s} {int(m.sum()):3d} traces, steps mean {steps[m].mean():6.0f} max {steps[m].max():6d}")
        if LARGE:
            for nx in sorted(set(int(x) for x in z["nx"])):
                m = v & (z["nx"] == nx)
                print(f"     nx {nx:4d}  {int(m.sum()):3d} traces, steps mean {steps[m].mean():6.0f} max {steps[m].max():6d}")
