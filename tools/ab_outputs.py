"""Outputs of two builds of the library on the config-2 batch, solve by solve (dev tool).
    python tools/ab_outputs.py metrotrpl_amd/lib/libA.so metrotrpl_amd/lib/libB.so [variant]
Each library runs in its own child process (one HIP runtime, one library per process)."""
import os
import subprocess
import sys
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) >= 2 and sys.argv[1] == "--child":
    sys.path.insert(0, REPO)
    from metrotrpl_amd import _lib as L, batch, workloads as W
    c = W.config2(1024)
    ctx = L.default_context()
    ctx.set_option("kernel_variant", int(sys.argv[3]))
    sig, status, st = batch.solve_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], c["nx"], c["tSteps"],
                                        return_stats=True)
    np.savez(sys.argv[2], sig=sig, status=status, st=st)
    sys.exit(0)
a, b = sys.argv[1], sys.argv[2]
variant = sys.argv[3] if len(sys.argv) > 3 else "0"
res = []
for k, lib in enumerate((a, b)):
    out = os.path.join(REPO, "gpurun_out", f"ab_{k}.npz")
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", out, variant],
                          env=dict(os.environ, MTRPL_LIB_PATH=os.path.join(REPO, lib)))
    res.append(np.load(out))
A, B = res
same = np.all(A["sig"] == B["sig"], axis=1)
print(f"{int(same.sum())} of {len(same)} solves bitwise identical; steps {A['st'][:, 0].mean():.2f} vs {B['st'][:, 0].mean():.2f}")
for i in np.flatnonzero(~same)[:12]:
    d = np.flatnonzero(A["sig"][i] != B["sig"][i])
    rel = np.abs(A["sig"][i][d] - B["sig"][i][d]) / np.abs(A["sig"][i][d])
    print(f"  solve {i}: first differing sample {d[0]} of {len(d)}, value {A['sig'][i][d[0]]:.6e} vs {B['sig'][i][d[0]]:.6e}, "
          f"max rel {rel.max():.2e}, steps {A['st'][i, 0]} vs {B['st'][i, 0]}; decades down at first diff "
          f"{np.log10(A['sig'][i][0] / A['sig'][i][d[0]]):.2f}")
