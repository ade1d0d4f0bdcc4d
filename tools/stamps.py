"""Phase breakdown of a solve from the -DMTRPL_STAMPS diagnostic build (dev tool; the product library has
no stamps).  Builds metrotrpl_amd/lib/libmetrotrpl_amd_stamps.so if needed.
usage: python tools/stamps.py [--variant 6,2] [--batch 1024]   (6: the default wave kernel, 2: its fp64-block form)"""
import argparse
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "metrotrpl_amd", "lib", "libmetrotrpl_amd_stamps.so")
CSRC = os.path.join(REPO, "metrotrpl_amd", "csrc")
_newest = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC))
if not os.path.exists(LIB) or os.path.getmtime(LIB) < _newest or os.environ.get("MTRPL_STAMPS_REBUILD"):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DMTRPL_STAMPS", "-o", LIB, os.path.join(REPO, "metrotrpl_amd", "csrc", "mtrpl_capi.hip")])
os.environ["MTRPL_LIB_PATH"] = LIB
import numpy as np  # noqa: E402
sys.path.insert(0, REPO)
from metrotrpl_amd import _lib as L, batch, workloads as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="6,2")
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--config5", action="store_true", help="the nx = 512 batch of bench.py --workload config5 (variant 0: SPIKE, 1: team)")
a = ap.parse_args()
ix, un = W.indexes(), W.units()
ctx = L.default_context()
if a.config5:
    c = W.config5(max(1, a.batch // 4))
    names = ["steps", "controller", "factorise", "rhs", "re-solve", "reduce+publish", "newton glue", "dense out"]
    for variant in [int(v) for v in a.variant.split(",")]:
        ctx.set_option("kernel_variant", variant)
        for rep in range(2):
            sig, status, st = batch.solve_batch(c["states"], ix, un, c["iniPar"], c["thickness"], c["nx"], c["tSteps"], hmax=c["hmax"],
                                                meas=c["meas"], model=c["model"], ini_mode=c["ini_mode"], return_stats=True)
        tot = st[:, 1:].sum(axis=1).astype(float)
        print(f"config 5, variant {variant}, B {st.shape[0]}: mean steps {st[:, 0].mean():.0f}, mean total {tot.mean():.0f} kcycles (wave 0 of "
              f"each solve), per step {1024 * tot.mean() / st[:, 0].mean():.0f} cycles")
        for k in range(1, 8):
            print(f"  {names[k]:15s}: {st[:, k].mean():9.0f} kcyc  {100 * st[:, k].sum() / tot.sum():5.1f}%   per step "
                  f"{1024 * st[:, k].mean() / st[:, 0].mean():7.0f} cyc")
    ctx.set_option("kernel_variant", 0)
    sys.exit(0)
c = W.config2(a.batch)
names = ["steps", "controller", "factorise", "rhs", "re-solve", "reduce+publish", "newton glue", "dense out"]
for variant in [int(v) for v in a.variant.split(",")]:
    ctx.set_option("kernel_variant", variant)
    for rep in range(2):
        sig, status, st = batch.solve_batch(c["states"], ix, un, c["iniPar"], c["thickness"], c["nx"], c["tSteps"],
                                            return_stats=True)
    tot = st[:, 1:].sum(axis=1).astype(float)
    print(f"variant {variant} B {a.batch}: mean steps {st[:, 0].mean():.0f}, mean total {tot.mean():.0f} kcycles, max {tot.max():.0f}; "
          f"per step {1024 * tot.mean() / st[:, 0].mean():.0f} cycles")
    for k in range(1, 8):
        print(f"  {names[k]:15s}: {st[:, k].mean():9.0f} kcyc  {100 * st[:, k].sum() / tot.sum():5.1f}%   per step "
              f"{1024 * st[:, k].mean() / st[:, 0].mean():7.0f} cyc")
ctx.set_option("kernel_variant", 0)
