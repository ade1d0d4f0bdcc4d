"""Repeatability soak (dev tool): the config-2 batch through the one- and the two-per-SIMD wave kernels, the config-5 batch at
nx = 512 and 256 through the SPIKE kernels and at nx = 512 through the team kernel, N times each -- every run must return
the bytes, statuses and step counts of the first.    python tools/soak.py [N=10]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import _lib as L, batch, workloads as W  # noqa: E402


def run(name, n, fn):
    first, bad = None, 0
    for k in range(n):
        sig, status, stats = fn()
        cur = (sig.tobytes(), status.tobytes(), stats[:, :7].tobytes())
        if first is None:
            first = cur
            info = "status %s, steps mean %.1f max %d, Newton failures max %d" % (dict(zip(*np.unique(status, return_counts=True))),
                                                                               stats[:, 0].mean(), stats[:, 0].max(), stats[:, 5].max())
        elif cur != first:
            bad += 1
    print(f"{name:34s} {n} runs, {bad} differ from the first | {info}", flush=True)
    return bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    ctx = L.default_context()
    ix, un = W.indexes(), W.units()
    c2 = W.config2(2048)
    t = c2["tSteps"]

    def cfg2(B, variant):
        def f():
            ctx.set_option("kernel_variant", variant)
            try:
                return batch.solve_batch(c2["states"][:B], ix, un, c2["iniPar"][:B], c2["thickness"], c2["nx"], t, hmax=c2["hmax"],
                                         return_stats=True)
            finally:
                ctx.set_option("kernel_variant", 0)
        return f

    def cfg5(nx, variant):
        c = W.config5(128, nx)

        def f():
            ctx.set_option("kernel_variant", variant)
            try:
                return batch.solve_batch(c["states"], ix, un, c["iniPar"], c["thickness"], c["nx"], c["tSteps"], hmax=c["hmax"],
                                         meas=c["meas"], model=c["model"], ini_mode=c["ini_mode"], return_stats=True)
            finally:
                ctx.set_option("kernel_variant", 0)
        return f
    bad = 0
    bad += run("config 2, 1024 solves, one per SIMD", n, cfg2(1024, 6))
    bad += run("config 2, 2048 solves, two per SIMD", n, cfg2(2048, 3))
    bad += run("config 2, 1024 solves, fp64 blocks", n, cfg2(1024, 2))
    bad += run("config 5, nx 512, SPIKE W = 4", n, cfg5(512, 0))
    bad += run("config 5, nx 256, SPIKE W = 2", n, cfg5(256, 0))
    bad += run("config 5, nx 512, team kernel", max(2, n // 3), cfg5(512, 1))
    print("soak", "OK" if bad == 0 else f"FAILED ({bad})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
