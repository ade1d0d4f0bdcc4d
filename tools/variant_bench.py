"""Compare the solver backends on the config-2 workload: launch time at several batch sizes (HIP events),
integrator step counts, per-solve residency, and agreement of the outputs with the default backend.
usage: python tools/variant_bench.py [--variants 6,3,2] [--batches 1024,384,64] [--reps 5]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="6,3,2")
    ap.add_argument("--batches", default="1024,512,384,64")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--nx", type=int, default=128)
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    import torch
    from metrotrpl_amd import _lib as L, batch, workloads as W
    dev = torch.device("cuda", 0)
    ctx = L.Context(0)
    for kv in a.opt:
        k, v = kv.split("=")
        ctx.set_option(k, float(v))
    variants = [int(v) for v in a.variants.split(",")]
    batches = [int(b) for b in a.batches.split(",")]
    c = W.config2(max(batches))
    t_host = np.ascontiguousarray(c["tSteps"])
    nt = t_host.shape[0]
    nx = a.nx
    d_t = torch.from_numpy(t_host).to(dev)
    ref = {}
    rows = []
    for B in batches:
        descs, ini = batch.build_batch(c["states"][:B], W.indexes(), W.units(), c["iniPar"][:B], c["thickness"], nx, nt,
                                       hmax=c["hmax"], meas=c["meas"], model=c["model"], ini_mode=c["ini_mode"])
        d_descs = torch.from_numpy(descs.view(np.uint8).reshape(-1)).to(dev)
        d_ini = torch.from_numpy(ini).to(dev)
        d_out = torch.zeros(B * nt, dtype=torch.float64, device=dev)
        d_status = torch.zeros(B, dtype=torch.int32, device=dev)
        d_stats = torch.zeros(B * 8, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream()
        for v in variants:
            ctx.set_option("kernel_variant", v)

            def step():
                ctx.check(L.lib().mtrpl_solve_batch_dev(ctx.handle, B, d_descs.data_ptr(), d_ini.data_ptr(), d_t.data_ptr(),
                                                        d_out.data_ptr(), d_status.data_ptr(), d_stats.data_ptr(), nx,
                                                        L.MODEL[c["model"]], s.cuda_stream))
            step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(a.reps):
                step()
            e1.record(s)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            sig = d_out.cpu().numpy().reshape(B, nt)
            st = d_status.cpu().numpy()
            stats = d_stats.cpu().numpy().reshape(B, 8)
            if B not in ref:
                ref[B] = sig.copy()
            m = np.abs(ref[B]) >= 1e-4 * np.abs(ref[B][:, :1])
            dev_ = float(np.max(np.where(m, np.abs(sig - ref[B]) / np.abs(ref[B]), 0.0)))
            row = {"variant": v, "B": B, "ms": ms, "solves_per_s": B / ms * 1e3, "ok": bool(np.all(st == 0)),
                   "steps_mean": float(stats[:, 0].mean()), "steps_max": int(stats[:, 0].max()),
                   "resid_mean_ticks": float(stats[:, 7].mean()) * 16, "resid_max_ticks": float(stats[:, 7].max()) * 16,
                   "max_rel_dev_vs_first_variant": dev_}
            rows.append(row)
            print(json.dumps(row), flush=True)
    ctx.set_option("kernel_variant", 0)


if __name__ == "__main__":
    main()
