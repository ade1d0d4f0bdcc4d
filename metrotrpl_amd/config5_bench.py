"""bench.py --workload config5 : BASELINE.json configs[4] -- nx=512 thick- and thin-film drift-diffusion, mixed
TRPL + TRTS, fp64.  One "step" = one launch of `solve_kernel_team<double, 4, ...>` (four wavefronts per solve,
elimination blocks stored as fp32 in LDS) over a 512-solve batch: 128 seeded config-2 states x (2000 nm, 311 nm) x
(TRPL, TRTS), inputs resident in HBM.  A parity / tolerance-sweep configuration, not the headline metric: the
line exists so that the nx=512 kernel has a measured rate and a roofline of its own (missing 5)."""
import time

import numpy as np

from . import _lib as L
from . import batch
from . import workloads as W
from .pt_bench import HBM_PEAK_GBS, algorithmic_bytes


def run(a, rank, world, local, dev):
    import torch
    import torch.distributed as dist
    ctx = L.Context(local)
    for kv in getattr(a, "opt", []) or []:
        k, v = kv.split("=")
        ctx.set_option(k, float(v))
    n_states = max(1, a.batch // 8)  # default --batch 1024 -> 128 states -> 512 solves
    c = W.config5(n_states * world)
    B = 4 * n_states
    sl = slice(rank * B, (rank + 1) * B)
    t_host = np.ascontiguousarray(c["tSteps"])
    nt = t_host.shape[0]
    descs, ini = batch.build_batch(c["states"][sl], W.indexes(), W.units(), c["iniPar"][sl], c["thickness"][sl],
                                   c["nx"], nt, hmax=c["hmax"], meas=c["meas"][sl.start:sl.stop], model=c["model"],
                                   ini_mode=c["ini_mode"])
    d_descs = torch.from_numpy(descs.view(np.uint8).reshape(-1)).to(dev)
    d_ini = torch.from_numpy(ini).to(dev)
    d_t = torch.from_numpy(t_host).to(dev)
    d_out = torch.zeros(B * nt, dtype=torch.float64, device=dev)
    d_status = torch.zeros(B, dtype=torch.int32, device=dev)
    d_stats = torch.zeros(B * 8, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()

    def step():
        ctx.check(L.lib().mtrpl_solve_batch_dev(ctx.handle, B, d_descs.data_ptr(), d_ini.data_ptr(), d_t.data_ptr(),
                                                d_out.data_ptr(), d_status.data_ptr(), d_stats.data_ptr(), int(c["nx"]),
                                                L.MODEL[c["model"]], stream.cuda_stream))

    for _ in range(max(1, a.warmup)):
        step()
    use_dist = getattr(a, "use_dist", world > 1)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for k in range(a.steps):
        ev[k][0].record(stream)
        step()
        ev[k][1].record(stream)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
    status = d_status.cpu().numpy()
    stats = d_stats.cpu().numpy().reshape(B, 8)
    sig = d_out.cpu().numpy()
    A = algorithmic_bytes(int(c["nx"]), nt)
    achieved = A * B / (kern_ms * 1e-3) / 1e9
    return {
        "metric": "trpl_forward_solves_per_sec", "value": world * B * a.steps / dt, "unit": "solves/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "config5: nx=512 drift-diffusion, 512 solves per GPU and step = 128 seeded config-2 states x "
                               "(2000 nm, 311 nm) x (TRPL, TRTS), 1001 output times over 2000 ns, hmax=4, std model, fp64",
                   "solves_per_gpu_per_step": B, "nx": int(c["nx"]), "nt_out": int(nt), "parallelism": f"batch-shard x{world}",
                   "all_solves_ok": bool(np.all(status == 0) and np.all(np.isfinite(sig))),
                   "mean_integrator_steps": float(stats[:, 0].mean()), "max_integrator_steps": int(stats[:, 0].max()),
                   "streams": 1, "issue_order": "as drawn", **({"options": a.opt} if getattr(a, "opt", None) else {})},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "solve_kernel_team<double, 4, false, false, true>", "kernel_ms": kern_ms,
                     "algorithmic_bytes_per_solve": A,
                     "note": "state-streaming model of SURVEY.md 8d (24.6 MB per nx=512 solve); the state lives in registers "
                             "and the elimination blocks in LDS (135 KB per workgroup, one workgroup per CU), so HBM is not "
                             "the limiter: four waves per solve on 256 CUs = 256 solves resident, the rest of the batch "
                             "queues behind them"},
    }
