"""bench.py --workload pt : BASELINE.json configs[2]/[3] -- a parallel-tempering
ensemble on the staub 6-measurement power scan (lengths 311/2000 nm x 3 fluences,
nx=128, 1001 sample times, IRF convolution with a 189-tap IRF, ladder
T_k = 1000^(k/(n-1))), chains block-partitioned over the ranks (strong scaling),
one RCCL all-gather of the per-chain log-likelihood table per iteration.

One "step" = one Metropolis iteration of the whole ensemble: propose for every
chain, n_chains x 6 forward solves + IRF convolutions + likelihoods at all ladder
temperatures on the GPUs, all-gather, accept/reject, and (every temper_freq
iterations, SURVEY.md 8d: 10) the replica-exchange replay.
"""
import os
import time

import numpy as np

from . import _lib as L
from . import forward_solver, laplace, pt
from . import workloads as W
from .likelihood import LikelihoodEngine
from .sim_utils import Grid


def synthetic_irf(nk=189, dt=0.025, t0=-0.5):
    """A 189-row, 0.025 ns IRF like IRFs/irf_745nm.csv (which cannot travel to the GPU
    box): asymmetric pulse peaking at t=0 on a small positive background."""
    t = t0 + dt * np.arange(nk)
    y = 20.0 + 4000.0 * np.where(t < 0, np.exp(-(t / 0.06) ** 2), np.exp(-t / 0.35) * 0.7 + 0.3 * np.exp(-(t / 0.1) ** 2))
    return np.stack([t, y], axis=1)


def synthetic_data(irf_tables, nx, times):
    """the six traces at the un-randomised initial guesses,
    convolved, as log10 values with 0.02 uncertainty."""
    ix, un, s0 = W.indexes(), W.units(), W.initial_state()
    vals, uncs = [], []
    for m in range(6):
        g = Grid(W.STAUB_LENGTHS[m], nx, times, 4.0)
        sig = forward_solver.solve(np.array(W.STAUB_FLUENCES[m]), g, s0.copy(), ix, meas="TRPL", units=un,
                                   ini_mode="fluence")
        ct, cy, ok = laplace.do_irf_convolution(times, sig, irf_tables[745], time_max_shift=True)
        y, tc, _, _ = laplace.post_conv_trim(ct, cy, times, np.zeros_like(times), np.zeros_like(times))
        v = np.full(times.shape[0], np.log10(np.abs(y[-1])))
        v[:len(y)] = np.log10(np.abs(np.where(np.isnan(y), y[-1], y)))
        vals.append(v)
        uncs.append(np.full(times.shape[0], 0.02))
    return vals, uncs


def run(a, rank, world, local, dev):
    import torch
    import torch.distributed as dist
    n_chains = a.chains
    nx = 128
    times = W.default_times()
    ctx = L.default_context(local)
    irf_tables = laplace.make_I_tables({745: synthetic_irf()})
    vals, uncs = synthetic_data(irf_tables, nx, times)
    eng = LikelihoodEngine(W.indexes(), W.units(), W.STAUB_LENGTHS, nx, ["TRPL"] * 6, [times] * 6, vals, uncs,
                           {"TRPL": 1.0}, np.array(W.STAUB_FLUENCES), ini_mode="fluence", irf_tables=irf_tables,
                           irf_convolution=[745] * 6, device=local, ctx=ctx,
                           sort_by_cost=os.environ.get("MTRPL_NO_SORT") != "1",
                           parts=int(os.environ.get("MTRPL_PARTS", "2")))
    ens0 = W.pt_ensemble(n_chains, nx)
    T = ens0["temperatures"]
    d_T = torch.from_numpy(np.ascontiguousarray(T)).to(dev)

    def evaluator(states_block, temps):
        eng.upload(states_block)
        ll = eng.launch(d_T, len(temps))
        return ll.view(-1, eng.M, len(temps)).sum(dim=1)

    names = W.PARAM_NAMES
    ix = W.indexes()
    active = [n in W.PRIOR_BOX for n in names]
    prior = {n: W.PRIOR_BOX.get(n, (0, np.inf)) for n in names}
    do_log = np.ones(len(names), dtype=bool)
    base_move = np.where(active, 0.003, 0.0)  # MCMC_script_writer.py:116

    def proposer(states, temps, rng):
        return pt.propose(states, temps, base_move, names, active, prior, do_log, ix, rng)

    ens = pt.PTEnsemble(ens0["chain_states"], T, evaluator, rank=rank, world=world)
    ens.initialise()
    temper_freq = 10
    k = 0
    for _ in range(a.warmup):
        k += 1
        ens.iterate(proposer, k % temper_freq == 0)
    torch.cuda.synchronize()
    if getattr(a, "use_dist", world > 1):
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_acc = 0
    for _ in range(a.steps):
        k += 1
        n_acc += int(ens.iterate(proposer, k % temper_freq == 0).sum())
    torch.cuda.synchronize()
    if getattr(a, "use_dist", world > 1):
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if getattr(a, "use_dist", world > 1):
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    stats = eng.d_stats.cpu().numpy().reshape(-1, 8)
    return {
        "metric": "mcmc_proposals_per_sec", "value": n_chains * a.steps / dt, "unit": "proposals/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"config3/4: {n_chains} PT chains x 6-measurement staub power scan, nx=128, 1001 times, "
                               "IRF convolution (189 taps, dt 0.025 ns) + likelihood at all ladder temperatures on "
                               "device, all-gather of the ll table every iteration, swaps every 10",
                   "chains": n_chains, "solves_per_step": n_chains * 6, "parallelism": f"chain-blocks x{world}",
                   "forward_solves_per_sec": n_chains * 6 * a.steps / dt, "accepted_moves": n_acc,
                   "swap_accept_rate": float(ens.swap_accepts.sum() / max(1, ens.swap_attempts.sum())),
                   "mean_integrator_steps_rank0": float(stats[:, 0].mean())},
    }
