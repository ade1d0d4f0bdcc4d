"""Child process of tests/test_gpu_ensemble.py: the config-3 parallel-tempering ensemble (64 chains x the
6-measurement staub power scan, nx=128, 1001 sample times, IRF 745 nm) for a few iterations on cuda:0,
optionally inside a one-rank `nccl` process group that is initialised BEFORE anything touches the GPU, so
that the all-gather of the log-likelihood table (per iteration with stream-mode draws, per replica exchange with
counter-mode draws) really goes through RCCL.
Writes the walk to an .npz.  Not a test by itself."""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dist", type=int, default=0)
    ap.add_argument("--parts", type=int, default=1)
    ap.add_argument("--sort", type=int, default=0)
    ap.add_argument("--chains", type=int, default=64)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--temper-freq", type=int, default=5)
    ap.add_argument("--port", type=int, default=29617)
    ap.add_argument("--force-min-y", type=int, default=1)
    ap.add_argument("--rng-mode", default="stream")
    ap.add_argument("--prefetch", type=int, default=0)
    ap.add_argument("--groups", type=int, default=1)
    ap.add_argument("--kernel-variant", type=int, default=0)
    ap.add_argument("--device-decisions", type=int, default=0)
    ap.add_argument("--packed-order", default="chains")
    ap.add_argument("--dealt", type=int, default=0)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    if a.dist:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{a.port}", rank=0, world_size=1, device_id=dev)
    torch.cuda.set_device(0)
    from metrotrpl_amd import _lib as L, pt, pt_bench
    L.default_context(0).set_option("kernel_variant", a.kernel_variant)
    irf_raw = np.load(os.path.join(REPO, "tests", "golden", "irf.npz"))["irf745_raw"]
    S = pt_bench.setup(a.chains, dev, 0, irf_raw=irf_raw, parts=a.parts, sort_by_cost=bool(a.sort), split_min=0,
                       sort_min=0, force_min_y=bool(a.force_min_y), groups=a.groups, packed_order=a.packed_order,
                       dealt=a.groups - 1 if a.dealt else None)
    ens = pt.PTEnsemble(S["ens0"]["chain_states"], S["T"], S["evaluators"] if a.groups > 1 else S["evaluator"], rank=0,
                        world=1, rng_mode=a.rng_mode, streams=S["streams"], device_decisions=bool(a.device_decisions),
                        regroup_by_cost=bool(a.dealt))
    ens.initialise()
    table0 = ens.table.copy()
    n_parts = len(S["eng"]._cuts) - 1
    swaps = a.iters // a.temper_freq
    calls = []
    def spy(inner):
        def f(block, temps):
            calls.append(block.shape[0])
            return inner(block, temps)
        if a.device_decisions:  # what pt._DecidePipe looks for
            hooks = ["pack_rows", "launch_packed", "plan_packed", "row_cost"]
            if a.device_decisions == 1:  # 2: the torch-op form of the two steps instead of the library's kernels
                hooks += ["select", "decide"]
            for k in hooks:
                if hasattr(inner, k):
                    setattr(f, k, getattr(inner, k))
        return f
    ens.evaluators = [spy(ev) for ev in ens.evaluators]
    n_acc = ens.advance(S["proposer"], a.iters, a.temper_freq, prefetch=bool(a.prefetch))
    torch.cuda.synchronize()
    backend = dist.get_backend() if (a.dist and dist.is_initialized()) else "none"
    np.savez(a.out, states0=S["ens0"]["chain_states"], table0=table0, states=ens.states, logll=ens.logll, table=ens.table,
             swap_attempts=ens.swap_attempts, swap_accepts=ens.swap_accepts, T=S["T"], times=S["times"],
             vals=np.array(S["vals"]), uncs=np.array(S["uncs"]), n_parts=n_parts, swaps=swaps, backend=np.array(backend),
             ordered=np.array(S["eng"]._order is not None), calls=np.array(calls), n_acc=n_acc, n_collectives=ens.n_collectives,
             n_pipelined=ens.n_pipelined, pipe_own=np.concatenate([ens.pipe_own[g] for g in sorted(ens.pipe_own)] or [np.zeros(0)]),
             chain_cost=ens.chain_cost)
    if a.dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
