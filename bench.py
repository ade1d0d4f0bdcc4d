#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X forward-solver engine.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its N ranks itself, as a
                                                            child `python -m torch.distributed.run`, and relays the line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md 8d "Config 2"): a 1024-solve
synthetic batch per GPU, nx=128, thickness 2000 nm, 1001 output times over
2000 ns, hmax=4, std model, TRPL, fp64, reference default tolerances.  One
"step" = one launch of the batched solve kernel over that batch with every input
(descriptors, time grid) already resident in HBM.  Ranks draw different states
(weak scaling, no data-path collective: solves are independent).

The K timed steps are independent batches, so they are issued the way a caller with
more than one batch to do issues them: round-robin on four HIP streams (with the runtime's
GPU_MAX_HW_QUEUES raised from 4 to 16 by the package, so that no two of them share a hardware queue), on the
two-solves-per-SIMD form of the wave kernel, every batch handed over AS DRAWN and issued in
that order -- a solve holds its SIMD slot for its whole duration and durations spread 1:1.5,
so the slots the short solves of one launch free go to the next launch.  (Round 3's issue
ordering by a cost model -- two small kernels in front of every launch -- bought 1.6 % and
is gone.)  Measured in the same run and reported beside it:
  single_stream            the same K steps back to back on ONE stream, one solve per
                           SIMD: every launch alone on the device (latency, the kernel's own speed)
  sorted_by_measured_cost  ... and with each batch's solves issued longest first by the cost
                           MEASURED on the identical batch in a warm-up launch -- knowledge a
                           first-time caller does not have: an upper bound of what ordering can
                           buy, not the headline
  fp64_linear_algebra      single_stream on kernel_variant 2 (fp64 blocks, fp64 re-solve); .streams: the timed
                           region itself on that kernel
  one_launch               one launch of 2048 / 4096 solves at a time: what a caller without streams gets
  pt                       BASELINE's second metric: MCMC proposals/s of the config-3 (64 chains)
                           and config-4-on-one-GPU (512 chains) parallel-tempering ensembles

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# one BLAS thread per process: the CPU baseline runs one solve per worker process, and LSODA's
# dense LU must not fan out over every host core from each of them
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes_per_solve(nx, nt, model="std", irf_nk=0):
    """A = nt_out * 8 * (2S + 1), S = 3nx+1 (std) / 4nx+1 (traps)."""
    from metrotrpl_amd.pt_bench import algorithmic_bytes
    return algorithmic_bytes(nx, nt, model, irf_nk)


def accuracy_object(ctx):
    """Per-decade worst relative error of THIS build on the committed converged-oracle traces of config 2
    (tests/golden/tail_decades.npz: 256 draws, every sample counts -- none is clamped to min_y)."""
    from metrotrpl_amd import batch
    fn = os.path.join(REPO, "tests", "golden", "tail_decades.npz")
    if not os.path.exists(fn):
        return None
    z = np.load(fn)
    n2 = int(z["n_cfg2"])
    idx = {n: i for i, n in enumerate(str(z["names"]).split(","))}
    sig, status, _ = batch.solve_batch(z["states"][:n2], idx, z["units"], z["iniPar"][:n2], z["thickness"][:n2], 128,
                                       z["tSteps"], return_stats=True, ctx=ctx)
    ref = z["sig"][:n2]
    dec = np.floor(np.log10(ref[:, :1]) - np.log10(ref)).astype(int)
    rel = np.abs(sig - ref) / ref
    per = [float(rel[dec == d].max()) for d in range(9)]
    return {"reference": "converged oracle (SciPy LSODA rtol 1e-12, atol 1e-24 densities / 1e-17 field), "
                         "tests/golden/tail_decades.npz, 256 config-2 draws x 1001 samples, none clamped",
            "gate": 1e-6, "max_rel_err_decades_0_to_5": max(per[:6]), "max_rel_err_decades_6_to_8": max(per[6:9]),
            "max_rel_err_per_decade_0_to_8": per, "all_ok": bool(np.all(status == 0))}


def cpu_baseline(sample=48):
    """Oracle (SciPy LSODA driving the gcc-built RHS through a Python trampoline --
    the reference's own structure with Numba replaced by gcc) on a bounded sample
    of the SAME workload, one core (the reference's unit of parallelism is one
    process per chain, metropolis.py:300)."""
    from metrotrpl_amd import workloads as W
    from oracle import oracle as O
    O.build()
    c = W.config2(sample)
    ix, un = W.indexes(), W.units()
    g = O.Grid(c["thickness"], c["nx"], c["tSteps"], c["hmax"])
    O.solve(c["iniPar"][0], g, c["states"][0], ix, units=un, ini_mode="fluence")  # warm-up (dlopen, caches)
    t0 = time.perf_counter()
    for k in range(sample):
        O.solve(c["iniPar"][k], g, c["states"][k], ix, units=un, ini_mode="fluence")
    dt = time.perf_counter() - t0
    out = {"value": sample / dt, "unit": "solves/s", "cores": 1, "kind": "port",
           "sample": f"first {sample} solves of the config-2 batch, SciPy LSODA (rtol 1e-7, atol 1e-10, max_step 4) "
                     f"+ gcc -O2 RHS via ctypes, {dt:.1f} s on 1 of {os.cpu_count()} host cores"}
    # the reference's way to use more cores is one process per chain (metropolis.py:300):
    # the same port, one solve per worker process, on this box's CPU share
    try:
        import multiprocessing as mp
        # every core this process may use, up to the GPU box's CPU share of 16 per GPU (its process guard; a one-GPU box
        # reports 256 host cores and grants 16)
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        share = 16 * max(1, int(os.environ.get("WORLD_SIZE", "1")))
        workers = max(1, min(avail, share))
        n_all = 24 * workers
        with mp.get_context("fork").Pool(workers) as pool:
            pool.map(_cpu_one, range(workers))  # warm-up
            t0 = time.perf_counter()
            pool.map(_cpu_one, range(n_all))
            dta = time.perf_counter() - t0
        out["all_cores"] = {"value": n_all / dta, "unit": "solves/s", "cores": workers,
                            "cores_available": os.cpu_count(), "cores_in_affinity_mask": avail,
                            "cores_note": "one worker process per core of this job's CPU share (16 per GPU on the GPU boxes "
                                          "of this pool, whatever os.cpu_count() says about the host)",
                            "sample": f"{n_all} solves of the same batch over {workers} worker processes, {dta:.1f} s"}
    except Exception as e:  # the single-core figure above is the contract; this one is extra
        out["all_cores"] = {"error": repr(e)}
    return out


def cpu_baseline_pt(S, n_chains_sample=4):
    """The oracle's eval_trial_move (SciPy LSODA + gcc RHS, IRF convolution and likelihood tail in C) on a
    bounded sample of the ensemble's chains, one core: proposals per second of the reference's own structure
    (one process evaluates one chain's proposal: 6 solves + 6 convolutions + the likelihood)."""
    from metrotrpl_amd import workloads as W
    from oracle import oracle as O
    O.build()
    irf = {745: (np.asarray(S["irf_tables"][745][0]), np.asarray(S["irf_tables"][745][1]))}
    shared = {"_sim_info": {"lengths": list(W.STAUB_LENGTHS), "nx": [128] * 6, "meas_types": ["TRPL"] * 6, "num_meas": 6},
              "_param_indexes": W.indexes(), "units": W.units(), "_init_params": np.array(W.STAUB_FLUENCES, dtype=float),
              "_times": [S["times"]] * 6, "_vals": S["vals"], "_uncs": S["uncs"], "hmax": 4.0, "solver": ("solveivp",),
              "model": "std", "ini_mode": "fluence", "rtol": 1e-7, "atol": 1e-10, "irf_convolution": [745] * 6,
              "_IRF_tables": irf}
    uniq = {"model_uncertainty": {"TRPL": 1.0}, "_T": 1.0}
    states = S["ens0"]["chain_states"]
    O.eval_trial_move(states[0], uniq, shared, temperatures=S["T"])  # warm-up
    t0 = time.perf_counter()
    for k in range(n_chains_sample):
        O.eval_trial_move(states[k % len(states)], uniq, shared, temperatures=S["T"])
    dt = time.perf_counter() - t0
    return {"value": n_chains_sample / dt, "unit": "proposals/s", "cores": 1, "kind": "port",
            "sample": f"oracle eval_trial_move (6 solves at rtol 1e-7 / atol 1e-10 + 6 IRF convolutions + likelihood at "
                      f"{len(S['T'])} temperatures) for the first {n_chains_sample} chains of the ensemble, {dt:.1f} s on 1 of "
                      f"{os.cpu_count()} host cores"}


def _cpu_one(k):
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    from metrotrpl_amd import workloads as W
    from oracle import oracle as O
    c = W.config2(64)
    g = O.Grid(c["thickness"], c["nx"], c["tSteps"], c["hmax"])
    O.solve(c["iniPar"][k % 64], g, c["states"][k % 64], W.indexes(), units=W.units(), ini_mode="fluence")
    return k


def _pt_leg(rank, world, local, dev, with_cpu, use_dist=False):
    """BASELINE.json's second metric, measured in the same run: a few iterations of the 64-chain (config 3) and
    the 512-chain (config 4 on this many GPUs) parallel-tempering ensembles through metrotrpl_amd.pt_bench.run."""
    import types
    from metrotrpl_amd import pt_bench
    out = {}

    def leg(chains, steps, warm, rank_, world_, use_dist_, cpu_):
        a = types.SimpleNamespace(chains=chains, steps=steps, warmup=warm, prefetch="auto", groups=0, opt=[],
                                  use_dist=use_dist_, no_cpu_baseline=not cpu_)
        ln = pt_bench.run(a, rank_, world_, local, dev, cpu_baseline_fn=cpu_baseline_pt)
        return {"chains": chains, "n_gpus": world_, "iterations": steps, "warmup": warm,
                "ms_per_iteration": ln["ms_per_step"], "proposals_per_sec": ln["value"],
                "forward_solves_per_sec": ln["config"]["forward_solves_per_sec"],
                "prefetch": ln["config"]["prefetch"], "chain_groups": ln["config"]["chain_groups"],
                "collectives_per_iteration": ln["config"].get("collectives_per_iteration"),
                "max_integrator_steps_per_rank": ln["config"].get("max_integrator_steps_per_rank"),
                "exchange_ms": ln["config"].get("exchange_ms"), "exchanges_timed": ln["config"].get("exchanges_timed"),
                **({"host_ms": ln["config"]["host_ms"]} if "host_ms" in ln["config"] else {}),
                **({"cpu_baseline": ln["cpu_baseline"]} if "cpu_baseline" in ln else {})}

    def together(ok):
        """Every rank leaves a leg at the same place and knows whether it failed anywhere: the ranks that have
        nothing to do in a rank-0-only leg wait HERE, visibly, not in whatever collective happens to come next."""
        if not (use_dist and world > 1):
            return ok
        import torch
        import torch.distributed as dist
        from metrotrpl_amd import pt as _pt
        flag = torch.tensor([0 if ok else 1], dtype=torch.int32, device=_pt.coll_device(dev))
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return int(flag.item()) == 0

    def guarded(key, fn):
        try:
            out[key] = fn()
            return True
        except Exception as e:  # the solves/s contract line must not depend on this leg
            out[key] = {"error": repr(e)}
            return False

    if world == 1:
        # (with a process group -- MTRPL_FORCE_DIST -- the warm-up spans two replica exchanges: the first collective of
        # each shape sets RCCL up)
        guarded("64", lambda: leg(64, 30, 10, rank, world, use_dist, with_cpu))
        guarded("512", lambda: leg(512, 20, 20 if use_dist else 10, rank, world, use_dist, False))
        return out
    # N > 1.  (1) the strong-scaling figure: the 512-chain ensemble block-partitioned over all ranks
    ok = guarded("512", lambda: leg(512, 30, 20, rank, world, use_dist, False))
    if not together(ok) and "error" not in out["512"]:
        out["512"]["error"] = "the leg failed on another rank"
    # (2) what only rank 0 runs -- config 3 (64 chains) is a ONE-GPU configuration, and the same 512-chain ensemble of
    # the same build on one GPU gives the line its own N = 1 figure -- while the others wait in together() below
    if rank == 0:
        guarded("64", lambda: leg(64, 30, 10, 0, 1, False, False))
        if "error" not in out["512"]:
            out["512"]["scaling"] = "strong"
            one = {}
            try:
                one = leg(512, 20, 10, 0, 1, False, False)
                out["512"]["one_gpu_same_build"] = one
                out["512"]["speedup_vs_one_gpu"] = one["ms_per_iteration"] / out["512"]["ms_per_iteration"]
            except Exception as e:
                out["512"]["one_gpu_same_build"] = {"error": repr(e)}
    together(True)
    return out


def _api_leg(local, dev, B=8192, n_samples=8192):
    """What a caller of the PRODUCT API gets, host to host -- no streams, no device pointers, nothing
    set by the caller: (a) mtrpl_solve_batch on B config-2 solves, descriptors in host memory in, samples in host memory
    out (the C call alone into pinned and into pageable memory, and batch.solve_batch with its Python packing);
    (b) dense_sampler.simulate -- the reference's Dense_Sample/dense_sampling.py:48-195 -- on the 6-measurement power scan
    with IRF convolution and likelihood, samples per second."""
    import ctypes as C
    import torch
    from metrotrpl_amd import _lib as L, batch, dense_sampler as DS, pt_bench, workloads as W
    out = {}
    ctx = L.default_context(local)
    c = W.config2(B)
    t = np.ascontiguousarray(c["tSteps"])
    nt = t.shape[0]
    d, ini = batch.build_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], c["nx"], nt, hmax=c["hmax"],
                               meas=c["meas"], model=c["model"], ini_mode=c["ini_mode"])
    status = np.zeros(B, dtype=np.int32)

    def c_call(buf):
        ctx.check(L.lib().mtrpl_solve_batch(ctx.handle, B, d.ctypes.data_as(C.c_void_p), L.dptr(ini), ini.shape[0], L.dptr(t),
                                            nt, L.dptr(buf), buf.shape[0], L.iptr(status), None))

    legs = {}
    for name, buf in (("pinned_out", L.pinned_f64(ctx, B * nt)), ("pageable_out", np.empty(B * nt))):
        c_call(buf)  # warm-up: device scratch, streams, staging
        t0 = time.perf_counter()
        n_rep = 3
        for _ in range(n_rep):
            c_call(buf)
        dt = (time.perf_counter() - t0) / n_rep
        legs[name] = {"ms_per_call": dt * 1e3, "solves_per_sec": B / dt,
                      "all_solves_ok": bool(np.all(status == 0) and np.all(np.isfinite(buf)))}
    t0 = time.perf_counter()
    sig = batch.solve_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], c["nx"], t, hmax=c["hmax"],
                            meas=c["meas"], model=c["model"], ini_mode=c["ini_mode"], ctx=ctx)
    dt = time.perf_counter() - t0
    legs["python_batch_solve_batch"] = {"ms_per_call": dt * 1e3, "solves_per_sec": B / dt, "all_solves_ok": bool(np.all(np.isfinite(sig)))}
    # ... and with the result left in the context's pinned memory (pinned=True: for results that are consumed and dropped)
    t0 = time.perf_counter()
    sigp = batch.solve_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], c["nx"], t, hmax=c["hmax"],
                             meas=c["meas"], model=c["model"], ini_mode=c["ini_mode"], ctx=ctx, pinned=True)
    dt = time.perf_counter() - t0
    legs["python_batch_solve_batch_pinned"] = {"ms_per_call": dt * 1e3, "solves_per_sec": B / dt,
                                               "same_values": bool(np.array_equal(sigp, sig))}
    del sig, sigp
    out["solve_batch"] = {"solves": B, "host_chunk": int(ctx.get_option("host_chunk")), "host_streams": int(ctx.get_option("host_streams")),
                          "solves_per_sec": legs["pinned_out"]["solves_per_sec"], **legs,
                          "note": "PCIe-inclusive: descriptors from host memory, 8 KB of samples per solve back into host "
                                  "memory, one blocking call; the library cuts it into sub-launches on its own streams"}
    # (b) the dense sampler on the config-3 geometry (6 measurements, nx = 128, IRF convolution, likelihood at T = 1)
    S = pt_bench.setup(64, dev, local)
    eng = S["eng"]
    names, ix = W.PARAM_NAMES, W.indexes()
    s0 = W.initial_state()
    lo = np.array([W.PRIOR_BOX.get(n, (s0[ix[n]], s0[ix[n]]))[0] for n in names])
    hi = np.array([W.PRIOR_BOX.get(n, (s0[ix[n]], s0[ix[n]]))[1] for n in names])
    X = DS.random_grid(lo, hi, [True] * len(names), n_samples, np.random.default_rng(W.SEED + 77))
    ok = np.abs(np.log10(X[:, ix["tauN"]]) - np.log10(X[:, ix["tauP"]])) <= 2  # the reference's physics check
    X = X[ok]
    DS.simulate(eng, X[:2048])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    P = DS.simulate(eng, X)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    P1 = DS.simulate(eng, X[:2048], chunk=512, lanes=1)
    dt1 = time.perf_counter() - t0
    out["dense_sampler"] = {"samples": int(X.shape[0]), "measurements": 6, "samples_per_sec": X.shape[0] / dt,
                            "forward_solves_per_sec": 6 * X.shape[0] / dt, "finite": int(np.isfinite(P).sum()),
                            "one_chunk_at_a_time": {"samples_per_sec": 2048 / dt1, "same_values": bool(np.array_equal(P[:2048], P1))},
                            "note": "dense_sampler.simulate: host array of samples in, log-likelihoods out; chunks of 1024 "
                                    "samples (6144 solves + IRF convolution + likelihood) pipelined over three lanes"}
    return out


_STDOUT_FD = None


def emit(line):
    """The ONE line of the contract, on the real stdout (see main)."""
    data = (json.dumps(line) + "\n").encode()
    if _STDOUT_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_STDOUT_FD, data)


def _leg_watchdog(rank, line, key):
    """N > 1 only: a timer that ends the job if the leg it guards is still running after MTRPL_BENCH_LEG_TIMEOUT seconds
    (default 420; the legs take about one minute) -- a collective that never returns would otherwise take the contract
    line, already measured, with it.  Rank 0 emits the line with `{key: {"error": ...}}`; every rank exits 0 (the other
    ranks a moment later, so that the launcher sees rank 0's output first)."""
    import threading
    limit = float(os.environ.get("MTRPL_BENCH_LEG_TIMEOUT", "420"))

    def fire():
        try:
            if rank == 0 and line is not None:
                line[key] = {"error": f"the {key} leg did not finish within {limit:.0f} s and was abandoned (bench.py watchdog)"}
                emit(line)
            else:
                time.sleep(3.0)
        finally:
            os._exit(0)
    t = threading.Timer(limit, fire)
    t.daemon = True
    t.start()
    return t


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as a CHILD
    `python -m torch.distributed.run` (never an exec, and before this process has made any GPU call -- counting devices
    does not initialise the runtime), relay rank 0's contract line to the real stdout and return the child's exit code.
    Without N visible GPUs nothing is started: exit code 3, nothing on stdout (there is no CPU fallback)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n and os.environ.get("MTRPL_REHEARSE_ONE_GPU") != "1":
        print(f"bench.py: --gpus {n} but {have} GPU(s) visible; the HIP path has no CPU fallback", file=sys.stderr)
        return 3
    with socket.socket() as s:  # a free rendezvous port on the loop-back interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: starting the ranks: " + " ".join(cmd), file=sys.stderr)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None)
    out, _ = child.communicate()
    line = None
    for ln in out.decode(errors="replace").splitlines():  # the launcher may talk on stdout too: the contract line only
        try:
            if "metric" in json.loads(ln):
                line = ln
        except (ValueError, TypeError):
            print(ln, file=sys.stderr)
    if line is not None and child.returncode == 0:
        os.write(_STDOUT_FD, (line + "\n").encode())
    elif child.returncode == 0:
        print("bench.py: the ranks ended without a contract line", file=sys.stderr)
        return 1
    return child.returncode


def main():
    # Libraries below this process write to file descriptor 1 on their own (RCCL prints a version banner when the first
    # communicator is made): everything but the contract line goes to stderr.
    global _STDOUT_FD
    # HIP maps streams onto 4 hardware queues by default: with the four streams of the timed region next to PyTorch's own,
    # two of them would share a queue and their launches serialise behind each other (measured: 330 k solves/s on four
    # streams against 434 k on three).  Read when the runtime starts; 16 is what the package sets for every caller
    # (metrotrpl_amd/_lib.py, the library's own constructor): nothing here that a user of the product would not get.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    sys.stdout.flush()
    _STDOUT_FD = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="solves per GPU and step (config 2: 1024)")
    ap.add_argument("--workload", default="batch", choices=["batch", "pt", "config5"])
    ap.add_argument("--chains", type=int, default=512, help="--workload pt: total PT chains (strong scaling)")
    ap.add_argument("--prefetch", default="auto", choices=["auto", "on", "off"],
                    help="--workload pt: two Metropolis iterations per launch (both outcomes of the first "
                         "evaluated ahead); auto = when a rank's chain block leaves most SIMDs idle")
    ap.add_argument("--groups", type=int, default=0,
                    help="--workload pt: chain groups that run between two replica exchanges at their own pace, one "
                         "HIP stream each (the launch of one overlaps the host work of the other); 0 = 2 where a "
                         "rank's chain block leaves SIMDs idle (the prefetching regime), else 1")
    ap.add_argument("--decide", default="device", choices=["device", "host"],
                    help="--workload pt, prefetching: where a launch's accept / reject decisions are taken -- on the device, "
                         "behind the launch on its stream (the host prepares the next launch's four possible proposal "
                         "sets while it runs), or on the host between two launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=4,
                    help="issue successive steps round-robin on this many HIP streams, so that the next batch fills "
                         "the SIMDs freed by the early finishers of the previous one (1: back-to-back launches on one "
                         "stream -- always measured as well and reported as single_stream)")
    ap.add_argument("--quick", action="store_true",
                    help="the timed region only: no single-stream / sorted / fp64-linear-algebra / PT legs (profiling "
                         "runs: the kernel statistics then hold the timed kind of launch only)")
    ap.add_argument("--no-pt", action="store_true", help="skip the parallel-tempering leg")
    ap.add_argument("--no-api", action="store_true", help="skip the host-to-host product-API leg")
    ap.add_argument("--sort-by-cost", type=int, default=0,
                    help="1: the TIMED region issues the solves of the batch longest first (order measured in a warm-up "
                         "launch of the identical batch; experiments only -- the default line reports that figure as "
                         "sorted_by_measured_cost next to the as-drawn headline)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="integrator option for experiments (mtrpl_set_option); reported in config.options")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # MTRPL_REHEARSE_ONE_GPU=1: every rank on device 0, process group `gloo` -- a rehearsal of the N > 1 control flow (the
    # rank-0-only legs, the hand-overs between them, the assembly of the line) on a one-GPU box; its numbers mean nothing
    rehearse = os.environ.get("MTRPL_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local = 0
    # one rank needs no process group; MTRPL_FORCE_DIST=1 makes a 1-rank run go through RCCL
    # anyway (rehearsal of the N > 1 code path on a one-GPU box)
    use_dist = world > 1 or os.environ.get("MTRPL_FORCE_DIST") == "1"
    a.use_dist = use_dist
    # CPU baseline first: its worker processes are forked before this process touches the GPU.  At N > 1 rank 0 times
    # it too (the other ranks wait for it in init_process_group), so that every line the driver records carries one.
    cpu = None
    if rank == 0 and a.workload == "batch" and not a.no_cpu_baseline:
        cpu = cpu_baseline()
    import torch
    import torch.distributed as dist
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and a.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from metrotrpl_amd import _lib as L, batch, pt as _pt, workloads as W
    if a.workload == "pt":
        from metrotrpl_amd import pt_bench
        line = pt_bench.run(a, rank, world, local, dev, cpu_baseline_fn=cpu_baseline_pt)
        if rank == 0:
            emit(line)
        if use_dist:
            dist.destroy_process_group()
        return
    if a.workload == "config5":
        from metrotrpl_amd import config5_bench
        line = config5_bench.run(a, rank, world, local, dev)
        if rank == 0:
            emit(line)
        if use_dist:
            dist.destroy_process_group()
        return

    ctx = L.Context(local)
    user_variant = False
    for kv in a.opt:
        k, v = kv.split("=")
        ctx.set_option(k, float(v))
        user_variant = user_variant or k in ("kernel_variant", "two_per_simd_min_batch")
    # (--opt kernel_variant / two_per_simd_min_batch: the launches then take what the options say, per_simd stays 0)
    B = a.batch
    # rank r takes draws [r*B, (r+1)*B) of the seeded config-2 stream
    c = W.config2(B * world)
    sl = slice(rank * B, (rank + 1) * B)
    t_host = np.ascontiguousarray(c["tSteps"])
    nt = t_host.shape[0]
    descs, ini = batch.build_batch(c["states"][sl], W.indexes(), W.units(), c["iniPar"][sl], c["thickness"], c["nx"],
                                   nt, hmax=c["hmax"], meas=c["meas"], model=c["model"], ini_mode=c["ini_mode"])
    d_descs_drawn = torch.from_numpy(descs.view(np.uint8).reshape(-1)).to(dev)
    cur = {"descs": d_descs_drawn, "per_simd": 0}
    d_ini = torch.from_numpy(ini).to(dev)
    d_t = torch.from_numpy(t_host).to(dev)
    S = max(1, a.streams)
    outs = [(torch.zeros(B * nt, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.int32, device=dev),
             torch.zeros(B * 8, dtype=torch.int32, device=dev)) for _ in range(S)]
    d_out, d_status, d_stats = outs[0]
    streams = [torch.cuda.current_stream()] + L.shared_streams(dev, S - 1)
    torch.cuda.synchronize()

    def step(k=0):
        o, st_, sa_ = outs[k % S]
        # (per_simd: 0 = the library picks the build of the wave kernel by the size of the launch; 2 = two solves per SIMD,
        # said per launch by the caller who has several launches in flight -- mtrpl_solve_batch_dev2)
        ctx.check(L.lib().mtrpl_solve_batch_dev2(ctx.handle, B, cur["descs"].data_ptr(), d_ini.data_ptr(), d_t.data_ptr(),
                                                 o.data_ptr(), st_.data_ptr(), sa_.data_ptr(),
                                                 int(c["nx"]), L.MODEL[c["model"]], streams[k % S].cuda_stream, cur["per_simd"]))

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    stats_drawn = d_stats.cpu().numpy().reshape(B, 8).copy()

    def timed(n_streams):
        """K steps issued round-robin on n_streams HIP streams, bracketed as the contract says."""
        for o, st_, _ in outs:  # a launch that does not run leaves a non-zero status behind
            st_.fill_(-1)
            o.fill_(float("nan"))
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
        t0 = time.perf_counter()
        for k in range(a.steps):
            q = k % n_streams
            ev[k][0].record(streams[q])
            step(q)
            ev[k][1].record(streams[q])
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([dt], dtype=torch.float64, device=_pt.coll_device(dev))
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        ok = all(bool((outs[q][1] == 0).all().item()) and bool(torch.isfinite(outs[q][0]).all().item())
                 for q in range(min(n_streams, a.steps)))
        ms = [e0.elapsed_time(e1) for e0, e1 in ev]
        spread[0] = (float(np.min(ms)), float(np.max(ms)))
        return dt, float(np.mean(ms)), ok

    spread = [None]  # (min, max) of the per-launch HIP-event brackets of the last timed() call

    def region(dt, kern_ms, ok, mm=None):
        return {"solves_per_sec": world * B * a.steps / dt, "ms_per_step": dt / a.steps * 1e3, "kernel_ms": kern_ms,
                **({"kernel_ms_min": mm[0], "kernel_ms_max": mm[1]} if mm else {}), "all_solves_ok": ok}

    # ---- (1) one launch at a time, as drawn: the latency figure and the kernel's own speed ----
    single = fp64la = sorted_leg = single_mm = None
    if not a.quick and S > 1:
        single = timed(1)
        single_mm = spread[0]
        if not user_variant:
            ctx.set_option("kernel_variant", 2)
            step()
            torch.cuda.synchronize()
            fp64la = timed(1)
            ctx.set_option("kernel_variant", 0)
    # ---- (2) the timed region of the headline: S streams, as drawn ----
    if S > 1 and not user_variant:
        # several launches in flight: more solves than SIMDs, so the two-solves-per-SIMD form of the wave kernel
        # (a launch on its own would pick it from 1025 solves on)
        cur["per_simd"] = 2
    if a.sort_by_cost:
        order = np.argsort(-stats_drawn[:, 7], kind="stable")
        cur["descs"] = torch.from_numpy(descs[order].view(np.uint8).reshape(-1)).to(dev)
    for k in range(S):
        step(k)
    torch.cuda.synchronize()
    dt, kern_ms, ok = timed(S)
    kern_mm = spread[0]
    if S == 1:
        single, single_mm = (dt, kern_ms, ok), kern_mm
    status = d_status.cpu().numpy()
    stats = d_stats.cpu().numpy().reshape(B, 8)
    # ---- (3) the same region with each batch's solves longest first by the cost measured on the identical batch ----
    if not a.quick and S > 1 and not a.sort_by_cost:
        order = np.argsort(-stats_drawn[:, 7], kind="stable")
        cur["descs"] = torch.from_numpy(descs[order].view(np.uint8).reshape(-1)).to(dev)
        for k in range(S):
            step(k)
        torch.cuda.synchronize()
        sorted_leg = timed(S)
        cur["descs"] = d_descs_drawn
    # ---- (4) the timed region on the all-fp64 linear algebra (kernel_variant 2: one solve per SIMD, 34 KB of LDS) ----
    fp64la_streams = None
    if not a.quick and S > 1 and not user_variant and not a.sort_by_cost:
        ctx.set_option("kernel_variant", 2)
        for k in range(S):
            step(k)
        torch.cuda.synchronize()
        fp64la_streams = timed(S)
        ctx.set_option("kernel_variant", 0)
    # ---- (5) what a caller WITHOUT streams gets from one bigger launch: 2048 and 4096 solves, one launch at a time
    #          (from 1025 solves on a launch picks the two-solves-per-SIMD build by itself) ----
    one_launch = {}
    if not a.quick and not user_variant and not a.sort_by_cost:
        for B2 in (2048, 4096):
            c2 = W.config2(B2 * world)
            sl2 = slice(rank * B2, (rank + 1) * B2)
            descs2, ini2 = batch.build_batch(c2["states"][sl2], W.indexes(), W.units(), c2["iniPar"][sl2], c2["thickness"],
                                             c2["nx"], nt, hmax=c2["hmax"], meas=c2["meas"], model=c2["model"],
                                             ini_mode=c2["ini_mode"])
            d2 = torch.from_numpy(descs2.view(np.uint8).reshape(-1)).to(dev)
            o2 = torch.zeros(B2 * nt, dtype=torch.float64, device=dev)
            st2 = torch.full((B2,), -1, dtype=torch.int32, device=dev)
            sa2 = torch.zeros(B2 * 8, dtype=torch.int32, device=dev)

            def step2():
                ctx.check(L.lib().mtrpl_solve_batch_dev(ctx.handle, B2, d2.data_ptr(), d_ini.data_ptr(), d_t.data_ptr(),
                                                        o2.data_ptr(), st2.data_ptr(), sa2.data_ptr(), int(c2["nx"]),
                                                        L.MODEL[c2["model"]], streams[0].cuda_stream))
            step2()
            torch.cuda.synchronize()
            n2 = 6
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(streams[0])
            for _ in range(n2):
                step2()
            e1.record(streams[0])
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            one_launch[str(B2)] = {"solves_per_launch": B2, "launches": n2, "ms_per_launch": dt2 / n2 * 1e3,
                                   "kernel_ms": e0.elapsed_time(e1) / n2, "solves_per_sec": B2 * n2 / dt2,
                                   "all_solves_ok": bool((st2 == 0).all().item()) and bool(torch.isfinite(o2).all().item())}
            del d2, o2, st2, sa2

    # measured HBM stream rate of this device (read + write of a 1 GiB buffer), next to the nominal peak
    copy_gbs = None
    if rank == 0:
        try:
            src = torch.empty(1 << 27, dtype=torch.float64, device=dev)  # 1 GiB
            dst = torch.empty_like(src)
            dst.copy_(src)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 5 * 2 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src, dst
        except Exception:
            copy_gbs = None
    acc = None
    if rank == 0:
        try:
            acc = accuracy_object(ctx)
        except Exception as e:  # the contract line must not depend on the fixture
            acc = {"error": repr(e)}
    api = None
    if rank == 0 and not a.quick and not user_variant and not a.no_api:
        try:
            api = _api_leg(local, dev)
        except Exception as e:  # the contract line must not depend on this leg
            api = {"error": repr(e)}
    line = None
    if rank == 0:
        A = algorithmic_bytes_per_solve(c["nx"], nt)
        achieved = A * B / (kern_ms * 1e-3) / 1e9
        from metrotrpl_amd.pt_bench import pmc_fields, valu_issue_fields
        two = (S > 1 and not user_variant) or (not user_variant and B >= 1025)
        kern_name = "solve_kernel_wave2" if two else ("solve_kernel_wave1" if not user_variant else "as set by --opt")
        pmc = pmc_fields(B, kern_name)
        value = world * B * a.steps / dt
        line = {
            "metric": "trpl_forward_solves_per_sec", "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "linear_algebra": "state, residual, RHS, Jacobian, factorisation and output in fp64; the elimination blocks of "
                              "the Newton matrix are stored as fp32 and re-applied in fp32 (approximate inverse on the fp64 "
                              "residual); error norms and step controller in fp32; the all-fp64 linear algebra "
                              "(kernel_variant 2) is timed as fp64_linear_algebra",
            "config": {"workload": "config2: 1024-solve synthetic batch per GPU, nx=128, 2000 nm single-layer absorber, "
                                   "1001 output times over 2000 ns, hmax=4, std model, TRPL, called with the reference's "
                                   "default RTOL 1e-7 / ATOL 1e-10 (the engine integrates to its own, tighter policy: "
                                   "converged trace within 1e-6, include/metrotrpl_amd.h)",
                       "solves_per_gpu_per_step": B, "nx": int(c["nx"]), "nt_out": int(nt), "parallelism": f"batch-shard x{world}",
                       "all_solves_ok": bool(ok and np.all(status == 0)), "mean_integrator_steps": float(stats[:, 0].mean()),
                       "mean_rhs_evals": float(stats[:, 1].mean()), "max_integrator_steps": int(stats[:, 0].max()),
                       **({"options": a.opt} if a.opt else {}),
                       "streams": S, "solves_per_simd": 2 if (S > 1 and not user_variant) else "as set by --opt" if user_variant
                       else 1, "issue_order": "longest first (measured in a warm-up launch of the identical batch)"
                       if a.sort_by_cost else "as drawn: batches handed over as drawn and issued in that order (round 3's "
                       "device-side ordering by a cost model is gone: it bought 1.6 %)",
                       "as_drawn": not a.sort_by_cost},
            "single_stream": None if single is None else {
                **region(*single, mm=single_mm), "note": "the same K steps back to back on one HIP stream, as drawn, one solve per SIMD "
                                           "(every launch alone on the device), measured in this run"},
            "sorted_by_measured_cost": None if sorted_leg is None else {
                **region(*sorted_leg), "note": "the timed region again with each batch's solves issued longest first, the "
                                               "order measured on the identical batch in a warm-up launch: an upper bound of "
                                               "what issue ordering buys, not available to a first-time caller"},
            "fp64_linear_algebra": None if fp64la is None else {
                **region(*fp64la), "note": "single_stream on kernel_variant 2: fp64 elimination blocks and fp64 re-solve",
                "streams": None if fp64la_streams is None else {
                    **region(*fp64la_streams), "note": "the timed region (same streams, as drawn) on kernel_variant 2: one "
                                                       "solve per SIMD, so the second and later launches queue for SIMDs"}},
            "one_launch": one_launch or None,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, **pmc, "kernel": kern_name,
                         "kernel_ms": kern_ms, "kernel_ms_min": kern_mm[0], "kernel_ms_max": kern_mm[1],
                         "kernel_ms_brackets": "HIP events on the launch's own stream around the solve "
                         "kernel alone (since round 4 nothing else is enqueued by a launch); with several launches in flight "
                         "the bracket includes the time a launch's workgroups wait for SIMDs the other launches hold",
                         "launches_in_flight": S, "algorithmic_bytes_per_solve": A,
                         "device": {"achieved": A * B / (dt / a.steps) / 1e9, "frac": A * B / (dt / a.steps) / 1e9 / HBM_PEAK_GBS,
                                    "note": "the same bytes per launch over the time per step: what the device does with "
                                            "launches_in_flight launches sharing it (achieved / frac above are per launch)"},
                         "single_stream": None if single is None else {
                             "kernel_ms": single[1], "achieved": A * B / (single[1] * 1e-3) / 1e9,
                             "frac": A * B / (single[1] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "valu_issue": valu_issue_fields(kern_name, value / world),
                         "measured_stream_copy": copy_gbs,
                         "note": "state-streaming model of SURVEY.md 8d; the solve state is register-resident by "
                                 "design, so HBM is not the limiter -- valu_issue is the binding roofline (DESIGN.md)"},
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        line["accuracy"] = acc
        if api is not None:
            line["api"] = api
        if world > 1:
            line["multi_gpu"] = {
                "first_run_on_hardware": True,
                "note": "no multi-GPU node was available while this was built: before this run the N > 1 path had been "
                        "exercised as one rank through RCCL and as two ranks sharing one GPU over gloo "
                        "(tests/test_gpu_bench_line.py) and on CPU ranks (tests/test_pt_distributed.py); the timed region "
                        "itself has no collective but the barrier and the max-over-ranks of the clock"}
    # ---- BASELINE's second metric, after everything the contract line needs is in hand.  At N > 1 its legs are the only
    # part of this program with collectives in them that no hardware has run before: a watchdog on every rank makes sure
    # that a leg that hangs costs the line its `pt` object, not the line (rank 0 emits what it has, every rank leaves).
    pt_leg = None
    if not a.quick and not a.no_pt and not user_variant:
        del outs, d_out, d_status, d_stats
        wd = _leg_watchdog(rank, line, "pt") if world > 1 else None
        pt_leg = _pt_leg(rank, world, local, dev, with_cpu=(cpu is not None), use_dist=use_dist)
        if wd is not None:
            wd.cancel()
    if rank == 0:
        if pt_leg is not None:
            line["pt"] = pt_leg
        emit(line)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
