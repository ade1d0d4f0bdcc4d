"""bench.py --workload pt : BASELINE.json configs[2]/[3] -- a parallel-tempering
ensemble on the staub 6-measurement power scan (lengths 311/2000 nm x 3 fluences,
nx=128, 1001 sample times, IRF convolution with a 189-tap IRF, ladder
T_k = 1000^(k/(n-1))), chains block-partitioned over the ranks (strong scaling),
one RCCL all-gather of the per-chain log-likelihood table (and the states) per replica exchange.

One "step" = one Metropolis iteration of the whole ensemble: propose for every
chain, n_chains x 6 forward solves + IRF convolutions + likelihoods at all ladder
temperatures on the GPUs, accept/reject of the rank's own chains, and (every temper_freq
iterations, SURVEY.md 8d: 10) the all-gather + replica-exchange replay.
"""
import gc
import json
import os
import time

import numpy as np

from . import _lib as L
from . import pt as _pt
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


def setup(n_chains, dev, local=0, nx=128, irf_raw=None, parts=2, sort_by_cost=True, split_min=2048, sort_min=1024,
          ctx=None, force_min_y=False, groups=1, packed_order="chains", dealt=None, two_per_simd=False):
    """The config-3/4 problem on one rank: engine(s), ladder, evaluator(s) and proposer.  groups > 1: one
    engine, evaluator and HIP stream per chain group of pt.PTEnsemble.  `dealt` = k: evaluators for an ensemble that deals its
    chains to the groups by cost (`PTEnsemble(regroup_by_cost=True)`: the two go together) -- they report what a chain cost,
    the first k groups (the expensive chains) run one solve per SIMD and the others two.  `two_per_simd`: every group's
    launches, whatever their size, run two solves per SIMD (several throughput-bound groups share the device)."""
    import torch
    times = W.default_times()
    ctx = ctx or L.default_context(local)
    irf_tables = laplace.make_I_tables({745: synthetic_irf() if irf_raw is None else irf_raw})
    vals, uncs = synthetic_data(irf_tables, nx, times)
    ens0 = W.pt_ensemble(n_chains, nx)
    T = ens0["temperatures"]
    d_T = torch.from_numpy(np.ascontiguousarray(T)).to(dev)

    def make(g_):
        eng = LikelihoodEngine(W.indexes(), W.units(), W.STAUB_LENGTHS, nx, ["TRPL"] * 6, [times] * 6, vals, uncs,
                               {"TRPL": 1.0}, np.array(W.STAUB_FLUENCES), ini_mode="fluence", irf_tables=irf_tables,
                               irf_convolution=[745] * 6, device=local, ctx=ctx, sort_by_cost=sort_by_cost, parts=parts,
                               split_min=split_min, sort_min=sort_min, force_min_y=force_min_y)
        eng.prof = None  # a dict: host seconds spent packing + uploading / enqueuing (tools/pt_host_phases.py)

        def evaluator(states_block, temps):
            t0 = time.perf_counter()
            eng.upload(states_block)
            t1 = time.perf_counter()
            ll = eng.launch(d_T, len(temps))
            tab = ll.view(-1, eng.M, len(temps)).sum(dim=1)
            if eng.prof is not None:
                eng.prof["upload"] = eng.prof.get("upload", 0.0) + t1 - t0
                eng.prof["launch"] = eng.prof.get("launch", 0.0) + time.perf_counter() - t1
            return tab

        # A latency-bound ensemble -- the launches of all its groups fit the device at once, if some of them run two solves
        # per SIMD -- has its chains dealt to the groups by cost at every replica exchange (pt.PTEnsemble.regroup_by_cost),
        # the first `dealt` groups (the expensive chains) run one solve per SIMD and the others two: nobody waits for a SIMD,
        # and the solves that take 1.4x as long are the ones with time to spare.  64 chains, two iterations per launch:
        # 396 + 378 + 189 of 1024 SIMDs (all one per SIMD: 1152); 256 chains, one iteration: 516 + 255 + 255.
        # (said per launch -- LikelihoodEngine.per_simd -> mtrpl_solve_batch_dev2 --, not by flipping an option of the context
        # the groups share around every launch: ADVICE r5)
        if dealt is not None:
            eng.per_simd = 1 if g_ < dealt else 2
        elif two_per_simd:
            eng.per_simd = 2

        def launch_packed(d_descs, n_solves, temps):
            eng.use_packed(d_descs, n_solves)
            return eng.launch(d_T, len(temps)).view(-1, eng.M, len(temps)).sum(dim=1)
        evaluator.pack_rows, evaluator.launch_packed, evaluator.packable = eng.pack_rows, launch_packed, eng.packable()

        # pt._DecidePipe: the two decisions of a launch and the pick of the next launch's descriptors, one kernel each
        def select(cand, n_cand, n_rows, chain_of_row, outcome, out):
            ctx.check(L.lib().mtrpl_pt_select_dev(ctx.handle, n_rows, n_cand, L.SOLVE_DESC_DTYPE.itemsize, cand.data_ptr(),
                                                  chain_of_row.data_ptr(), outcome.data_ptr(), out.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream))

        def decide(tab, n_props, col, row, u, cur_ll, cur_tab, outcome, result):
            ctx.check(L.lib().mtrpl_pt_decide_dev(ctx.handle, col.shape[0], tab.shape[1], n_props, tab.data_ptr(),
                                                  col.data_ptr(), row.data_ptr(), u.data_ptr(), cur_ll.data_ptr(),
                                                  cur_tab.data_ptr(), outcome.data_ptr(), result.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream))
        evaluator.select, evaluator.decide = select, decide

        def row_cost(n_rows):  # integrator steps of the slowest of a state row's M solves in the last launch
            return eng.d_stats[:8 * n_rows * eng.M].view(n_rows, eng.M, 8)[:, :, 0].max(dim=1).values.cpu().numpy()
        # the order of a packed launch's solves: big launches solve by solve (the engine's measured-cost order and
        # stream split, fixed per stretch between two exchanges), small ones chain by chain (pt._DecidePipe)
        # (ordering alone measured nothing on the three-group 64-chain ensemble: 2.37 ms per iteration chain by chain, solve
        # by solve and as given; what `row_cost` is for there is the deal of the chains to the groups, above)
        if os.environ.get("MTRPL_PT_NO_ORDER") != "1":
            if packed_order == "solves":
                # every group's launches, whatever their size (three groups of a 512-chain ensemble are 1026,
                # 1026 and 1020 solves -- a threshold of 1024 ordered two of them; at N > 1 none)
                eng.packed_sort_min = 0
                evaluator.plan_packed = eng.plan_packed
            elif dealt is not None:
                evaluator.row_cost = row_cost
        return eng, evaluator
    made = [make(g) for g in range(max(1, groups))]
    engines, evaluators = [m[0] for m in made], [m[1] for m in made]
    streams = L.shared_streams(dev, len(engines)) if groups > 1 else None  # (the same few streams in every stage)

    names = W.PARAM_NAMES
    ix = W.indexes()
    active = [n in W.PRIOR_BOX for n in names]
    prior = {n: W.PRIOR_BOX.get(n, (0, np.inf)) for n in names}
    do_log = np.ones(len(names), dtype=bool)
    base_move = np.where(active, 0.003, 0.0)  # MCMC_script_writer.py:116
    proposer = pt.Proposer(base_move, names, active, prior, do_log, ix)

    return dict(eng=engines[0], engines=engines, ens0=ens0, T=T, d_T=d_T, evaluator=evaluators[0],
                evaluators=evaluators, streams=streams, proposer=proposer, times=times, vals=vals, uncs=uncs,
                irf_tables=irf_tables)


def run(a, rank, world, local, dev, cpu_baseline_fn=None):
    """One `bench.py --workload pt` line (also the `pt` legs of the default line).  Experiment switches, all off by default
    (environment): MTRPL_NO_SORT=1 / MTRPL_PARTS=n (the engine's measured-cost order / stream split), MTRPL_PT_NO_ORDER=1
    (packed launches as given), MTRPL_PT_NO_REGROUP=1 (a latency-bound ensemble keeps its interleaved chain groups and one
    kernel build for all of them), MTRPL_PT_ONE_GROUPS=k (how many of its groups run one solve per SIMD),
    MTRPL_LL_WAVE=0|1|2|4 (waves per likelihood block where several throughput-bound groups share the device; 0 = twelve),
    MTRPL_PT_PROF=1 (host milliseconds per phase into config.host_ms)."""
    import torch
    n_chains = a.chains
    # Prefetching (pt.PTEnsemble.iterate2): when this rank's block leaves most SIMDs idle -- one solve occupies one
    # SIMD for the whole launch -- evaluate iteration k+1's proposals for both outcomes of iteration k in the same
    # launch: 3x the solves, 2 iterations per solve latency.  Pays while a launch of 3x the solves takes less than
    # two launches (measured: up to ~0.8 of the SIMD count); pointless once the launch is throughput-bound.
    slots = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
    lo, hi = pt.partition(n_chains, world)[rank]
    mode = getattr(a, "prefetch", "auto")
    prefetch = mode == "on" or (mode == "auto" and (hi - lo) * 6 <= 0.8 * slots)
    # chain groups (one engine + HIP stream each, each walking between two replica exchanges at its own pace).  With the
    # decisions taken on the device no launch waits for the host, and what several groups buy is that one group's
    # launch fills the SIMDs the others' slowest solves leave idle: three, in the latency-bound (prefetching) regime
    # and in the throughput-bound one.  With the decisions on the host: two where the host work of one can hide
    # behind the launch of the other (prefetching), one otherwise.
    decide = getattr(a, "decide", "device")
    groups = int(getattr(a, "groups", 0)) or (3 if decide == "device" else (2 if prefetch else 1))
    # Latency-bound or throughput-bound?  g solves per group launch, all groups in flight: with k groups one solve per SIMD
    # and the others two they need k g + (groups - k) g / 2 SIMDs.  k = the most that fits (64 chains, prefetching: g = 396,
    # k = 2; 256 chains, one iteration per launch: g = 516, k = 1); nothing fits even with every solve doubled up (512 chains:
    # g = 1026): throughput-bound, the groups stay as they are and every launch runs two per SIMD.  A latency-bound ensemble
    # deals its chains to the groups by cost (setup(dealt = k), pt.PTEnsemble.regroup_by_cost).
    g_solves = -(-(hi - lo) // groups) * 6 * (3 if prefetch else 1)
    k_one = max(0, min(groups, int(2 * slots * 1.02 / max(g_solves, 1) - groups)))  # (a dozen solves over is still a fit)
    if prefetch and groups > 1 and k_one == 0:
        k_one = 1  # (128 chains, prefetching: 2304 solves in flight, more than fit either way; measured: k = 1 2.86 ms per
        #            iteration, k = 2 2.92, interleaved groups one per SIMD 3.00)
    k_one = int(os.environ.get("MTRPL_PT_ONE_GROUPS", k_one))
    fits = groups * g_solves <= 2 * slots * 1.15
    dealt = k_one if (groups > 2 and decide == "device" and fits and 0 < k_one < groups
                      and os.environ.get("MTRPL_PT_NO_REGROUP") != "1") else None
    ctx = L.default_context(local)
    restore = {}
    two_per_simd = False  # every group's launches on the two-solves-per-SIMD build (said per launch, by the engines)

    def set_option(key, value):
        restore.setdefault(key, ctx.get_option(key))
        ctx.set_option(key, value)
    if groups > 1 and not prefetch:
        # several groups, one iteration per launch.  Throughput-bound: every launch, whatever its size, runs two solves per
        # SIMD, and a group's likelihood launch has to find room among the other groups' queued solves -- blocks of one
        # wave do, blocks of twelve wait until that queue has drained (include/metrotrpl_amd.h, loglik_wave_blocks).
        # Latency-bound (dealt): each group's evaluator says which build its launches use, and there is room for the
        # twelve-wave blocks (256 chains: 4.24 ms per iteration against 4.50 with blocks of one wave, 4.84 not dealt)
        two_per_simd = dealt is None
        set_option("loglik_wave_blocks", float(os.environ.get("MTRPL_LL_WAVE", "0" if dealt is not None else "1")))
    for kv in getattr(a, "opt", []) or []:
        k_, v_ = kv.split("=")
        set_option(k_, float(v_))
    try:
        return _run(a, rank, world, local, dev, cpu_baseline_fn, n_chains, prefetch, groups, decide, dealt, two_per_simd)
    finally:
        for k_, v_ in restore.items():  # the context is the process's default one: other bench legs use it too
            ctx.set_option(k_, v_)


def _run(a, rank, world, local, dev, cpu_baseline_fn, n_chains, prefetch, groups, decide, dealt, two_per_simd=False):
    import torch
    import torch.distributed as dist
    # the order the solves of a launch are issued in: the small launches of the prefetching ensemble (launches of a few
    # hundred solves from several chain groups share 1024 SIMDs: what has to wait for a SIMD should be the cheapest
    # solves) chain by chain, the throughput-bound launches solve by solve, both by the integrator's cost in the
    # previous launch
    small = prefetch and groups > 1
    S = setup(n_chains, dev, local, sort_by_cost=os.environ.get("MTRPL_NO_SORT") != "1",
              parts=int(os.environ.get("MTRPL_PARTS", "2" if groups == 1 else "1")), groups=groups,
              packed_order="chains" if (small or dealt is not None) else "solves", dealt=dealt, two_per_simd=two_per_simd)
    eng, ens0, T, proposer = S["eng"], S["ens0"], S["T"], S["proposer"]
    # history-free draws (pt.PTEnsemble "counter" mode): the chain walked does not depend on N, on prefetching
    # or on the chain groups
    # a ONE-rank ensemble that is not asked to rehearse the collectives (use_dist) never touches the process group it
    # may happen to live in: bench.py runs such legs on rank 0 of an N-rank job while the other ranks wait
    local_only = world == 1 and not getattr(a, "use_dist", False)
    ens = pt.PTEnsemble(ens0["chain_states"], T, S["evaluators"] if groups > 1 else S["evaluator"], rank=rank,
                        world=world, rng_mode="counter", streams=S["streams"], group=pt.LOCAL_ONLY if local_only else None,
                        device_decisions=decide == "device", gc_freeze=True,  # (the benchmark owns its process)
                        regroup_by_cost=dealt is not None)
    ens.initialise()
    temper_freq = 10
    ens.advance(proposer, a.warmup, temper_freq, prefetch)
    torch.cuda.synchronize()
    gc.collect()
    gc.freeze()  # what exists now is not garbage: later full collections do not walk it (torch alone is ~1e6 objects)
    timing = []
    for e_ in S["engines"]:
        e_.timing = timing
    if getattr(a, "use_dist", world > 1):
        dist.barrier()
    torch.cuda.synchronize()
    coll0 = ens.n_collectives
    ex0 = dict(ens.exchange_s)
    if os.environ.get("MTRPL_PT_PROF") == "1":
        ens.prof = {}
    t0 = time.perf_counter()
    n_acc = ens.advance(proposer, a.steps, temper_freq, prefetch)
    torch.cuda.synchronize()
    if getattr(a, "use_dist", world > 1):
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if getattr(a, "use_dist", world > 1):
        tt = torch.tensor([dt], dtype=torch.float64, device=_pt.coll_device(dev))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    stats = eng.d_stats.cpu().numpy().reshape(-1, 8)
    # the slowest solve of a rank's last launch is what that rank's iteration waits for: one figure per rank
    steps_rank = [int(stats[:, 0].max())]
    if getattr(a, "use_dist", world > 1) and world > 1:
        mine = torch.tensor(steps_rank, dtype=torch.int64, device=_pt.coll_device(dev))
        every = torch.zeros(world, dtype=torch.int64, device=_pt.coll_device(dev))
        pt.all_gather_rows(every, mine)
        steps_rank = [int(v) for v in every.cpu()]
    line = {
        "metric": "mcmc_proposals_per_sec", "value": n_chains * a.steps / dt, "unit": "proposals/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"config3/4: {n_chains} PT chains x 6-measurement staub power scan, nx=128, 1001 times, "
                               "IRF convolution (189 taps, dt 0.025 ns) + likelihood at all ladder temperatures on "
                               "device; every rank proposes and decides for its own chain block, one all-gather of [own rows x "
                               "(n_T + n_params)] per replica exchange (every 10 iterations)",
                   "chains": n_chains, "solves_per_step": n_chains * 6, "parallelism": f"chain-blocks x{world}",
                   "prefetch": bool(prefetch), "chain_groups": groups,
                   # a latency-bound ensemble: chains dealt to the groups by cost, this many groups one solve per SIMD
                   "groups_dealt_by_cost": dealt is not None, "groups_one_solve_per_simd": dealt,
                   "decisions": "device" if ens.n_pipelined else "host",
                   "collectives_per_iteration": (ens.n_collectives - coll0) / max(1, a.steps),
                   "forward_solves_per_sec": n_chains * 6 * a.steps / dt, "accepted_moves": n_acc,
                   "swap_accept_rate": float(ens.swap_accepts.sum() / max(1, ens.swap_attempts.sum())),
                   "mean_integrator_steps_rank0": float(stats[:, 0].mean()),
                   "max_integrator_steps_rank0": int(stats[:, 0].max()),
                   "max_integrator_steps_per_rank": steps_rank,
                   # what a replica exchange costs this rank's host, per exchange of the timed iterations: waiting for its
                   # groups' last launches (the latency floor: the slowest solve) | the all-gather + copy to the host (the
                   # collective) | swap replay, permutation, upload of the own rows (then the first launches go out)
                   "exchange_ms": (lambda n_: None if n_ == 0 else {
                       k_: round((ens.exchange_s[k_] - ex0[k_]) / n_ * 1e3, 4) for k_ in ("wait", "gather", "replay")})(
                       ens.exchange_s["n"] - ex0["n"]),
                   "exchanges_timed": ens.exchange_s["n"] - ex0["n"]},
    }
    if ens.prof is not None:
        line["config"]["host_ms"] = {k: round(v * 1e3, 3) for k, v in ens.prof.items()}
    if rank == 0:
        # roofline of the dominant kernel (the forward solve) from the HIP events recorded around every
        # solve launch of the timed iterations, on the stream it was launched on
        A = algorithmic_bytes(128, len(S["times"]), irf_nk=80000)
        t_solve = [e0.elapsed_time(e1) for e0, e1, e2, n in timing]
        t_like = [e1.elapsed_time(e2) for e0, e1, e2, n in timing]
        n_solve = [n for *_, n in timing]
        kern_ms = float(np.mean(t_solve))
        # launches of more than 1024 solves go to the two-solves-per-SIMD form of the wave kernel (include/metrotrpl_amd.h)
        two = eng.per_simd == 2 or (eng.per_simd == 0 and
                                    float(np.mean(n_solve)) >= L.default_context(local).get_option("two_per_simd_min_batch"))
        kern_name = "solve_kernel_wave2" if two else "solve_kernel_wave1"
        achieved = A * float(np.mean(n_solve)) / (kern_ms * 1e-3) / 1e9
        line["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBS, "kernel": kern_name, "kernel_ms": kern_ms,
                            "solves_per_launch": float(np.mean(n_solve)), "launches_per_step": len(timing) / max(1, a.steps),
                            "algorithmic_bytes_per_solve": A, "loglik_kernel_ms": float(np.mean(t_like)),
                            **pmc_fields(float(np.mean(n_solve)), kern_name),
                            "note": "state-streaming model of SURVEY.md 8d incl. the 8*(3*nk+2) IRF term; the solve state "
                                    "is register-resident by design, so HBM is not the limiter (DESIGN.md section 4)"}
        if cpu_baseline_fn is not None and world == 1 and not getattr(a, "no_cpu_baseline", False):
            line["cpu_baseline"] = cpu_baseline_fn(S)  # the oracle lives outside the package (bench.py)
    return line


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(nx, nt, model="std", irf_nk=0):
    """A = nt_out * 8 * (2S + 1), S = 3nx+1 (std) / 4nx+1 (traps); + 8*(3*nk+2) with IRF on."""
    S_ = (3 if model == "std" else 4) * nx + 1
    a = nt * 8 * (2 * S_ + 1)
    if irf_nk:
        a += 8 * (3 * irf_nk + 2)
    return a


def pmc_fields(solves_per_launch=1024, kernel="solve_kernel_wave1"):
    """Counter-derived figures of the solve kernel from the committed rocprofv3 --pmc summaries (they are not
    measured in this run; traffic_source says where they come from).  The traffic is per solve -- descriptor
    in, samples out -- and is scaled from the profiled launch to `solves_per_launch`."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = "hbm_traffic_wave2.json" if kernel.endswith("wave2") else "hbm_traffic.json"
    out = {"traffic": None, "traffic_source": None}
    try:
        z = json.load(open(os.path.join(repo, "profiles", name)))
        out["traffic"] = z.get("solve_kernel_bytes_per_launch")
        if out["traffic"] is not None:
            out["traffic"] = out["traffic"] * solves_per_launch / float(z.get("solves_per_launch", 1024))
        out["traffic_source"] = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of {kernel}, one launch at a time)"
        d = z.get("derived", {})
        out["valu_active_frac"] = d.get("valu_active_fraction_of_wave_cycles")
        out["occupancy_waves_per_simd"] = d.get("occupancy_waves_per_simd", 1)
    except Exception:
        pass
    return out


def pmc_file_fields(suffix, solves_per_launch):
    """traffic / occupancy figures of a kernel from the newest committed profiles/rNN_z_<suffix> (rocprofv3 --pmc
    passes; not measured in this run)."""
    import glob
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {"traffic": None, "traffic_source": None}
    found = sorted(glob.glob(os.path.join(repo, "profiles", "r*_z_" + suffix)))
    if not found:
        return out
    try:
        z = json.load(open(found[-1]))
        t = z.get("solve_kernel_bytes_per_launch")
        if t is not None:
            out["traffic"] = t * solves_per_launch / float(z.get("solves_per_launch", solves_per_launch))
        out["traffic_source"] = f"profiles/{os.path.basename(found[-1])} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, one launch at a time)"
        d = z.get("derived", {})
        out["valu_active_frac"] = d.get("valu_active_fraction_of_wave_cycles")
        out["parked_on_waitcnt_frac"] = d.get("parked_on_waitcnt_fraction_of_wave_cycles")
        out["occupancy_waves_per_simd"] = d.get("occupancy_waves_per_simd")
    except Exception:
        pass
    return out


N_SIMD, CLOCK_HZ = 1024, 2.4e9  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz max clock


# Issue cost of the instruction classes with two waves per SIMD, cycles (at 2.4 GHz) per wave-instruction per SIMD, measured
# on this device by tools/mb/rcp_rate.hip (profiles/r06_z_issue_cost.txt, two waves per SIMD: v_add_f64 1.93 ns, v_fma_f64
# 2.06 ns, v_rcp_f64 7.3 ns, v_fma_f32 1.20 ns, v_pk_fma_f32 1.95 ns per wave-op per SIMD).
# ROUND 6 CORRECTION: rounds 3-5 priced v_fma_f64 at 5.8 and the fp32 classes at 4.0 -- their micro-benchmark ran eight
# operations per trip of its loop and counted the loop's own compare-and-branch (about 20 cycles a trip) into them; with
# 128 operations per trip an fp64 FMA costs 4.9 cycles and a wave ALONE on its SIMD issues one every 5.2 (not "every 8":
# what a lone wave of the solve kernel loses is dependency and LDS / scalar-memory latency, not issue width).  The bound
# is higher and the measured rate a smaller fraction of it than rounds 3-5 reported (config 2: 893 k solves/s and 76 %
# where they said 829 k and 82-84 %).
# (fp32 classes: scalar fp32 2.9 cycles, packed fp32 -- counted in the same classes -- 4.7; 3.5 is the blend at the kernel's
# static ratio of one packed instruction in three.  "other" -- moves, selects, conversions, DPP, lane reads -- is not
# classed by the counters: a 32-bit move costs 2.4, an fp64 conversion 4.5, a DPP move behind its wait state 6.5; 4.0 kept.)
ISSUE_COST_2WAVES = {"SQ_INSTS_VALU_ADD_F64": 4.6, "SQ_INSTS_VALU_FMA_F64": 4.9, "SQ_INSTS_VALU_MUL_F64": 4.9,
                     "SQ_INSTS_VALU_TRANS_F64": 17.0, "SQ_INSTS_VALU_ADD_F32": 3.5, "SQ_INSTS_VALU_FMA_F32": 3.5,
                     "SQ_INSTS_VALU_MUL_F32": 3.5, "SQ_INSTS_VALU_TRANS_F32": 8.0, "other": 4.0}


def _issue_cost_file(repo):
    """The newest committed run of the issue-cost micro-benchmark (profiles/rNN_z_issue_cost.txt)."""
    import glob
    found = sorted(glob.glob(os.path.join(repo, "profiles", "r*_z_issue_cost.txt")))
    return "profiles/" + os.path.basename(found[-1]) if found else "no committed run"


def valu_issue_fields(kernel, solves_per_sec_per_gpu):
    """The binding roofline of the solve kernel: vector-ALU issue.  Instructions and VALU-busy cycles per solve
    come from the committed rocprofv3 --pmc SQ_* pass of the named kernel (not measured in this run); frac = the
    share of the device's SIMD-cycles (1024 SIMDs x 2.4 GHz nominal) the measured solve rate keeps the vector
    ALUs busy BY THE COUNTER, which books four cycles per instruction.  An fp64 instruction occupies the pipeline
    longer than that (ISSUE_COST_2WAVES): `throughput` prices the kernel's own instruction mix (a second --pmc pass,
    profiles/hbm_traffic_mix_wave2.json) with the measured costs -- the time the SIMD needs to issue one solve's
    instructions with nothing ever waiting -- and frac_of_bound is the measured rate against that."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = "hbm_traffic_wave2.json" if kernel.endswith("wave2") else "hbm_traffic.json"
    try:
        z = json.load(open(os.path.join(repo, "profiles", name)))
        sq, d = z["sq_counters_per_launch"], z["derived"]
        busy = 4.0 * sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVES"]  # SQ_ACTIVE_* count quad-cycles
        out = {"insts_per_solve": d["valu_insts_per_solve"], "busy_cycles_per_solve": busy,
               "frac": solves_per_sec_per_gpu * busy / (N_SIMD * CLOCK_HZ), "peak": "1024 SIMDs x 2.4 GHz",
               "source": f"profiles/{name} (rocprofv3 --pmc SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_WAVES of {kernel}); "
                         "the solve rate is this run's"}
        if kernel.endswith("wave2"):
            try:
                mix = json.load(open(os.path.join(repo, "profiles", "hbm_traffic_mix_wave2.json")))["per_solve"]
                classed = sum(v for k, v in mix.items() if k != "SQ_INSTS_VALU")
                cycles = sum(ISSUE_COST_2WAVES[k] * v for k, v in mix.items() if k != "SQ_INSTS_VALU")
                cycles += ISSUE_COST_2WAVES["other"] * (mix["SQ_INSTS_VALU"] - classed)
                out["throughput"] = {
                    "fp64_insts_per_solve": sum(v for k, v in mix.items() if k.endswith("_F64")),
                    "fp32_insts_per_solve": sum(v for k, v in mix.items() if k.endswith("_F32")),
                    "other_insts_per_solve": mix["SQ_INSTS_VALU"] - classed,
                    "issue_cycles_per_solve": cycles,
                    "bound_solves_per_sec": N_SIMD * CLOCK_HZ / cycles,
                    "frac_of_bound": solves_per_sec_per_gpu * cycles / (N_SIMD * CLOCK_HZ),
                    "source": "profiles/hbm_traffic_mix_wave2.json (instruction classes, rocprofv3 --pmc) x the issue costs "
                              f"measured by tools/mb/rcp_rate.hip with two waves per SIMD ({_issue_cost_file(repo)})"}
            except Exception as e:
                out["throughput"] = {"error": repr(e)}
        return out
    except Exception as e:
        return {"error": repr(e)}
