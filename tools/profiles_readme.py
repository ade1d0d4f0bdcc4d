"""Rewrite the GENERATED block of profiles/README.md for one round's profile set from the files themselves
(tools/collect_profiles.sh calls this: the numbers the README quotes are read, not typed):
    python tools/profiles_readme.py r04_z"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
T = sys.argv[1]


def load(name):
    fn = os.path.join(P, f"{T}_{name}")
    if not os.path.exists(fn):
        return None
    txt = open(fn).read().strip()
    return json.loads(txt.splitlines()[-1]) if txt else None


def kstats(name, kern):
    fn = os.path.join(P, f"{T}_{name}")
    if not os.path.exists(fn):
        return None
    for r in csv.DictReader(open(fn)):
        if kern in r["Name"]:
            return int(r["Calls"]), float(r["AverageNs"]) / 1e6
    return None


rows = []
b = load("bench_config2.json")
if b:
    c = b["config"]
    legs = []
    for k in ("single_stream", "sorted_by_measured_cost", "fp64_linear_algebra"):
        if b.get(k):
            legs.append(f"`{k}` {b[k]['solves_per_sec'] / 1e3:.1f} k ({b[k]['ms_per_step']:.2f} ms)")
    if (b.get("fp64_linear_algebra") or {}).get("streams"):
        legs.append(f"`fp64_linear_algebra.streams` {b['fp64_linear_algebra']['streams']['solves_per_sec'] / 1e3:.1f} k")
    for k, v in (b.get("one_launch") or {}).items():
        legs.append(f"`one_launch[{k}]` {v['solves_per_sec'] / 1e3:.1f} k ({v['ms_per_launch']:.2f} ms)")
    pt = b.get("pt") or {}
    ptl = ", ".join(f"{k} chains {v['ms_per_iteration']:.2f} ms / iteration" for k, v in pt.items() if "ms_per_iteration" in v)
    acc = b.get("accuracy") or {}
    api = b.get("api") or {}
    if api.get("solve_batch"):
        legs.append(f"`api.solve_batch` {api['solve_batch']['pinned_out']['solves_per_sec'] / 1e3:.1f} k host to host (pageable output "
                    f"{api['solve_batch']['pageable_out']['solves_per_sec'] / 1e3:.1f} k), `api.dense_sampler` {api['dense_sampler']['samples_per_sec'] / 1e3:.1f} k samples/s")
    rows.append((f"`{T}_bench_config2.json`",
                 f"the `bench.py` line (K = {b['steps']}): **{b['value'] / 1e3:.1f} k solves/s**, {b['ms_per_step']:.3f} ms per 1024-solve step, "
                 f"{c['streams']} streams as drawn; mean / max integrator steps {c['mean_integrator_steps']:.0f} / {c['max_integrator_steps']}; "
                 + "; ".join(legs) + (f"; `pt`: {ptl}" if ptl else "")
                 + (f"; `accuracy` worst decade 0-8 {max(acc['max_rel_err_per_decade_0_to_8']):.2e}" if acc.get("max_rel_err_per_decade_0_to_8") else "")
                 + f"; `roofline.frac` {b['roofline']['frac']:.3f} (kernel {b['roofline']['kernel_ms']:.2f} ms), `cpu_baseline` "
                 + (f"{b['cpu_baseline']['value']:.2f} solves/s on 1 core" if b.get("cpu_baseline") else "not run")))
b = load("bench_config2_k20.json")
if b:
    rows.append((f"`{T}_bench_config2_k20.json`", f"the same build over the K = 20 steps the driver times: {b['value'] / 1e3:.1f} k solves/s"))
for name, kern, what in (("kernel_stats_bench_config2.csv", "solve_kernel_wave2", "four launches in flight"),
                         ("kernel_stats_bench_config2_one_stream.csv", "solve_kernel_wave1", "one launch at a time"),
                         ("kernel_stats_bench_config5.csv", "solve_kernel_spike", "config 5, nx = 512"),
                         ("kernel_stats_bench_pt512.csv", "solve_kernel_wave", "512-chain PT"),
                         ("kernel_stats_bench_pt64.csv", "solve_kernel_wave", "64-chain PT")):
    r = kstats(name, kern)
    if r:
        rows.append((f"`{T}_{name}`", f"`rocprofv3 --kernel-trace --stats`: {r[0]} launches of `{kern}...`, average {r[1]:.2f} ms ({what})"))
for name, what in (("pmc_wave1.json", "wave1"), ("pmc_wave2.json", "wave2"), ("pmc_spike4.json", "spike<4>"), ("pmc_loglik.json", "loglik")):
    fn = os.path.join(P, f"{T}_{name}")
    if os.path.exists(fn):
        z = json.load(open(fn))
        d = z.get("derived", {})
        rows.append((f"`{T}_{name}`", f"`rocprofv3 --pmc` passes, {what}: {d.get('valu_insts_per_solve', 0) / 1e6:.3f} M VALU instructions per solve, VALU issuing "
                                     f"{100 * (d.get('valu_active_fraction_of_wave_cycles') or 0):.0f} % / parked on `s_waitcnt` "
                                     f"{100 * (d.get('parked_on_waitcnt_fraction_of_wave_cycles') or 0):.0f} % of the wave cycles, HBM "
                                     + (f"{z['solve_kernel_bytes_per_launch'] / 1e6:.2f} MB per launch" if z.get("solve_kernel_bytes_per_launch") else "traffic n/a")))
b = load("bench_config5.json")
if b:
    rows.append((f"`{T}_bench_config5.json`", f"`bench.py --workload config5`: **{b['value'] / 1e3:.1f} k solves/s** ({b['ms_per_step']:.2f} ms per 512-solve step, "
                                               f"{b['config']['streams']} streams), single stream {b['single_stream']['solves_per_sec'] / 1e3:.1f} k; kernel `{b['roofline']['kernel']}`"))
for name in ("bench_pt512_1gpu.json", "bench_pt256_1gpu.json", "bench_pt128_1gpu.json", "bench_pt64_1gpu.json", "bench_pt512_1gpu_host.json",
             "bench_pt64_1gpu_host.json", "bench_pt64_1gpu_plain.json"):
    b = load(name)
    if b:
        c = b["config"]
        dealt = f", chains dealt by cost, {c['groups_one_solve_per_simd']} group(s) one solve per SIMD" if c.get("groups_dealt_by_cost") else ""
        rows.append((f"`{T}_{name}`", f"`bench.py --workload pt`: {c['chains']} chains, {b['ms_per_step']:.2f} ms per iteration = {b['value'] / 1e3:.1f} k proposals/s "
                                      f"(prefetch {c['prefetch']}, {c['chain_groups']} group(s){dealt}, decisions on the {c.get('decisions', 'host')})"))
for name, what in (("tail_decades.txt", "per-decade accuracy table (config 2 / config-3 geometry)"), ("heldout.txt", "the held-out converged set, per decade"),
                   ("heldout_large.txt", "the nx > 128 held-out converged set, per decade, through the SPIKE and the team backend"),
                   ("bind_trace.txt", "which half of the goal-oriented error test sets the step size, per decade of time (diagnostic build)"),
                   ("pt_chain_granularity.txt", "what a chain-granular PT pipeline could buy: sum over launches of the slowest chain against the slowest chain's own sum"),
                   ("api_streams.txt", "`mtrpl_solve_batch` host to host at B = 8192: sub-launch size x streams; and inside processes that own other streams, with 8 and 16 hardware queues"),
                   ("startup_trace.txt", "where the steps of a solve go along the time axis"), ("slow_solves.txt", "the slowest solves of the config-2 batch"),
                   ("variant_agreement.txt", "default kernel vs fp64-block kernel on 8192 draws"), ("solve_latency.txt", "latency of the drop-in `solve()`"),
                   ("config5_sweep.txt", "config 5: fp32 / fp64 tolerance sweep on this build"), ("issue_cost.txt", "issue cost of the instruction classes (`tools/mb/rcp_rate.hip`; from round 6 on with 128 operations per trip of its loop -- rounds 3-5 counted the loop's own branch into eight)"),
                   ("xchg_cost.txt", "what an LDS exchange between the waves of a workgroup costs (`tools/mb/xchg_cost.hip`)"),
                   ("soak.txt", "`tools/soak.py`: every solve kernel N times over the config-2 / config-5 batches, bytes and step counts repeat"),
                   ("step_phases.txt", "cycles per integrator step and phase (`-DMTRPL_STAMPS` diagnostic build): SPIKE at nx = 512, the one-wave kernel one and two per SIMD"),
                   ("pt64_timeline.txt", "64-chain PT from a `rocprofv3 --kernel-trace`: per hardware queue (= chain group) the solve / likelihood kernel times and the gap from one solve to the next; per stretch between two exchanges when each group starts and ends"),
                   ("pt512_timeline.txt", "the same for the 512-chain ensemble"),
                   ("pt64_host_phases.txt", "64-chain PT: host milliseconds per launch cycle and phase, decisions on the device (three groups) and on the host (two)")):
    if os.path.exists(os.path.join(P, f"{T}_{name}")):
        rows.append((f"`{T}_{name}`", what))

block = [f"<!-- generated:{T} (tools/profiles_readme.py: every number below is read from the file it stands next to) -->",
         f"Round {int(T[1:3])} (`{T}_*`), all on one MI355X through `gpurun`, `tools/final_profiles.sh {T}` after the last code commit of the round:", "",
         "| file | what (generated) |", "|---|---|"] + [f"| {a} | {b_} |" for a, b_ in rows] + [f"<!-- /generated:{T} -->"]
fn = os.path.join(P, "README.md")
txt = open(fn).read()
pat = re.compile(rf"<!-- generated:{T} .*?<!-- /generated:{T} -->", re.S)
new = "\n".join(block)
txt = pat.sub(lambda m: new, txt) if pat.search(txt) else txt.rstrip("\n") + "\n\n" + new + "\n"
open(fn, "w").write(txt)
print(new)
