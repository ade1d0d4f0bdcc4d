"""GPU: the contract line of `python bench.py` as the driver will get it, with the N > 1 code path rehearsed on one GPU
(MTRPL_FORCE_DIST=1: a one-rank `nccl` group, so every barrier, the max-over-ranks all-reduce and the PT all-gather go
through RCCL) -- VERDICT r4 next 6b.  bench.py runs as a CHILD process (it owns its stdout protocol and its GPU set-up)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_through_a_one_rank_process_group():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MTRPL_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, cwd=REPO, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-500:]  # ONE line on stdout, whatever RCCL and the launcher print
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "trpl_forward_solves_per_sec" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["all_solves_ok"] is True
    assert abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r_ = d["roofline"]
    assert r_["bound"] == "hbm" and r_["unit"] == "GB/s" and abs(r_["frac"] - r_["achieved"] / r_["peak"]) < 1e-12
    assert r_["kernel_ms_min"] <= r_["kernel_ms"] <= r_["kernel_ms_max"]
    assert "r03" not in r_["valu_issue"]["throughput"]["source"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert c["all_cores"]["cores"] >= 1 and c["all_cores"]["cores_available"] >= c["all_cores"]["cores"]
    # the product API, host to host
    api = d["api"]
    assert api["solve_batch"]["pinned_out"]["all_solves_ok"] and api["solve_batch"]["pageable_out"]["all_solves_ok"]
    assert api["solve_batch"]["solves_per_sec"] > 2e5
    assert api["dense_sampler"]["samples_per_sec"] > 2e4 and api["dense_sampler"]["one_chunk_at_a_time"]["same_values"] is True
    # BASELINE's second metric: both ensembles, through the process group
    for chains in ("64", "512"):
        p = d["pt"][chains]
        assert p["chains"] == int(chains) and p["ms_per_iteration"] > 0, p
        assert abs(p["proposals_per_sec"] - int(chains) / (p["ms_per_iteration"] * 1e-3)) < 1e-6 * p["proposals_per_sec"]
        assert 0 < p["collectives_per_iteration"] <= 0.2  # one all-gather per replica exchange (every 10 iterations), through RCCL
        assert len(p["max_integrator_steps_per_rank"]) == 1 and p["max_integrator_steps_per_rank"][0] > 100
        ex = p["exchange_ms"]
        assert p["exchanges_timed"] >= 1 and all(ex[k] >= 0 for k in ("wait", "gather", "replay"))
    assert d["pt"]["64"]["cpu_baseline"]["kind"] == "port"
    assert d["accuracy"]["all_ok"] is True and max(d["accuracy"]["max_rel_err_per_decade_0_to_8"]) < 1e-6


def test_two_rank_bench_line_rehearsed_on_one_gpu():
    """The N > 1 control flow of bench.py -- rank 0's CPU baseline before its first GPU call while the other rank waits in
    init_process_group, the weak-scaling region with its barriers and max-over-ranks, the rank-0-only legs (api, accuracy,
    the 64-chain and one-GPU 512-chain ensembles) with every rank leaving them through one all-reduce, the 512-chain ensemble
    block-partitioned over the ranks with its all-gather at every exchange, the assembly of the N > 1 line -- has never run on an
    8-GPU node.  Here it runs as TWO ranks that share device 0 through a `gloo` group (MTRPL_REHEARSE_ONE_GPU=1: the numbers
    mean nothing, the code path is the driver's): the line must come out, with the N > 1 fields in place."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MTRPL_REHEARSE_ONE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, cwd=REPO, timeout=1100, env=env)
    if r.returncode != 0 and os.path.isdir(os.path.join(REPO, "gpurun_out")):
        open(os.path.join(REPO, "gpurun_out", "bench_rehearsal.err"), "w").write(r.stderr)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["all_solves_ok"] is True
    assert abs(d["value"] - 2 * 1024 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]  # whole-job rate: both ranks' solves
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0     # at N > 1 too
    assert "roofline" in d and "api" in d and d["accuracy"]["all_ok"] is True
    assert d["multi_gpu"]["first_run_on_hardware"] is True  # (the N > 1 line says that no multi-GPU hardware had run it)
    p = d["pt"]["512"]
    assert p["n_gpus"] == 2 and p["scaling"] == "strong" and p["chains"] == 512 and p["ms_per_iteration"] > 0
    assert 0 < p["collectives_per_iteration"] <= 0.2 and len(p["max_integrator_steps_per_rank"]) == 2
    assert p["exchanges_timed"] >= 1 and p["exchange_ms"]["gather"] > 0
    assert p["one_gpu_same_build"]["n_gpus"] == 1 and p["speedup_vs_one_gpu"] > 0
    assert d["pt"]["64"]["n_gpus"] == 1 and d["pt"]["64"]["ms_per_iteration"] > 0
