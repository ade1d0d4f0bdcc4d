"""CPU-only checks of bench.py's bookkeeping (the numbers SURVEY.md 8d fixes)."""
import importlib.util
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_survey():
    b = _bench()
    assert b.algorithmic_bytes_per_solve(128, 1001) == 6174168          # 6.17 MB
    assert round(b.algorithmic_bytes_per_solve(100, 1001) / 1e6, 2) == 4.83
    assert round(b.algorithmic_bytes_per_solve(512, 1001) / 1e6, 1) == 24.6
    assert round(b.algorithmic_bytes_per_solve(128, 1001, model="traps") / 1e6, 2) == 8.22
    assert b.algorithmic_bytes_per_solve(128, 1001, irf_nk=80000) - b.algorithmic_bytes_per_solve(128, 1001) == 1920016
    assert b.HBM_PEAK_GBS == 8000.0


def test_cpu_baseline_shape():
    b = _bench()
    out = b.cpu_baseline(sample=2)
    assert out["kind"] == "port" and out["cores"] == 1 and out["unit"] == "solves/s" and out["value"] > 0
    assert "sample" in out and ("value" in out["all_cores"] or "error" in out["all_cores"])


def test_committed_bench_line_has_the_contract_fields():
    """profiles/r01_m_bench_config2.json is a real `python bench.py` line from the MI355X box."""
    import json
    d = json.load(open(os.path.join(REPO, "profiles", "r01_m_bench_config2.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "trpl_forward_solves_per_sec" and d["dtype"] == "f64" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch / kernel time
    assert abs(r["achieved"] - 6174168 * 1024 / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] and r["traffic"] < 2 * 8531784  # PMC bytes per launch close to the compulsory figure
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "solves/s" and "sample" in c
    assert d["value"] > 1e4  # north-star target: >= 10k solves/s on one GPU


def test_round3_bench_line_tells_the_whole_story():
    """The newest committed round-3 line (tools/final_profiles.sh on the MI355X box): the headline is measured as drawn,
    and the single-stream, unordered, oracle-ordered, fp64-linear-algebra and PT figures ride along as top-level keys
    (VERDICT r2 item 2)."""
    import glob
    import json
    path = sorted(glob.glob(os.path.join(REPO, "profiles", "r03_?_bench_config2.json")))[-1]
    d = json.loads(open(path).read().strip().splitlines()[-1])
    assert d["metric"] == "trpl_forward_solves_per_sec" and d["dtype"] == "f64" and d["vs_baseline"] is None
    cfg = d["config"]
    assert cfg["as_drawn"] is True and cfg["all_solves_ok"] is True and "model" not in cfg
    assert cfg["solves_per_gpu_per_step"] == 1024 and cfg["nx"] == 128 and cfg["nt_out"] == 1001
    for k in ("single_stream", "as_given", "sorted_by_measured_cost", "fp64_linear_algebra"):
        leg = d[k]
        assert leg["all_solves_ok"] is True and leg["solves_per_sec"] > 1e5, k
        assert abs(leg["solves_per_sec"] - 1024 / (leg["ms_per_step"] * 1e-3)) < 1e-6 * leg["solves_per_sec"], k
    assert abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # the headline is not the oracle-ordered figure, and ordering by the model sits between none and the measured one
    assert d["as_given"]["solves_per_sec"] * 0.97 < d["value"] < d["sorted_by_measured_cost"]["solves_per_sec"] * 1.03
    assert d["single_stream"]["solves_per_sec"] > d["fp64_linear_algebra"]["solves_per_sec"]
    assert "fp32" in d["linear_algebra"] and "fp64" in d["linear_algebra"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - 6174168 * 1024 / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    v = r["valu_issue"]
    assert 0.3 < v["frac"] < 1.0 and abs(v["frac"] - v["busy_cycles_per_solve"] * d["value"] / (1024 * 2.4e9)) < 1e-6
    assert 5e5 < v["insts_per_solve"] < 1.2e6
    for chains in ("64", "512"):
        p = d["pt"][chains]
        assert p["chains"] == int(chains) and p["ms_per_iteration"] > 0 and p["collectives_per_iteration"] <= 1.0
        assert abs(p["proposals_per_sec"] - int(chains) / (p["ms_per_iteration"] * 1e-3)) < 1e-6 * p["proposals_per_sec"]
    assert d["pt"]["64"]["cpu_baseline"]["kind"] == "port"
    acc = d["accuracy"]
    assert acc["all_ok"] is True and max(acc["max_rel_err_per_decade_0_to_8"]) < 1.01e-6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "solves/s"


def test_without_a_gpu_bench_fails_loudly_and_prints_no_line():
    """The HIP path has no CPU fallback: without a device bench.py says so on stderr, exits non-zero and leaves stdout
    empty (the only thing it ever writes there is the one contract line)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a machine without a GPU")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--no-cpu-baseline"], capture_output=True, text=True,
                       cwd=REPO, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-300:])
    assert r.stdout == ""
    assert "no GPU visible" in r.stderr


def test_gpus_2_without_a_launcher_starts_nothing_on_a_gpu_less_box():
    """`python bench.py --gpus N` (WORLD_SIZE unset) starts its N ranks itself as a child torch.distributed.run -- before
    this process makes any GPU call, never an exec.  Without N visible GPUs nothing is started: exit code 3, the reason
    on stderr, nothing on stdout (the bare command used to exit 2 with 'launch with torch.distributed.run')."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a machine with fewer than two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, cwd=REPO, timeout=300, env=env)
    assert r.returncode == 3, (r.returncode, r.stderr[-300:])
    assert r.stdout == ""
    assert "--gpus 2" in r.stderr and "GPU(s) visible" in r.stderr


def test_round4_bench_line_tells_the_whole_story():
    """The newest committed round-4 line (tools/final_profiles.sh on the MI355X box): measured as drawn with no ordering
    kernel in front of a launch, the kernel time brackets the solve kernel alone and says so, and the figures VERDICT r3
    asked for ride along: one launch at a time at 1024 / 2048 / 4096 solves, the all-fp64 kernel on one and on four
    streams, both parallel-tempering ensembles with the slowest solve per rank."""
    import glob
    import json
    path = sorted(glob.glob(os.path.join(REPO, "profiles", "r04_?_bench_config2.json")))[-1]
    d = json.loads(open(path).read().strip().splitlines()[-1])
    assert d["metric"] == "trpl_forward_solves_per_sec" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["n_gpus"] == 1 and d["data"] == "synthetic"
    cfg = d["config"]
    assert cfg["as_drawn"] is True and cfg["all_solves_ok"] is True and "model" not in cfg and "as drawn" in cfg["issue_order"]
    assert cfg["solves_per_gpu_per_step"] == 1024 and cfg["nx"] == 128 and cfg["nt_out"] == 1001
    assert cfg["mean_integrator_steps"] <= 900 and cfg["max_integrator_steps"] <= 1400
    assert "as_given" not in d  # the headline IS as given
    for k in ("single_stream", "sorted_by_measured_cost", "fp64_linear_algebra"):
        leg = d[k]
        assert leg["all_solves_ok"] is True and leg["solves_per_sec"] > 1e5, k
        assert abs(leg["solves_per_sec"] - 1024 / (leg["ms_per_step"] * 1e-3)) < 1e-6 * leg["solves_per_sec"], k
    assert abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["single_stream"]["solves_per_sec"] < d["value"] < d["sorted_by_measured_cost"]["solves_per_sec"]
    f64 = d["fp64_linear_algebra"]
    assert f64["streams"]["all_solves_ok"] is True and f64["solves_per_sec"] < d["single_stream"]["solves_per_sec"]
    assert f64["streams"]["solves_per_sec"] < d["value"]
    for b2 in ("2048", "4096"):
        o = d["one_launch"][b2]
        assert o["all_solves_ok"] is True and o["solves_per_launch"] == int(b2)
        assert d["single_stream"]["solves_per_sec"] < o["solves_per_sec"] < d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - 6174168 * 1024 / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert "solve kernel alone" in r["kernel_ms_brackets"] and r["kernel"] == "solve_kernel_wave2"
    assert r["traffic"] and r["traffic"] < 2 * 8531784 * 1.2  # PMC bytes per launch close to the compulsory figure
    v = r["valu_issue"]
    assert 0.3 < v["frac"] < 1.0 and abs(v["frac"] - v["busy_cycles_per_solve"] * d["value"] / (1024 * 2.4e9)) < 1e-6
    assert 5e5 < v["insts_per_solve"] < 8e5 and 0.5 < v["throughput"]["frac_of_bound"] < 1.0
    for chains in ("64", "512"):
        p = d["pt"][chains]
        assert p["chains"] == int(chains) and p["ms_per_iteration"] > 0 and p["collectives_per_iteration"] <= 1.0
        assert abs(p["proposals_per_sec"] - int(chains) / (p["ms_per_iteration"] * 1e-3)) < 1e-6 * p["proposals_per_sec"]
        assert len(p["max_integrator_steps_per_rank"]) == 1
    assert d["pt"]["64"]["cpu_baseline"]["kind"] == "port"
    acc = d["accuracy"]
    assert acc["all_ok"] is True and max(acc["max_rel_err_per_decade_0_to_8"]) < 1e-6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "solves/s" and c["value"] > 0
    # ... and config 5 has a line of its own, on the SPIKE kernel
    c5 = json.loads(open(path.replace("bench_config2", "bench_config5")).read().strip().splitlines()[-1])
    assert c5["config"]["nx"] == 512 and c5["config"]["all_solves_ok"] is True and "spike" in c5["roofline"]["kernel"]
    assert c5["single_stream"]["solves_per_sec"] < c5["value"]


def test_round5_bench_line_tells_the_whole_story():
    """The newest committed round-5 line (tools/final_profiles.sh on the MI355X box): what VERDICT r4 asked of the line itself --
    the product API timed host to host (`api`), the spread of the per-launch kernel brackets, the exchange timed in three parts,
    every host core accounted for in `cpu_baseline.all_cores`, a current source for the issue costs -- next to everything the
    round-4 line carried; and config 5's line prices its roofline with the kernel's own duration and says what it moves."""
    import glob
    import json
    path = sorted(glob.glob(os.path.join(REPO, "profiles", "r05_?_bench_config2.json")))[-1]
    d = json.loads(open(path).read().strip().splitlines()[-1])
    assert d["metric"] == "trpl_forward_solves_per_sec" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["n_gpus"] == 1 and d["data"] == "synthetic"
    cfg = d["config"]
    assert cfg["as_drawn"] is True and cfg["all_solves_ok"] is True and "model" not in cfg
    assert cfg["solves_per_gpu_per_step"] == 1024 and cfg["nx"] == 128 and cfg["nt_out"] == 1001
    assert cfg["mean_integrator_steps"] <= 760 and cfg["max_integrator_steps"] <= 1250  # (round 4: 801 / 1259)
    assert abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"] and d["value"] >= 6.5e5
    for k in ("single_stream", "sorted_by_measured_cost", "fp64_linear_algebra"):
        leg = d[k]
        assert leg["all_solves_ok"] is True and leg["solves_per_sec"] > 2e5, k
    assert d["single_stream"]["kernel_ms_min"] <= d["single_stream"]["kernel_ms"] <= d["single_stream"]["kernel_ms_max"]
    assert d["single_stream"]["solves_per_sec"] < d["value"] < d["sorted_by_measured_cost"]["solves_per_sec"]
    for b2 in ("2048", "4096"):
        assert d["one_launch"][b2]["all_solves_ok"] is True
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - 6174168 * 1024 / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["kernel_ms_min"] <= r["kernel_ms"] <= r["kernel_ms_max"] and r["kernel"] == "solve_kernel_wave2"
    assert r["traffic"] and r["traffic"] < 2 * 8531784 * 1.2  # PMC bytes per launch close to the compulsory figure
    v = r["valu_issue"]
    assert 0.3 < v["frac"] < 1.0 and 4e5 < v["insts_per_solve"] < 7e5 and 0.5 < v["throughput"]["frac_of_bound"] < 1.0
    assert "r05" in v["throughput"]["source"]  # (round 4's line still cited round 3's issue-cost file)
    # the product API, host to host (>= 450 k solves/s PCIe-inclusive, >= 70 k samples/s)
    api = d["api"]
    sb = api["solve_batch"]
    assert sb["solves"] == 8192 and sb["pinned_out"]["all_solves_ok"] and sb["pageable_out"]["all_solves_ok"]
    assert sb["pinned_out"]["solves_per_sec"] >= 4.5e5 and sb["pageable_out"]["solves_per_sec"] >= 4.0e5
    assert api["dense_sampler"]["samples_per_sec"] >= 7e4 and api["dense_sampler"]["one_chunk_at_a_time"]["same_values"] is True
    for chains in ("64", "512"):
        p = d["pt"][chains]
        assert p["chains"] == int(chains) and p["ms_per_iteration"] > 0 and p["collectives_per_iteration"] <= 1.0
        assert abs(p["proposals_per_sec"] - int(chains) / (p["ms_per_iteration"] * 1e-3)) < 1e-6 * p["proposals_per_sec"]
        assert p["exchanges_timed"] >= 1 and set(p["exchange_ms"]) == {"wait", "gather", "replay"}
    assert d["pt"]["64"]["ms_per_iteration"] <= 2.4 and d["pt"]["512"]["ms_per_iteration"] <= 6.7  # round 4's figures
    assert d["pt"]["64"]["cpu_baseline"]["kind"] == "port"
    acc = d["accuracy"]
    assert acc["all_ok"] is True and max(acc["max_rel_err_per_decade_0_to_8"]) < 1e-6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "solves/s" and c["value"] > 0
    assert c["all_cores"]["cores"] >= 1 and c["all_cores"]["cores_available"] >= c["all_cores"]["cores"]
    c5 = json.loads(open(path.replace("bench_config2", "bench_config5")).read().strip().splitlines()[-1])
    assert c5["config"]["nx"] == 512 and c5["config"]["all_solves_ok"] is True and "spike" in c5["roofline"]["kernel"]
    assert c5["value"] >= 7.5e4 and c5["single_stream"]["solves_per_sec"] < c5["value"]
    r5 = c5["roofline"]
    assert abs(r5["kernel_ms"] - c5["single_stream"]["kernel_ms"]) < 1e-9  # priced with the kernel's own duration ...
    assert r5["in_flight"]["kernel_ms"] > r5["kernel_ms"]                   # ... the four-in-flight bracket beside it
    assert r5["traffic"] is not None and "r05" in r5["traffic_source"]


def test_leg_watchdog_emits_the_line_and_ends_the_job():
    """bench.py at N > 1: a leg that outlives MTRPL_BENCH_LEG_TIMEOUT costs the contract line that leg's object, not the
    line -- rank 0 emits what it has with {"pt": {"error": ...}} and exits 0; the other ranks exit 0 without a line."""
    import json
    import subprocess
    import sys
    import time
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "rank = int(sys.argv[1])\n"
            "bench._leg_watchdog(rank, {'metric': 'trpl_forward_solves_per_sec', 'value': 1.0} if rank == 0 else None, 'pt')\n"
            "time.sleep(30)\n"
            "print('the leg came back')\n" % REPO)
    env = dict(os.environ, MTRPL_BENCH_LEG_TIMEOUT="0.3")
    for rank in (0, 1):
        t0 = time.time()
        r = subprocess.run([sys.executable, "-c", code, str(rank)], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-500:]
        assert time.time() - t0 < 25 and "the leg came back" not in r.stdout
        if rank == 0:
            line = json.loads(r.stdout.strip().splitlines()[-1])
            assert line["value"] == 1.0 and "did not finish" in line["pt"]["error"]
        else:
            assert r.stdout.strip() == ""


def test_round6_bench_line_prices_its_roofline_with_the_corrected_issue_costs():
    """The committed round-6 line: everything the round-5 line carried, the VALU-issue bound priced with the issue costs of the
    corrected micro-benchmark (tools/mb/rcp_rate.hip: 128 operations per trip of its loop; rounds 3-5 counted the loop's branch
    into eight -- profiles/r06_z_issue_cost.txt), and batch.solve_batch timed both ways (an ordinary array by default, the
    context's pinned memory on request: same values)."""
    import glob
    import json
    from metrotrpl_amd import pt_bench
    path = sorted(glob.glob(os.path.join(REPO, "profiles", "r06_?_bench_config2.json")))[-1]
    d = json.loads(open(path).read().strip().splitlines()[-1])
    assert d["metric"] == "trpl_forward_solves_per_sec" and d["n_gpus"] == 1 and d["value"] >= 6.5e5
    assert d["config"]["all_solves_ok"] is True and d["config"]["solves_per_simd"] == 2
    t = d["roofline"]["valu_issue"]["throughput"]
    assert "r06_z_issue_cost" in t["source"]
    assert 8.5e5 < t["bound_solves_per_sec"] < 9.5e5 and 0.65 < t["frac_of_bound"] < 0.85
    assert abs(t["frac_of_bound"] - d["value"] / t["bound_solves_per_sec"]) < 1e-9
    assert pt_bench.ISSUE_COST_2WAVES["SQ_INSTS_VALU_FMA_F64"] == 4.9  # (5.8 until round 5)
    # ... and the measurement it comes from: an fp64 FMA with two waves per SIMD, ns per wave-op per SIMD -> cycles at 2.4 GHz
    txt = open(os.path.join(REPO, "profiles", "r06_z_issue_cost.txt")).read().split("== 2 wave(s) per SIMD")[1].split("==")[0]
    ns = float([ln for ln in txt.splitlines() if ln.startswith("v_fma_f64")][0].split(",")[1].split()[0])
    assert abs(ns * 2.4 - 4.9) < 0.15
    sb = d["api"]["solve_batch"]
    assert sb["python_batch_solve_batch"]["all_solves_ok"] is True and sb["python_batch_solve_batch_pinned"]["same_values"] is True
    assert sb["pinned_out"]["solves_per_sec"] >= 4.5e5 and sb["python_batch_solve_batch"]["solves_per_sec"] >= 4.0e5
    assert max(d["accuracy"]["max_rel_err_per_decade_0_to_8"]) < 1e-6
    assert d["pt"]["64"]["ms_per_iteration"] <= 2.2 and d["pt"]["512"]["ms_per_iteration"] <= 6.3
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
