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
