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
