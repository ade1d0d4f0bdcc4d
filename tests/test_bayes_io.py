"""On-disk formats: this package's readers against what the reference's own bayes_io
returned for the same files (tests/golden/configs/*, tests/golden/io_data.json; generator:
tests/golden/make_io_golden.py) and the scenarios of the reference's Tests/test_bayes_io.py /
test_io_remapping.py."""
import glob
import json
import os

import numpy as np
import pytest

from metrotrpl_amd import bayes_io as B

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def jsonable(o):
    if isinstance(o, dict):
        return {k: jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    if isinstance(o, np.ndarray):
        return o.tolist()
    if isinstance(o, (np.floating, np.integer)):
        return o.item()
    return o


CONFIGS = sorted(glob.glob(os.path.join(GOLDEN, "configs", "*.txt")))


@pytest.mark.parametrize("path", CONFIGS, ids=[os.path.basename(p)[:-4] for p in CONFIGS])
def test_read_config_script_file_matches_reference(path):
    want = json.load(open(path[:-4] + ".json"))
    sim, pi, meas, mc = B.read_config_script_file(path)
    got = jsonable({"sim_info": sim, "param_info": pi, "meas_fields": meas, "MCMC_fields": mc})
    assert got == want
    # types the driver relies on
    assert isinstance(mc["solver"], tuple) and isinstance(meas["select_obs_sets"], np.ndarray)


def test_config_count():
    assert len(CONFIGS) == 6


def test_not_a_config(tmp_path):
    p = tmp_path / "x.txt"
    p.write_text("hello\n")
    with pytest.raises(OSError):
        B.read_config_script_file(str(p))


def test_get_data_matches_reference():
    ref = json.load(open(os.path.join(GOLDEN, "io_data.json")))
    fn = os.path.join(GOLDEN, "ref_testfiles", "test_data.csv")
    for case in ref["get_data"]:
        with np.errstate(divide="ignore", invalid="ignore"):
            t, y, u = B.get_data(fn, dict(case["ic_flags"]), {"log_y": case["log_y"]})
        assert len(t) == len(case["t"])
        for a, b in zip(t + y + u, case["t"] + case["y"] + case["u"]):
            np.testing.assert_array_equal(np.asarray(a, dtype=float), np.asarray(b, dtype=float))


def test_get_initpoints_matches_reference():
    ref = json.load(open(os.path.join(GOLDEN, "io_data.json")))
    fn = os.path.join(GOLDEN, "ref_testfiles", "test_initpoints.csv")
    for case in ref["get_initpoints"]:
        got = B.get_initpoints(fn, {"select_obs_sets": case["select"]})
        np.testing.assert_array_equal(got, np.asarray(case["out"], dtype=float))


def test_remapping_examples_from_the_reference_docstrings():
    # bayes_io.py:197-214, :221-239
    np.testing.assert_array_equal(B.remap_fittable_inds([0, 1, 3, 4], np.array([0, 2, 4])), [0, 2])
    assert B.remap_constraint_grps([(0, 1, 2), (3, 4, 5)], np.array([0, 2, 4])) == [(0, 1)]


def test_insert_param_constraint_groups():
    pi = {"names": ["a"], "do_log": {}, "prior_dist": {}, "init_guess": {}, "trial_move": {}, "active": {}}
    mc = {"fittable_fluences": [0.02, [0, 1, 2, 4], [(0, 2), (1, 4)], [1.0, 1.5, 1.0, 1.0, 1.5]]}
    B.insert_param(pi, mc, mode="fluences")
    assert pi["names"] == ["a", "_f0", "_f1"]
    assert pi["init_guess"]["_f1"] == 1.5 and pi["trial_move"]["_f0"] == 0.02
    assert pi["prior_dist"]["_f0"] == (0, np.inf) and pi["active"]["_f1"] == 1
    with pytest.raises(NotImplementedError):
        B.insert_param(pi, mc, mode="bogus")


def test_extract_tuples():
    assert B.extract_tuples("(0, inf)\t(-inf, 3.5)", "\t") == [(0.0, np.inf), (-np.inf, 3.5)]
    assert B.extract_tuples("(TRPL, 1)|(TRTS, 10)", "|") == [("TRPL", 1.0), ("TRTS", 10.0)]
    assert B.extract_tuples("(0, 2, 4)|(1, 3)", "|", dtype=int) == [(0, 2, 4), (1, 3)]
