"""
make_io_golden.py -- on-disk format fixtures produced by the reference's own
bayes_io (imported from /root/reference in the build container):

  tests/golden/configs/<name>.txt    config scripts WRITTEN by the reference's
                                     generate_config_script_file
  tests/golden/configs/<name>.json   what the reference's read_config_script_file
                                     returns for that file (sim_info, param_info,
                                     meas_fields, MCMC_fields)
  tests/golden/io_data.json          get_data / get_initpoints results of the reference
                                     for tests/golden/ref_testfiles/*.csv (data files the
                                     reference's own tests hold, Tests/testfiles/)

Only data is committed.    python tests/golden/make_io_golden.py
"""
import copy
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _import_reference  # noqa: E402
from make_mcmc_golden import jsonable  # noqa: E402


def base():
    names = ["n0", "p0", "mu_n", "mu_p", "ks", "Cn", "Cp", "Sf", "Sb", "tauN", "tauP", "eps", "Tm"]
    sim = {"lengths": [311, 2000, 311, 2000, 311, 2000], "nx": [128] * 6, "meas_types": ["TRPL"] * 6, "num_meas": 6}
    pi = {"names": list(names),
          "active": {n: int(n not in ("n0", "eps", "Tm")) for n in names},
          "unit_conversions": {"n0": 1e-21, "p0": 1e-21, "mu_n": 1e5, "mu_p": 1e5, "ks": 1e12, "Cn": 1e33,
                               "Cp": 1e33, "Sf": 1e-2, "Sb": 1e-2, "Tm": 1},
          "do_log": {n: 1 for n in names},
          "prior_dist": {"n0": (0, np.inf), "p0": (1e14, 1e16), "mu_n": (1e0, 1e2), "mu_p": (1e0, 1e2),
                         "ks": (1e-11, 1e-9), "Cn": (1e-29, 1e-27), "Cp": (1e-29, 1e-27), "Sf": (1e-1, 1e4),
                         "Sb": (1e-1, 1e4), "tauN": (1, 2e3), "tauP": (1, 1e4), "eps": (0, np.inf),
                         "Tm": (0, np.inf)},
          "init_guess": {"n0": 1e8, "p0": 3e15, "mu_n": 20, "mu_p": 20, "ks": 4.8e-11, "Cn": 4.4e-29,
                         "Cp": 4.4e-29, "Sf": 10, "Sb": 10, "tauN": 511, "tauP": 871, "eps": 10, "Tm": 300},
          "trial_move": {n: 0.003 for n in names}}
    meas = {"time_cutoff": [0, 2000], "select_obs_sets": None}
    mc = {"init_cond_path": os.path.join("Inputs", "fluences.csv"),
          "measurement_path": os.path.join("Inputs", "meas.csv"),
          "output_path": os.path.join("bay_outputs", "run"), "ini_mode": "fluence", "num_iters": 100,
          "solver": ("solveivp",), "model": "std", "model_uncertainty": 1, "log_y": 1, "scale_factor": None,
          "irf_convolution": None, "hard_bounds": 1, "force_min_y": 0, "checkpoint_freq": 1000,
          "load_checkpoint": None}
    return sim, pi, meas, mc


def variants():
    out = {}
    out["default"] = base()
    sim, pi, meas, mc = base()
    sim["meas_types"] = ["TRPL", "TRTS"] * 3
    meas["select_obs_sets"] = [0, 2, 3, 4]
    meas["time_cutoff"] = [0, np.inf]
    mc.update(parallel_tempering=[1, 2, 4, 8], temper_freq=5, irf_convolution=[745, 0, 745, 0, 700, 0],
              rtol=1e-8, atol=1e-12, hmax=2, model_uncertainty={"TRPL": 1, "TRTS": 10},
              scale_factor=[0.02, [0, 1, 2, 3, 4, 5], [(0, 2, 4), (1, 3, 5)], [1.0, 2.0, 1.0, 2.0, 1.0, 2.0]],
              fittable_fluences=[0.01, [0, 2, 4], None, [1.0, 1.0, 1.1, 1.0, 1.2, 1.0]],
              load_checkpoint="checkpoint_500.pik", force_min_y=1, hard_bounds=0, model="traps",
              solver=("odeint",), ini_mode="density")
    pi["do_mu_constraint"] = [20, 3]
    out["pt_fittables"] = (sim, pi, meas, mc)
    sim, pi, meas, mc = base()
    mc.pop("model_uncertainty")
    mc.update(likel2move_ratio={"TRPL": 500}, fittable_absps=[0.05, [1, 3], [(1, 3)], [1, 0.9, 1, 0.9, 1, 1]],
              scale_factor=[0.1, [0, 5], None])
    out["l2m_absps"] = (sim, pi, meas, mc)
    return out


if __name__ == "__main__":
    _import_reference()
    import bayes_io as B
    cfg_dir = os.path.join(HERE, "configs")
    os.makedirs(cfg_dir, exist_ok=True)
    for name, (sim, pi, meas, mc) in variants().items():
        for verbose in (True, False):
            tag = name + ("" if verbose else "_terse")
            path = os.path.join(cfg_dir, tag + ".txt")
            B.generate_config_script_file(path, copy.deepcopy(sim), copy.deepcopy(pi), copy.deepcopy(meas),
                                          copy.deepcopy(mc), verbose=verbose)
            got = B.read_config_script_file(path)
            with open(os.path.join(cfg_dir, tag + ".json"), "w") as f:
                json.dump(jsonable({"sim_info": got[0], "param_info": got[1], "meas_fields": got[2],
                                    "MCMC_fields": got[3]}), f, indent=1)
            print("wrote", tag)
    # data readers on the reference's own test files
    tf = os.path.join(HERE, "ref_testfiles")
    os.makedirs(tf, exist_ok=True)
    for fn in ("test_data.csv", "test_initpoints.csv"):
        shutil.copyfile(os.path.join("/root/reference/Tests/testfiles", fn), os.path.join(tf, fn))
        os.chmod(os.path.join(tf, fn), 0o644)
    cases = []
    for ic in ({"time_cutoff": None, "select_obs_sets": None, "noise_level": None},
               {"time_cutoff": [0, 3], "select_obs_sets": None, "noise_level": None},
               {"time_cutoff": [1, 10], "select_obs_sets": [0, 2], "noise_level": None},
               {"time_cutoff": None, "select_obs_sets": [1, 3], "noise_level": None}):
        for log_y in (0, 1):
            with np.errstate(divide="ignore", invalid="ignore"):
                t, y, u = B.get_data(os.path.join(tf, "test_data.csv"), dict(ic), {"log_y": log_y})
            cases.append({"ic_flags": jsonable(ic), "log_y": log_y, "t": jsonable(t), "y": jsonable(y),
                          "u": jsonable(u)})
    inits = []
    for sel in (None, [0, 2], [1]):
        inits.append({"select": sel, "out": jsonable(B.get_initpoints(os.path.join(tf, "test_initpoints.csv"),
                                                                      {"select_obs_sets": sel}))})
    with open(os.path.join(HERE, "io_data.json"), "w") as f:
        json.dump({"get_data": cases, "get_initpoints": inits}, f, indent=1)
    print("wrote io_data.json")
