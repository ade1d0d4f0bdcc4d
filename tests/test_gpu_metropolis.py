"""The batched Metropolis / parallel-tempering driver on the GPU against walks produced
by the reference's serial loop (tests/golden/mcmc.npz): every accept / swap decision, the
visited states and the log-likelihoods, with the LikelihoodEngine (HIP solve + fused IRF /
likelihood kernels) as the evaluator."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def gpu_walk(name, **kw):
    import torch
    from mcmc_common import load_case
    from metrotrpl_amd import laplace, metropolis as M
    from metrotrpl_amd.likelihood import engine_from_fields
    E, num_iters, ref = load_case(name, laplace.make_I_tables)
    ef = E.ensemble_fields
    eng = engine_from_fields(ef, E.unique_fields[0])
    T = np.ascontiguousarray(ef["_T"], dtype=np.float64)

    def gather(states):
        return eng.eval(states, T).sum(axis=1)
    walk = M.Walk(ef, E.unique_fields, np.random.default_rng(M.SEED), gather, rng_mode="serial", **kw)
    walk.run(E.H.states, E.H.loglikelihood, E.H.accept, 0, num_iters, need_initial_state=True)
    return E, walk, ref


def test_single_chain_walk_on_gpu():
    from mcmc_common import check_walk
    E, walk, ref = gpu_walk("single")
    check_walk(E, walk, ref, rtol_ll=2e-5)


def test_parallel_tempering_walk_on_gpu():
    from mcmc_common import check_walk
    E, walk, ref = gpu_walk("pt", serial_swap_aliasing=True)
    check_walk(E, walk, ref, rtol_ll=2e-5)


def test_metro_entry_point_on_gpu(tmp_path):
    """metro() end to end on one GPU (engine built from the field dictionaries, IRF CSV read
    from irf_dir, checkpoints written), against the same golden walk."""
    import copy
    import json
    from metrotrpl_amd import metropolis as M, sim_utils as S
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mcmc.npz"))
    cfg = json.loads(str(z["pt__config"]))
    pi, si, mf = copy.deepcopy(cfg["param_info"]), cfg["sim_info"], copy.deepcopy(cfg["MCMC_fields"])
    pi["prior_dist"] = {k: tuple(v) for k, v in pi["prior_dist"].items()}
    mf["solver"] = tuple(mf["solver"])
    mf.update(output_path=str(tmp_path / "out"), checkpoint_freq=10)
    irf_dir = tmp_path / "IRFs"
    irf_dir.mkdir()
    np.savetxt(irf_dir / "irf_745nm.csv", z["pt__irf"], delimiter=",")
    times = z["pt__times"]
    e_data = ([times, times], [v for v in z["pt__vals"]], [u for u in z["pt__uncs"]])
    E = M.metro(si, z["pt__iniPar"].copy(), e_data, mf, pi, export_path="final.pik", serial_fallback=True,
                serial_swap_aliasing=True, irf_dir=str(irf_dir))
    np.testing.assert_array_equal(E.H.accept, z["pt__accept"])
    np.testing.assert_allclose(E.H.states, z["pt__states"], rtol=1e-12)
    np.testing.assert_allclose(E.H.loglikelihood, z["pt__logll"], rtol=2e-5)
    np.testing.assert_array_equal(E.H.swap_accept, z["pt__swap_accept"])
    back = S.load_pickle(str(tmp_path / "out" / "final.pik"))
    assert back.latest_iter == mf.get("num_iters", 24) or back.latest_iter == 24
    np.testing.assert_array_equal(back.H.states, E.H.states)
    assert back.H.mean_tauN.shape == (3, 24)


def test_main_from_config_script_and_csv_files(tmp_path, monkeypatch):
    """The whole reference workflow from files: config script (written by the reference's
    generator: tests/golden/configs/default_terse.txt, paths re-pointed), initial-condition
    CSV, stacked measurement CSV -> metro -> pickle."""
    from metrotrpl_amd import forward_solver, main, sim_utils as S, workloads as W
    golden = os.path.join(os.path.dirname(__file__), "golden")
    text = open(os.path.join(golden, "configs", "default_terse.txt")).read()
    text = text.replace("Inputs/fluences.csv", str(tmp_path / "flu.csv")).replace(
        "Inputs/meas.csv", str(tmp_path / "meas.csv")).replace("bay_outputs/run", str(tmp_path / "out")).replace(
        "Num iters: 100", "Num iters: 6").replace("Time cutoffs: 0\t2000", "Time cutoffs: 0\t200")
    (tmp_path / "mcmc0.txt").write_text(text)
    ini = np.array(W.STAUB_FLUENCES, dtype=float)
    np.savetxt(tmp_path / "flu.csv", ini, delimiter=",")
    times = np.linspace(0, 400, 201)
    ix, un, s0 = W.indexes(), W.units(), W.initial_state()
    rows = []
    for m in range(6):
        g = S.Grid(W.STAUB_LENGTHS[m], 128, times, 4.0)
        sig = forward_solver.solve(np.array(ini[m]), g, s0.copy(), ix, units=un, ini_mode="fluence")
        rows.append(np.stack([times, sig, 0.05 * sig], axis=1))
    np.savetxt(tmp_path / "meas.csv", np.concatenate(rows), delimiter=",")
    monkeypatch.chdir(tmp_path)
    ens = main.run("mcmc", 0)
    assert ens.H.states.shape == (1, 13, 6)
    assert len(ens.ensemble_fields["_times"]) == 6 and len(ens.ensemble_fields["_times"][0]) == 101  # cut to 200 ns
    # the data were simulated at the initial guess: its log-likelihood is ~0 and never gets worse than a move allows
    assert -1e-6 < ens.H.loglikelihood[0, 0] <= 0
    assert np.all(np.isfinite(ens.H.loglikelihood))
    back = S.load_pickle(str(tmp_path / "out" / "CPU0-final.pik"))
    np.testing.assert_array_equal(back.H.states, ens.H.states)
    assert back.ensemble_fields["ini_mode"] == "fluence" and back.latest_iter == 6


def test_main_with_selection_scale_factors_and_tempering(tmp_path, monkeypatch):
    """The file-driven flow with the options that re-index things: select_obs_sets keeps measurements
    0, 2, 4 (sim_info["nx"] stays at full length, as the reference leaves it), two fittable scale
    factors tied by a constraint group, a 3-rung ladder with swaps every 2 iterations."""
    from metrotrpl_amd import forward_solver, main, sim_utils as S, workloads as W
    golden = os.path.join(os.path.dirname(__file__), "golden")
    text = open(os.path.join(golden, "configs", "default_terse.txt")).read()
    text = text.replace("Inputs/fluences.csv", str(tmp_path / "flu.csv")).replace(
        "Inputs/meas.csv", str(tmp_path / "meas.csv")).replace("bay_outputs/run", str(tmp_path / "out")).replace(
        "Num iters: 100", "Num iters: 8").replace("Time cutoffs: 0\t2000", "Time cutoffs: 0\t100").replace(
        "Select measurement: None", "Select measurement: 0\t2\t4")
    text += ("Scale factor: 0.02\t[0, 2, 4]\t(0, 2)\t[1.0, 1.0, 1.0, 1.0, 1.5, 1.0]\n"
             "Parallel tempering: 1\t3\t9\nTempering frequency: 2\n")
    assert "Select measurement: 0\t2\t4" in text
    (tmp_path / "mcmc0.txt").write_text(text)
    ini = np.array(W.STAUB_FLUENCES, dtype=float)
    np.savetxt(tmp_path / "flu.csv", ini, delimiter=",")
    times = np.linspace(0, 200, 101)
    ix, un, s0 = W.indexes(), W.units(), W.initial_state()
    rows = []
    for m in range(6):
        g = S.Grid(W.STAUB_LENGTHS[m], 128, times, 4.0)
        sig = forward_solver.solve(np.array(ini[m]), g, s0.copy(), ix, units=un, ini_mode="fluence")
        rows.append(np.stack([times, sig * (1.5 if m == 4 else 1.0), 0.05 * sig], axis=1))
    np.savetxt(tmp_path / "meas.csv", np.concatenate(rows), delimiter=",")
    monkeypatch.chdir(tmp_path)
    ens = main.run("mcmc", 0, serial_fallback=True)
    ef = ens.ensemble_fields
    assert ef["_sim_info"]["num_meas"] == 3 and list(ef["_sim_info"]["lengths"]) == [311, 311, 311]
    assert ef["names"][-2:] == ["_s0", "_s2"]            # measurement 1 of the selection rides on _s0 (constraint group)
    assert ens.H.states.shape == (3, 15, 8) and len(ef["_times"]) == 3 and len(ef["_times"][0]) == 51
    # data were simulated at the initial guess with measurement 4 scaled by 1.5 = the initial guess of _s2
    assert -1e-5 < ens.H.loglikelihood[0, 0] <= 0
    assert np.all(np.isfinite(ens.H.loglikelihood)) and ens.H.swap_attempts.sum() == 3 * 2
    assert ens.H.accept.sum() > 0
