"""The sharded parallel-tempering path at a BASELINE size, on the GPU and through RCCL
(reference: trial_move_evaluation.py:9-28 per chain, metropolis.py:118-135 and :211-261 for the exchange).

Config 3 -- 64 chains x 6 measurements, nx=128, 1001 sample times, the shipped 745 nm IRF -- runs in fresh
child processes (tests/gpu_ensemble_child.py): inside a one-rank `nccl` group that is initialised before any
GPU call, with the two-stream split and the measured-cost ordering of the likelihood engine switched on, and
without any of it.  (a) the log-likelihood table of 8 chains equals the CPU oracle's eval_trial_move at all
64 ladder temperatures; (b) all runs produce the identical walk, bit for bit.

The ensemble runs with the reference's dynamic-range clamp (`force_min_y`, utils.py:16-31): the chain
states are spread over the whole prior box, so many simulated traces fall 10 and more decades within the
2000 ns window, where the signal is a difference of nearly equal numbers and neither side is converged; the
clamp holds the comparison to the dynamic range of the data, as an experimental noise floor does."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _child(tmp_path, tag, **kw):
    out = str(tmp_path / f"{tag}.npz")
    cmd = [sys.executable, os.path.join(HERE, "gpu_ensemble_child.py"), "--out", out]
    for k, v in kw.items():
        cmd += [f"--{k.replace('_', '-')}", str(v)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return np.load(out)


def _oracle_row(args):
    k, state, shared, uniq, T = args
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    from oracle import oracle as O
    _, table = O.eval_trial_move(state, uniq, shared, temperatures=T)
    return k, table.sum(axis=0)


def test_config3_ensemble_through_rccl_and_two_streams(tmp_path):
    from multiprocessing import get_context
    from metrotrpl_amd import workloads as W
    from oracle import oracle as O
    a = _child(tmp_path, "dist_p2", dist=1, parts=2, sort=1, port=29617)
    b = _child(tmp_path, "plain", dist=0, parts=1, sort=0)
    c = _child(tmp_path, "dist_p1", dist=1, parts=1, sort=1, port=29618)
    assert str(a["backend"]) == "nccl" and str(c["backend"]) == "nccl" and str(b["backend"]) == "none"
    assert int(a["n_parts"]) == 2 and int(b["n_parts"]) == 1 and bool(a["ordered"]) and not bool(b["ordered"])
    assert int(a["swaps"]) == 2 and a["swap_attempts"].sum() == 2 * 63
    assert int(a["n_collectives"]) == 1 + 12  # stream-mode draws: the table is gathered at every iteration
    assert a["states"].shape == (64, len(W.PARAM_NAMES))
    # (b) the same walk whatever the plumbing: states, log-likelihoods, the whole table, swap decisions
    for other in (b, c):
        for key in ("table0", "states", "logll", "table", "swap_attempts", "swap_accepts"):
            np.testing.assert_array_equal(a[key], other[key], err_msg=key)
    assert not np.array_equal(a["states"], a["states0"])  # something moved
    # (a) against the oracle: 8 chains x 6 measurements x 64 temperatures
    irf_raw = np.load(os.path.join(HERE, "golden", "irf.npz"))["irf745_raw"]
    times = a["times"]
    shared = {"_sim_info": {"lengths": list(W.STAUB_LENGTHS), "nx": [128] * 6, "meas_types": ["TRPL"] * 6, "num_meas": 6},
              "_param_indexes": W.indexes(), "units": W.units(), "_init_params": np.array(W.STAUB_FLUENCES, dtype=float),
              "_times": [times] * 6, "_vals": list(a["vals"]), "_uncs": list(a["uncs"]), "hmax": 4.0,
              "solver": ("solveivp",), "model": "std", "ini_mode": "fluence", "rtol": 1e-10, "atol": 1e-14,
              "irf_convolution": [745] * 6, "_IRF_tables": O.make_I_tables({745: irf_raw}), "force_min_y": True}
    uniq = {"model_uncertainty": {"TRPL": 1.0}, "_T": 1.0}
    T = a["T"]
    chains = [0, 1, 2, 3, 31, 32, 62, 63]
    O.build()
    with get_context("spawn").Pool(8) as pool:
        rows = dict(pool.map(_oracle_row, [(k, a["states0"][k], shared, uniq, T) for k in chains]))
    for k in chains:
        ref = rows[k]
        assert np.all(np.isfinite(ref))
        np.testing.assert_allclose(a["table0"][k], ref, rtol=1e-6, atol=1e-9 * np.max(np.abs(ref)), err_msg=f"chain {k}")


def test_prefetched_ensemble_walks_the_same_chain(tmp_path):
    """Two iterations per launch (192 states x 6 solves, both outcomes of the first evaluated ahead,
    pt.PTEnsemble.iterate2) through RCCL == one iteration per launch, bit for bit, on the real evaluator."""
    # (384-solve launches run one solve per SIMD, 1152-solve launches two per SIMD: the same arithmetic)
    one = _child(tmp_path, "one", dist=0, parts=1, sort=0, rng_mode="counter", prefetch=0, iters=12, temper_freq=4)
    two = _child(tmp_path, "two", dist=1, parts=1, sort=1, rng_mode="counter", prefetch=1, iters=12, temper_freq=4,
                 port=29619)
    assert list(one["calls"]) == [64] * 12 and list(two["calls"]) == [192] * 6
    assert str(two["backend"]) == "nccl" and bool(two["ordered"])
    # counter-mode draws: the ranks exchange at replica-exchange steps only -- one all-gather at initialise and one
    # per exchange (iterations 4, 8, 12), through RCCL; the run without a process group issues none
    assert int(two["n_collectives"]) == 1 + 3 and int(one["n_collectives"]) == 0
    for key in ("states", "logll", "table", "swap_attempts", "swap_accepts", "n_acc"):
        np.testing.assert_array_equal(one[key], two[key], err_msg=key)
    assert int(one["n_acc"]) > 0 and one["swap_attempts"].sum() == 3 * 63
    # ... and cut into two chain groups, each with its own engine and HIP stream, launches overlapping
    grp = _child(tmp_path, "grp", dist=1, parts=1, sort=1, rng_mode="counter", prefetch=1, iters=12, temper_freq=4,
                 groups=2, port=29620)
    assert list(grp["calls"]) == [96] * 12 and str(grp["backend"]) == "nccl"
    for key in ("states", "logll", "table", "swap_attempts", "swap_accepts", "n_acc"):
        np.testing.assert_array_equal(one[key], grp[key], err_msg=key)
    # ... and with the decisions taken on the device, behind the launch on its stream (pt._DecidePipe: the host prepares
    # the four proposal sets a launch's two decisions can ask for while it runs; no launch waits for the host): segments
    # of three launches per group, two in flight
    dd = _child(tmp_path, "dd", dist=1, parts=1, sort=1, rng_mode="counter", prefetch=1, iters=12, temper_freq=6,
                groups=2, device_decisions=1, port=29621)
    ref6 = _child(tmp_path, "ref6", dist=0, parts=1, sort=0, rng_mode="counter", prefetch=0, iters=12, temper_freq=6)
    assert int(dd["n_pipelined"]) == 12 and list(dd["calls"]) == [] and str(dd["backend"]) == "nccl"
    for key in ("states", "logll", "table", "swap_attempts", "swap_accepts", "n_acc"):
        np.testing.assert_array_equal(ref6[key], dd[key], err_msg=key)
    assert int(ref6["n_acc"]) > 0
    # (device_decisions=1: the library's mtrpl_pt_select_dev / mtrpl_pt_decide_dev; 2: the same two steps as torch ops)
    dd2 = _child(tmp_path, "dd2", dist=0, parts=1, sort=0, rng_mode="counter", prefetch=1, iters=12, temper_freq=6,
                 groups=1, device_decisions=2)
    assert int(dd2["n_pipelined"]) == 6 and list(dd2["calls"]) == []
    for key in ("states", "logll", "table", "swap_attempts", "swap_accepts", "n_acc"):
        np.testing.assert_array_equal(ref6[key], dd2[key], err_msg=key)
    # ... and three groups whose chains are dealt again by cost at every replica exchange, the last group two solves per SIMD
    # (the latency-bound form of bench.py's 64-chain ensemble): the second stretch of six iterations runs on dealt groups
    dealt = _child(tmp_path, "dealt", dist=0, parts=1, sort=0, rng_mode="counter", prefetch=1, iters=12, temper_freq=6,
                   groups=3, device_decisions=1, dealt=1)
    assert int(dealt["n_pipelined"]) == 18 and list(dealt["calls"]) == []
    for key in ("states", "logll", "table", "swap_attempts", "swap_accepts", "n_acc"):
        np.testing.assert_array_equal(ref6[key], dealt[key], err_msg=key)
    own, cost = dealt["pipe_own"].astype(int), dealt["chain_cost"]
    assert sorted(own.tolist()) == list(range(64)) and np.all(cost > 0)
    assert not np.array_equal(own, np.concatenate([np.arange(g, 64, 3) for g in range(3)]))  # not the interleaved groups
    # the fp64-block kernel walks the same chain: the same decisions, log-likelihoods to 1e-7
    f64 = _child(tmp_path, "f64", dist=0, parts=1, sort=1, rng_mode="counter", prefetch=1, iters=12, temper_freq=4,
                 kernel_variant=2)
    np.testing.assert_array_equal(one["states"], f64["states"])
    np.testing.assert_array_equal(one["swap_accepts"], f64["swap_accepts"])
    np.testing.assert_allclose(one["logll"], f64["logll"], rtol=1e-7)
