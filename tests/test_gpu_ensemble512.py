"""BASELINE configs[3] on ONE GPU: the 512-chain parallel-tempering ensemble and its 512-rung ladder
(reference: trial_move_evaluation.py:147-158 for ll(T), metropolis.py:129-135 and :211-261 for the exchange).

What 8 GPUs would run as 8 blocks of 64 chains runs here as one block of 512, through the same code: the engine's
launches of 3072 solves, `loglik_kernel`'s many-temperature branch -- the one that no longer fits its partial sums
into the IRF work space (8 * n_T > 6 * nk, i.e. n_T > 141 with the 189-tap IRF) and the one without any IRF work
space (irf < 0) --, the all-gather of the [512 x 512] table through a one-rank `nccl` group (the rehearsal
MTRPL_FORCE_DIST=1 gives bench.py), the replay of the 511 swap attempts.  VERDICT r3, missing 2 / weak 1."""
import os

import numpy as np
import pytest

from tests.test_gpu_ensemble import _child, _oracle_row

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _shared(times, vals, uncs, irf):
    from metrotrpl_amd import workloads as W
    from oracle import oracle as O
    sh = {"_sim_info": {"lengths": list(W.STAUB_LENGTHS), "nx": [128] * 6, "meas_types": ["TRPL"] * 6, "num_meas": 6},
          "_param_indexes": W.indexes(), "units": W.units(), "_init_params": np.array(W.STAUB_FLUENCES, dtype=float),
          "_times": [times] * 6, "_vals": list(vals), "_uncs": list(uncs), "hmax": 4.0,
          "solver": ("solveivp",), "model": "std", "ini_mode": "fluence", "rtol": 1e-10, "atol": 1e-14,
          "force_min_y": True}
    if irf:
        irf_raw = np.load(os.path.join(HERE, "golden", "irf.npz"))["irf745_raw"]
        sh["irf_convolution"] = [745] * 6
        sh["_IRF_tables"] = O.make_I_tables({745: irf_raw})
    return sh


def test_config4_512_chains_512_temperatures_on_one_gpu(tmp_path):
    from multiprocessing import get_context
    from oracle import oracle as O
    # one iteration per launch, one engine, through RCCL: ten iterations, the replica exchange at the tenth
    a = _child(tmp_path, "c512", chains=512, dist=1, parts=2, sort=1, rng_mode="counter", prefetch=0, iters=10,
               temper_freq=10, port=29631)
    assert str(a["backend"]) == "nccl" and a["table0"].shape == (512, 512) and list(a["calls"]) == [512] * 10
    assert int(a["n_collectives"]) == 1 + 1  # initialise + the one exchange
    assert a["swap_attempts"].sum() == 511 and int(a["n_acc"]) > 0
    assert np.all(np.isfinite(a["table0"])) and np.all(np.isfinite(a["logll"]))
    # ll(T) rises with T along every row of the table (the denominator sigma^2 T + 2 u^2 grows)
    assert np.all(np.diff(a["table0"], axis=1) > 0)
    # (a) the table rows of two cold and two hot chains against the oracle at ALL 512 temperatures
    shared = _shared(a["times"], a["vals"], a["uncs"], irf=True)
    uniq = {"model_uncertainty": {"TRPL": 1.0}, "_T": 1.0}
    chains = [0, 1, 510, 511]
    O.build()
    with get_context("spawn").Pool(4) as pool:
        rows = dict(pool.map(_oracle_row, [(k, a["states0"][k], shared, uniq, a["T"]) for k in chains]))
    for k in chains:
        ref = rows[k]
        assert np.all(np.isfinite(ref))
        np.testing.assert_allclose(a["table0"][k], ref, rtol=1e-6, atol=1e-9 * np.max(np.abs(ref)), err_msg=f"chain {k}")
    # (b) the same walk with two iterations per launch and two chain groups on two streams: bit for bit
    b = _child(tmp_path, "c512_pre", chains=512, dist=1, parts=1, sort=1, rng_mode="counter", prefetch=1, iters=10,
               temper_freq=10, groups=2, port=29632)
    assert list(b["calls"]) == [768] * 10  # 256 chains per group x 3 states, five double iterations per group
    for key in ("table0", "states", "logll", "table", "swap_attempts", "swap_accepts", "n_acc"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    assert not np.array_equal(a["states"], a["states0"])
    # (c) ... and with the decisions of a launch taken on the device (pt._DecidePipe, one iteration per launch: the host
    # has both outcomes' proposals packed and on the device before the launch ends; 3072-solve launches in the engine's
    # measured-cost order, split over two streams)
    c = _child(tmp_path, "c512_dev", chains=512, dist=1, parts=2, sort=1, rng_mode="counter", prefetch=0, iters=10,
               temper_freq=10, device_decisions=1, packed_order="solves", port=29633)
    assert int(c["n_pipelined"]) == 10 and list(c["calls"]) == [] and int(c["n_parts"]) == 2 and bool(c["ordered"])
    for key in ("table0", "states", "logll", "table", "swap_attempts", "swap_accepts", "n_acc"):
        np.testing.assert_array_equal(a[key], c[key], err_msg=key)


def test_ladder_of_512_temperatures_without_irf():
    """No IRF: the trace is trimmed (trial_move_evaluation.py:98-101), there is no IRF work space, and with n_T >= 32
    the ladder sums run one thread per temperature over all samples (`nT >= 32 && !fits` with irf < 0)."""
    from multiprocessing import get_context
    from metrotrpl_amd import forward_solver, workloads as W
    from metrotrpl_amd.likelihood import LikelihoodEngine
    from metrotrpl_amd.sim_utils import Grid
    from oracle import oracle as O
    times = W.default_times()
    ix, un, s0 = W.indexes(), W.units(), W.initial_state()
    vals, uncs = [], []
    for m in range(6):
        g = Grid(W.STAUB_LENGTHS[m], 128, times, 4.0)
        sig = forward_solver.solve(np.array(W.STAUB_FLUENCES[m]), g, s0.copy(), ix, meas="TRPL", units=un, ini_mode="fluence")
        vals.append(np.log10(sig))
        uncs.append(np.full(times.shape[0], 0.02))
    eng = LikelihoodEngine(ix, un, W.STAUB_LENGTHS, 128, ["TRPL"] * 6, [times] * 6, vals, uncs, {"TRPL": 1.0},
                           np.array(W.STAUB_FLUENCES), ini_mode="fluence", force_min_y=True)
    T = W.ladder(512)
    ens = W.pt_ensemble(512)
    chains = [0, 1, 510, 511]
    table = eng.eval(ens["chain_states"][chains], T)  # [4, 6, 512]
    assert table.shape == (4, 6, 512) and np.all(np.isfinite(table))
    shared = _shared(times, vals, uncs, irf=False)
    uniq = {"model_uncertainty": {"TRPL": 1.0}, "_T": 1.0}
    O.build()
    with get_context("spawn").Pool(4) as pool:
        rows = dict(pool.map(_oracle_row, [(k, ens["chain_states"][c], shared, uniq, T) for k, c in enumerate(chains)]))
    for k in range(4):
        np.testing.assert_allclose(table[k].sum(axis=0), rows[k], rtol=1e-6, atol=1e-9 * np.max(np.abs(rows[k])),
                                   err_msg=f"chain {chains[k]}")
    # the same four chains at 64 temperatures take the LDS-partials branch only with an IRF; here both sizes run the
    # one-thread-per-temperature loop and must agree where the ladders coincide (T_0 = 1, T_last = 1000)
    t64 = eng.eval(ens["chain_states"][chains], W.ladder(64))
    np.testing.assert_allclose(t64[:, :, [0, -1]], table[:, :, [0, -1]], rtol=1e-12)
