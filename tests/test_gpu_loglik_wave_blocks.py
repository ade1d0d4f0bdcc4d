"""The one-wave-per-block form of the fused likelihood kernel (context option `loglik_wave_blocks`: blocks that fit
wherever a single solve wave has left, for launches that share the device with another stream's queued solves) against
the twelve-wave form: with 32 and more ladder temperatures no sum depends on the block size, so the tables must be the
same numbers bit for bit -- with the IRF (both ladder branches: partial sums in the IRF work space at 64 temperatures,
one thread per temperature at 512) and without (trial_move_evaluation.py:96-158)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


@pytest.mark.parametrize("n_T,irf", [(64, True), (512, True), (64, False), (32, True)])
def test_wave_blocks_give_the_same_table(n_T, irf):
    import torch
    from metrotrpl_amd import _lib as L, laplace, pt_bench, workloads as W
    from metrotrpl_amd.likelihood import LikelihoodEngine
    ctx = L.Context(0)
    times = W.default_times()
    irf_tables = laplace.make_I_tables({745: pt_bench.synthetic_irf()})
    vals, uncs = pt_bench.synthetic_data(irf_tables, 128, times)
    eng = LikelihoodEngine(W.indexes(), W.units(), W.STAUB_LENGTHS, 128, ["TRPL"] * 6, [times] * 6, vals, uncs, {"TRPL": 1.0},
                           np.array(W.STAUB_FLUENCES), irf_tables=irf_tables if irf else None,
                           irf_convolution=[745] * 6 if irf else None, ctx=ctx, force_min_y=True)
    states = W.pt_ensemble(64)["chain_states"][::5]  # 13 states across the prior box
    T = W.ladder(n_T)
    ref = eng.eval(states, T)
    assert ref.shape == (13, 6, n_T) and np.all(np.isfinite(ref))
    for waves in (1, 2, 4):
        ctx.set_option("loglik_wave_blocks", waves)
        assert ctx.get_option("loglik_wave_blocks") == waves
        got = eng.eval(states, T)
        np.testing.assert_array_equal(got, ref, err_msg=f"{waves} wave(s) per block")
    ctx.set_option("loglik_wave_blocks", 0)
    with pytest.raises(ValueError):
        ctx.set_option("loglik_wave_blocks", 3)
    assert np.all(np.diff(ref, axis=2) > 0)  # ll(T) rises with T
    torch.cuda.synchronize()
