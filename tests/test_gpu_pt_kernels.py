"""mtrpl_pt_decide_dev / mtrpl_pt_select_dev against the host rule they replace (metropolis.py:42-63 as
pt.metropolis_accept + the bookkeeping of pt.PTEnsemble._end): accept flags, the state's log-likelihood, the rows moved
into the current table, NaN and infinite log-likelihoods, one and three proposals per chain; the descriptor pick for
1, 2 and 4 candidate sets in chain-major and in a scrambled row order; argument errors."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _host_rule(tab, col, u, cur_ll, G):
    from metrotrpl_amd import pt

    class U:  # metropolis_accept draws its uniforms from a generator: hand it ours
        def __init__(self, v):
            self.v = v

        def random(self, n):
            return self.v
    c = col.shape[0]
    L = tab[:, np.arange(c), col]  # [G, c]
    a1 = pt.metropolis_accept(cur_ll, L[0], U(u[0]))
    cur1 = np.where(a1, L[0], cur_ll)
    if G == 1:
        return a1, np.zeros(c, bool), cur1, L
    ll2 = np.where(a1, L[2], L[1])
    a2 = pt.metropolis_accept(cur1, ll2, U(u[1]))
    return a1, a2, np.where(a2, ll2, cur1), L


@pytest.mark.parametrize("G,c,nT", [(3, 64, 64), (1, 171, 512), (3, 5, 7), (1, 1, 1)])
def test_decide_matches_the_host_rule(G, c, nT):
    import torch
    from metrotrpl_amd import _lib as L
    ctx = L.Context(0)
    rng = np.random.default_rng(G * 1000 + c)
    C_local = c + 3
    tab = -rng.random((G, c, nT)) * 30
    cur_tab = -rng.random((C_local, nT)) * 30
    col = rng.integers(0, nT, c).astype(np.int32)
    row = rng.permutation(C_local)[:c].astype(np.int32)
    cur_ll = cur_tab[row, col].copy()
    # the special values of trial_move_evaluation.py:70-72, :159-165 and a NaN
    for k, v in enumerate((np.nan, -np.inf, np.inf)):
        if c > 3 * k + 2:
            tab[0, 3 * k, col[3 * k]] = v
            cur_ll[3 * k + 1] = v if v != np.inf else -np.inf
            cur_tab[row[3 * k + 1], col[3 * k + 1]] = cur_ll[3 * k + 1]
            if G == 3:
                tab[2, 3 * k + 2, col[3 * k + 2]] = v
    u = rng.random((2, c))
    with np.errstate(all="ignore"):
        a1, a2, new_ll, Ls = _host_rule(tab, col, u, cur_ll, G)
    dev = torch.device("cuda", 0)
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in
         dict(tab=tab, col=col, row=row, u=u, cur_ll=cur_ll, cur_tab=cur_tab).items()}
    outcome = torch.full((c,), -1, dtype=torch.int32, device=dev)
    result = torch.zeros((4, c), dtype=torch.float64, device=dev)
    ctx.check(L.lib().mtrpl_pt_decide_dev(ctx.handle, c, nT, G, d["tab"].data_ptr(), d["col"].data_ptr(), d["row"].data_ptr(),
                                          d["u"].data_ptr(), d["cur_ll"].data_ptr(), d["cur_tab"].data_ptr(),
                                          outcome.data_ptr(), result.data_ptr(), 0))
    torch.cuda.synchronize()
    k = outcome.cpu().numpy()
    # (a uniform within rounding of the ratio could fall on the other side of the device's exp: none in these draws)
    np.testing.assert_array_equal(k & 1, a1.astype(int))
    np.testing.assert_array_equal(k >> 1, a2.astype(int))
    np.testing.assert_array_equal(d["cur_ll"].cpu().numpy(), new_ll)
    res = result.cpu().numpy()
    np.testing.assert_array_equal(res[0], k)
    np.testing.assert_array_equal(res[1], Ls[0])
    if G == 3:
        np.testing.assert_array_equal(res[2], Ls[1])
        np.testing.assert_array_equal(res[3], Ls[2])
    want = cur_tab.copy()
    src = np.where(a2[:, None], np.where(a1[:, None], tab[2 % G], tab[1 % G]), tab[0])
    moved = a1 | a2
    want[row[moved]] = src[moved]
    np.testing.assert_array_equal(d["cur_tab"].cpu().numpy(), want)
    if c > 2:
        assert np.isnan(tab[0, 0, col[0]]) and not a1[0]  # the NaN proposal was rejected


@pytest.mark.parametrize("n_cand,G,c,M,scrambled", [(1, 3, 21, 6, False), (4, 3, 22, 6, False), (2, 1, 171, 6, True),
                                                    (4, 3, 64, 1, True)])
def test_select_picks_every_row_from_its_chains_outcome(n_cand, G, c, M, scrambled):
    import torch
    from metrotrpl_amd import _lib as L
    ctx = L.Context(0)
    rng = np.random.default_rng(n_cand * 100 + c)
    B = L.SOLVE_DESC_DTYPE.itemsize
    n_rows = G * c * M
    srow = np.repeat(np.arange(G * c), M)
    if scrambled:
        srow = srow[rng.permutation(n_rows)]
    chain = (srow % c).astype(np.int32)
    cand = rng.integers(0, 256, (n_cand, n_rows, B), dtype=np.uint8)
    outcome = rng.integers(0, n_cand, c).astype(np.int32)
    dev = torch.device("cuda", 0)
    d_cand, d_chain, d_out = (torch.from_numpy(x).to(dev) for x in (cand, chain, outcome))
    nxt = torch.zeros(n_rows * B, dtype=torch.uint8, device=dev)
    ctx.check(L.lib().mtrpl_pt_select_dev(ctx.handle, n_rows, n_cand, B, d_cand.data_ptr(), d_chain.data_ptr(),
                                          d_out.data_ptr(), nxt.data_ptr(), 0))
    torch.cuda.synchronize()
    want = cand[outcome[chain] if n_cand > 1 else 0, np.arange(n_rows)]
    np.testing.assert_array_equal(nxt.cpu().numpy().reshape(n_rows, B), want)


def test_argument_errors():
    import torch
    from metrotrpl_amd import _lib as L
    ctx = L.Context(0)
    lib = L.lib()
    buf = torch.zeros(4096, dtype=torch.uint8, device="cuda:0")
    p = buf.data_ptr()
    assert lib.mtrpl_pt_decide_dev(ctx.handle, 0, 4, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0) == 0  # nothing to do
    assert lib.mtrpl_pt_decide_dev(ctx.handle, 4, 4, 2, p, p, p, p, p, p, p, p, 0) != 0  # 1 or 3 proposals
    assert "mtrpl_pt_decide_dev" in lib.mtrpl_last_error(ctx.handle).decode()
    assert lib.mtrpl_pt_decide_dev(ctx.handle, 4, 4, 3, p, p, p, p, 0, p, p, p, 0) != 0  # null pointer
    assert lib.mtrpl_pt_select_dev(ctx.handle, 0, 4, 240, 0, 0, 0, 0, 0) == 0
    assert lib.mtrpl_pt_select_dev(ctx.handle, 4, 5, 240, p, p, p, p + 2048, 0) != 0  # at most four sets
    assert lib.mtrpl_pt_select_dev(ctx.handle, 4, 2, 100, p, p, p, p + 2048, 0) != 0  # rows in 16-byte units
    assert lib.mtrpl_pt_select_dev(ctx.handle, 4, 2, 240, p + 8, p, p, p + 2048, 0) != 0  # alignment
    assert lib.mtrpl_pt_select_dev(ctx.handle, 4, 2, 240, p, 0, p, p + 2048, 0) != 0  # no chain map for several sets
    assert lib.mtrpl_pt_decide_dev(None, 4, 4, 3, p, p, p, p, p, p, p, p, 0) != 0
