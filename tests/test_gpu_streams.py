"""Stream semantics of the device-resident entry points (include/metrotrpl_amd.h): launches on different
streams of one context may be in flight together.  The nx > 256 kernels keep their elimination blocks in a
slab of device memory -- one slab per launch, handed out from a pool -- so two concurrent launches must
give exactly what the same launches give one after the other."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batch(dev, n_states, nx, seed_off=0, t_end=300.0, nt=151):
    import torch
    from metrotrpl_amd import _lib as L, batch, workloads as W
    c = W.config5(n_states + seed_off, nx)
    sl = slice(4 * seed_off, 4 * (n_states + seed_off))
    t = np.ascontiguousarray(np.linspace(0, t_end, nt))
    d, ini = batch.build_batch(c["states"][sl], W.indexes(), W.units(), c["iniPar"][sl], c["thickness"][sl], nx, len(t),
                               meas=c["meas"][4 * seed_off:4 * (n_states + seed_off)])
    B = d.shape[0]
    return dict(B=B, nt=len(t), descs=torch.from_numpy(d.view(np.uint8).reshape(-1)).to(dev),
                ini=torch.from_numpy(ini).to(dev), t=torch.from_numpy(t).to(dev),
                out=torch.zeros(B * len(t), dtype=torch.float64, device=dev),
                status=torch.zeros(B, dtype=torch.int32, device=dev), stats=torch.zeros(B * 8, dtype=torch.int32, device=dev))


def _launch(ctx, b, nx, stream):
    from metrotrpl_amd import _lib as L
    ctx.check(L.lib().mtrpl_solve_batch_dev(ctx.handle, b["B"], b["descs"].data_ptr(), b["ini"].data_ptr(), b["t"].data_ptr(),
                                            b["out"].data_ptr(), b["status"].data_ptr(), b["stats"].data_ptr(), nx, 0,
                                            stream.cuda_stream))


def test_two_nx512_batches_on_two_streams_equal_the_serial_results():
    import torch
    from metrotrpl_amd import _lib as L
    dev = torch.device("cuda", 0)
    ctx = L.Context(0)
    nx = 512
    a, b = _batch(dev, 6, nx), _batch(dev, 6, nx, seed_off=6)
    s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    # serial: one after the other on one stream
    _launch(ctx, a, nx, s0)
    _launch(ctx, b, nx, s0)
    torch.cuda.synchronize()
    ref_a, ref_b = a["out"].cpu().numpy().copy(), b["out"].cpu().numpy().copy()
    assert np.all(a["status"].cpu().numpy() == 0) and np.all(b["status"].cpu().numpy() == 0)
    assert not np.array_equal(ref_a, ref_b)
    for rep in range(3):
        a["out"].zero_()
        b["out"].zero_()
        torch.cuda.synchronize()
        _launch(ctx, a, nx, s0)   # both in flight: 24 + 24 workgroups of 256 threads, far below the machine
        _launch(ctx, b, nx, s1)
        _launch(ctx, a, nx, s1)   # and the same batch again behind the other one, into the same output
        torch.cuda.synchronize()
        np.testing.assert_array_equal(a["out"].cpu().numpy(), ref_a)
        np.testing.assert_array_equal(b["out"].cpu().numpy(), ref_b)
    ctx.close()


def test_fp32_nx512_launches_on_a_fresh_context_grow_the_slab_pool_safely():
    """precision = 32, nx = 512, B = 256: the per-solve block slab of the fp32 team kernel comes from the context's
    pool.  (Round 3 handed out pointers into a std::vector of slabs; a call that took a second slab -- the issue-order
    permutation of that round -- could reallocate it under the first, ADVICE r3.  The pool is a deque now, outgrown
    slabs are parked instead of hipFree'd inside the launch path, and the ordering kernels are gone.)  Two such launches
    on two streams of a FRESH context, twice: every solve must have run (status written) and both passes must agree bit
    for bit; then the same launches on swapped streams with a third queued behind (pool reuse)."""
    import torch
    from metrotrpl_amd import _lib as L
    dev = torch.device("cuda", 0)
    ctx = L.Context(0)
    ctx.set_option("precision", 32)
    ctx.set_option("rtol_force", 1e-3)  # the fp32 build only integrates at loose tolerances
    ctx.set_option("max_steps", 4000)   # ... and a solve it cannot do stops early instead of crawling (status 1)
    nx = 512
    a, b = _batch(dev, 64, nx, t_end=20.0, nt=21), _batch(dev, 64, nx, seed_off=64, t_end=20.0, nt=21)
    assert a["B"] == 256 and b["B"] == 256
    s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    outs = []
    for rep in range(2):
        for q in (a, b):
            q["out"].fill_(float("nan"))
            q["status"].fill_(-1)
        torch.cuda.synchronize()
        _launch(ctx, a, nx, s0)
        _launch(ctx, b, nx, s1)
        torch.cuda.synchronize()
        for q in (a, b):
            st = q["status"].cpu().numpy()
            assert np.all(st >= 0), "a solve never ran"
            assert np.mean(st == 0) > 0.5  # fp32 at rtol 1e-3: most solves finish inside the step budget
        outs.append((a["out"].cpu().numpy().copy(), b["out"].cpu().numpy().copy()))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    # and the same two launches once more, now on swapped streams, with a third queued behind: pool reuse
    _launch(ctx, a, nx, s1)
    _launch(ctx, b, nx, s0)
    _launch(ctx, a, nx, s0)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(a["out"].cpu().numpy(), outs[0][0])
    np.testing.assert_array_equal(b["out"].cpu().numpy(), outs[0][1])
    ctx.close()
