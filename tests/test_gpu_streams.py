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


# ---- the pipelined host-pointer solve: sub-launches on the context's own streams ---------------
def _host_call(ctx, d, ini, t, out, with_stats=True):
    import ctypes as C
    from metrotrpl_amd import _lib as L
    B = d.shape[0]
    status = np.full(B, -7, dtype=np.int32)
    stats = np.zeros((B, 8), dtype=np.int32)
    ctx.check(L.lib().mtrpl_solve_batch(ctx.handle, B, d.ctypes.data_as(C.c_void_p), L.dptr(ini), ini.shape[0], L.dptr(t),
                                        t.shape[0], L.dptr(out), out.shape[0], L.iptr(status),
                                        L.iptr(stats) if with_stats else None))
    return status, stats


def test_pipelined_host_solve_batch_equals_the_blocking_launch():
    """mtrpl_solve_batch with more than host_chunk solves runs as sub-launches on streams the context owns, the samples
    of a finished sub-launch travelling back under the later ones' compute -- into pinned memory directly, into pageable
    memory through the staging ring.  Every sample, status and step count must equal the one blocking launch
    (host_streams = 0), for a batch that is not a multiple of the chunk, with more sub-launches than staging slots, with a
    gap in the output array, and for descriptors that do NOT fill `out` in order (fallback to one launch)."""
    from metrotrpl_amd import _lib as L, batch, workloads as W
    ctx = L.Context(0)
    B, nt = 1100, 201
    c = W.config2(B)
    t = np.ascontiguousarray(np.linspace(0, 400, nt))
    d, ini = batch.build_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], c["nx"], nt, hmax=c["hmax"],
                               meas=c["meas"], model=c["model"], ini_mode=c["ini_mode"])
    d["out_off"][700:] += 57  # a gap the kernel never writes: zero in both paths
    n_out = B * nt + 57
    ctx.set_option("host_streams", 0)
    ref = np.full(n_out, np.nan)
    st_ref, stats_ref = _host_call(ctx, d, ini, t, ref)
    assert np.all(st_ref == 0) and np.all(np.isfinite(ref)) and np.all(ref[700 * nt:700 * nt + 57] == 0)
    ctx.set_option("host_streams", 4)
    for chunk in (128, 97, 1024):  # 9 sub-launches (more than the 8 staging slots) | 12 ragged ones | 2
        ctx.set_option("host_chunk", chunk)
        for pinned in (False, True):
            out = L.pinned_f64(ctx, n_out) if pinned else np.empty(n_out)
            out[:] = np.nan
            st, stats = _host_call(ctx, d, ini, t, out)
            np.testing.assert_array_equal(out, ref)
            np.testing.assert_array_equal(st, st_ref)
            np.testing.assert_array_equal(stats[:, :7], stats_ref[:, :7])  # (slot 7 is the residency: a clock reading)
    # no stats buffer; and descriptors that run backwards through `out`
    out = np.empty(n_out)
    st, _ = _host_call(ctx, d, ini, t, out, with_stats=False)
    np.testing.assert_array_equal(out, ref)
    rev = d[::-1].copy()
    out = np.full(n_out, np.nan)
    st, _ = _host_call(ctx, rev, ini, t, out)
    np.testing.assert_array_equal(out, ref)
    np.testing.assert_array_equal(st, st_ref[::-1])
    # the batch API: an ordinary array by default, pinned memory of the context on request (big batches): same numbers
    ctx.set_option("host_chunk", 1024)
    nt_big = 1001
    tb = np.ascontiguousarray(c["tSteps"])
    sig = batch.solve_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], c["nx"], tb, ctx=ctx)
    assert sig.shape == (B, nt_big) and not isinstance(_root(sig), L.PinnedBuffer)
    pin = batch.solve_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], c["nx"], tb, ctx=ctx, pinned=True)
    assert isinstance(_root(pin), L.PinnedBuffer)
    np.testing.assert_array_equal(pin, sig)
    one = batch.solve_batch(c["states"][1000:1003], W.indexes(), W.units(), c["iniPar"][1000:1003], c["thickness"], c["nx"], tb,
                            ctx=ctx, pinned=True)  # (a small batch: pinned=True changes nothing)
    assert not isinstance(_root(one), L.PinnedBuffer)
    np.testing.assert_array_equal(sig[1000:1003], one)
    # A pinned result outlives close(): the context is marked closed -- every further call raises -- and the native
    # context (which owns the pinned memory) goes with the last buffer the caller holds (`sig = solve_batch(...);
    # ctx.close(); sig.sum()` read freed memory)
    ctx.trim()
    ctx.close()
    with pytest.raises(RuntimeError, match="closed"):
        ctx.set_option("host_chunk", 512)
    with pytest.raises(RuntimeError, match="closed"):
        batch.solve_batch(c["states"][:2], W.indexes(), W.units(), c["iniPar"][:2], c["thickness"], c["nx"], tb, ctx=ctx)
    assert ctx._h is not None and ctx._pinned_live == 1
    np.testing.assert_array_equal(pin, sig)  # still there, still the same numbers
    assert float(pin.sum()) == float(sig.sum())
    del pin
    import gc
    gc.collect()
    assert ctx._h is None and ctx._pinned_live == 0  # the last buffer took the native context with it
    # ... and a context nobody holds pinned memory of closes at once
    ctx2 = L.Context(0)
    L.pinned_f64(ctx2, 1 << 20)[:8] = 1.0  # (dropped at once: back to the context's free list)
    ctx2.close()
    assert ctx2._h is None


def _root(a):
    while getattr(a, "base", None) is not None:
        a = a.base
    return a


def test_the_build_of_the_wave_kernel_chosen_per_launch():
    """mtrpl_solve_batch_dev2: solves_per_simd = 1 / 2 picks the one- / two-solves-per-SIMD build for THAT launch, whatever
    the context's two_per_simd_min_batch says; 0 is mtrpl_solve_batch_dev; the samples are the same bits on either build;
    anything else is a bad argument.  What LikelihoodEngine.per_simd passes."""
    import torch
    from metrotrpl_amd import _lib as L, batch, workloads as W
    dev = torch.device("cuda", 0)
    ctx = L.Context(0)
    c = W.config2(96)
    t = np.ascontiguousarray(np.linspace(0, 300, 151))
    d, ini = batch.build_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], c["nx"], len(t))
    B = d.shape[0]
    dd = torch.from_numpy(d.view(np.uint8).reshape(-1)).to(dev)
    di, dt = torch.from_numpy(ini).to(dev), torch.from_numpy(t).to(dev)
    res = {}
    for per in (0, 1, 2):
        out = torch.full((B * len(t),), float("nan"), dtype=torch.float64, device=dev)
        st = torch.full((B,), -1, dtype=torch.int32, device=dev)
        sa = torch.zeros(B * 8, dtype=torch.int32, device=dev)
        ctx.check(L.lib().mtrpl_solve_batch_dev2(ctx.handle, B, dd.data_ptr(), di.data_ptr(), dt.data_ptr(), out.data_ptr(),
                                                 st.data_ptr(), sa.data_ptr(), 128, 0, torch.cuda.current_stream().cuda_stream, per))
        torch.cuda.synchronize()
        assert bool((st == 0).all())
        res[per] = (out.cpu().numpy(), sa.cpu().numpy().reshape(B, 8)[:, :7])
    for per in (1, 2):
        np.testing.assert_array_equal(res[per][0], res[0][0])
        np.testing.assert_array_equal(res[per][1], res[0][1])
    out = torch.zeros(B * len(t), dtype=torch.float64, device=dev)
    st = torch.zeros(B, dtype=torch.int32, device=dev)
    with pytest.raises(ValueError, match="solves_per_simd"):
        ctx.check(L.lib().mtrpl_solve_batch_dev2(ctx.handle, B, dd.data_ptr(), di.data_ptr(), dt.data_ptr(), out.data_ptr(),
                                                 st.data_ptr(), None, 128, 0, None, 3))
    ctx.close()


def test_nx512_host_batch_is_pipelined_in_quarter_chunks():
    """The four-wave kernel of nx > 256 takes host_chunk / 4 solves per sub-launch; same samples as one launch."""
    from metrotrpl_amd import _lib as L, batch, workloads as W
    ctx = L.Context(0)
    c = W.config5(10, 512)
    t = np.ascontiguousarray(np.linspace(0, 100, 51))
    d, ini = batch.build_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], 512, len(t), meas=c["meas"])
    B = d.shape[0]
    ctx.set_option("host_streams", 0)
    ref = np.empty(B * len(t))
    st_ref, _ = _host_call(ctx, d, ini, t, ref)
    assert np.all(st_ref == 0)
    ctx.set_option("host_streams", 3)
    ctx.set_option("host_chunk", 48)  # 12 solves per sub-launch
    out = np.empty(B * len(t))
    st, _ = _host_call(ctx, d, ini, t, out)
    np.testing.assert_array_equal(out, ref)
    np.testing.assert_array_equal(st, st_ref)
    ctx.close()


def test_likelihood_lanes_and_the_pipelined_dense_sampler():
    """LikelihoodEngine.eval_async on clones of an engine (lanes), three evaluations in flight on three streams, and
    dense_sampler.simulate over them: the table of every chunk equals the blocking eval's, whatever the lane count."""
    import torch
    from metrotrpl_amd import _lib as L, dense_sampler as DS, workloads as W
    from metrotrpl_amd.likelihood import LikelihoodEngine
    names, ix, un = W.PARAM_NAMES, W.indexes(), W.units()
    s0 = W.initial_state()
    lo = np.array([W.PRIOR_BOX.get(n, (s0[ix[n]], s0[ix[n]]))[0] for n in names])
    hi = np.array([W.PRIOR_BOX.get(n, (s0[ix[n]], s0[ix[n]]))[1] for n in names])
    X = DS.random_grid(lo, hi, [True] * len(names), 500, np.random.default_rng(11))
    t = np.linspace(0, 400, 201)
    lengths, flu = [311.0, 2000.0], np.array([W.STAUB_FLUENCES[0], W.STAUB_FLUENCES[5]])
    vals = [np.full(201, 20.0) - t / 200, np.full(201, 22.0) - t / 300]
    uncs = [np.full(201, 0.05)] * 2
    eng = LikelihoodEngine(ix, un, lengths, 128, ["TRPL", "TRPL"], [t, t], vals, uncs, {"TRPL": 1.0}, flu, sort_min=64)
    T = [1.0, 7.0, 100.0]
    ref = [eng.eval(X[a:a + 120], T) for a in range(0, 480, 120)]
    lanes = [eng, eng.clone(), eng.clone()]
    streams = L.shared_streams(torch.device("cuda", 0), 3)
    hs = [lanes[k].eval_async(X[120 * k:120 * (k + 1)], T, streams[k]) for k in range(3)]
    for k in (2, 0, 1):
        np.testing.assert_array_equal(hs[k].result(), ref[k])
    h = lanes[1].eval_async(X[360:480], T, streams[1])  # a lane again, after its result has been read
    np.testing.assert_array_equal(h.result(), ref[3])
    one = DS.simulate(eng, X, chunk=64, lanes=1)
    for n_lanes, chunk in ((3, 64), (2, 200), (4, 50)):
        np.testing.assert_array_equal(DS.simulate(eng, X, chunk=chunk, lanes=n_lanes), one)
    np.testing.assert_array_equal(one[:120], ref[0][:, :, 0].sum(axis=1))
