"""Dense random sampling of the prior box, evaluated in large batches on the GPU --
the batched counterpart of the reference's Dense_Sample/dense_sampling.py
(`random_grid` :16-33, `simulate` :48-195: per sample solve -> IRF -> log-likelihood).

The reference loops over samples one solve at a time; here every chunk of samples is
one launch of the solve kernel plus one launch of the fused likelihood kernel.
"""
import numpy as np


def random_grid(min_X, max_X, do_log, num_samples, rng=None):
    """dense_sampling.py:16-33: uniform (or log-uniform where do_log) draws inside the box;
    a parameter with min == max is held fixed."""
    rng = np.random.default_rng() if rng is None else rng
    min_X, max_X = np.asarray(min_X, dtype=float), np.asarray(max_X, dtype=float)
    grid = np.empty((num_samples, len(min_X)))
    for i in range(len(min_X)):
        if min_X[i] == max_X[i]:
            grid[:, i] = min_X[i]
        elif do_log[i]:
            grid[:, i] = 10 ** rng.uniform(np.log10(min_X[i]), np.log10(max_X[i]), num_samples)
        else:
            grid[:, i] = rng.uniform(min_X[i], max_X[i], num_samples)
    return grid


def simulate(engine, X, temperature=1.0, chunk=1024, lanes=3):
    """Log-likelihood of every sample row of X (summed over the engine's measurements).
    `engine` is a metrotrpl_amd.likelihood.LikelihoodEngine.

    The samples are independent, so the chunks are pipelined: up to `lanes` chunks are in flight on streams of their
    own (clones of the engine: same device constants, own launch buffers) -- while chunk k's likelihood kernel runs and
    its table travels back, the solves of chunks k+1, ... hold the SIMDs, and the slots a chunk's short solves free go
    to the next chunk's (a solve holds its SIMD for its whole duration; durations spread 1 : 1.5).  lanes=1: one chunk
    at a time, each waited for (round 4's loop); the values do not depend on `lanes` or `chunk`."""
    X = np.atleast_2d(np.asarray(X, dtype=float))
    n = X.shape[0]
    P = np.empty(n)
    starts = list(range(0, n, chunk))
    lanes = max(1, min(int(lanes), len(starts)))
    if lanes == 1:
        for a in starts:
            table = engine.eval(X[a:a + chunk], [temperature])   # [c, M, 1]
            P[a:a + chunk] = table[:, :, 0].sum(axis=1)
        return P
    from . import _lib as L
    pool = engine.__dict__.setdefault("_sampler_lanes", [])
    while len(pool) < lanes:
        pool.append(engine.clone())
    streams = L.shared_streams(engine.dev, lanes)
    # every chunk shares the device with the other lanes' chunks: more solves than SIMDs in flight whatever a chunk's
    # size, so each lane's launches take the two-solves-per-SIMD build of the wave kernel (same results) -- said per
    # launch (LikelihoodEngine.per_simd -> mtrpl_solve_batch_dev2): the lanes are this sampler's own clones, nothing of
    # the context other engines share is touched
    for lane in pool[:lanes]:
        lane.per_simd = 2
    pending = [None] * lanes
    for k, a in enumerate(starts):
        q = k % lanes
        if pending[q] is not None:
            a0, h = pending[q]
            P[a0:a0 + chunk] = h.result()[:, :, 0].sum(axis=1)
        pending[q] = (a, pool[q].eval_async(X[a:a + chunk], [temperature], streams[q]))
    order = sorted((p for p in pending if p is not None), key=lambda p: p[0])
    for a0, h in order:
        P[a0:a0 + chunk] = h.result()[:, :, 0].sum(axis=1)
    return P
