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


def simulate(engine, X, temperature=1.0, chunk=512):
    """Log-likelihood of every sample row of X (summed over the engine's measurements).
    `engine` is a metrotrpl_amd.likelihood.LikelihoodEngine."""
    X = np.atleast_2d(np.asarray(X, dtype=float))
    P = np.empty(X.shape[0])
    for a in range(0, X.shape[0], chunk):
        table = engine.eval(X[a:a + chunk], [temperature])   # [c, M, 1]
        P[a:a + chunk] = table[:, :, 0].sum(axis=1)
    return P
