"""
make_pickle_golden.py -- an Ensemble pickle WRITTEN BY THE REFERENCE (its own
sim_utils.Ensemble.checkpoint, imported from /root/reference in the build container),
so that the loader of this package is tested against the reference's real on-disk
output: tests/golden/ref_ensemble.pik (a data file; ~20 KB).  The ensemble is the
"single" case of mcmc.npz with its reference walk filled in.

    python tests/golden/make_pickle_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _import_reference  # noqa: E402

if __name__ == "__main__":
    _, _, sim_utils, _ = _import_reference()
    z = np.load(os.path.join(HERE, "mcmc.npz"))
    cfg = json.loads(str(z["pt__config"]))
    pi, si, mf = cfg["param_info"], cfg["sim_info"], cfg["MCMC_fields"]
    pi["prior_dist"] = {k: tuple(v) for k, v in pi["prior_dist"].items()}
    mf["solver"] = tuple(mf["solver"])
    E = sim_utils.Ensemble(pi, si, mf, mf["num_iters"])
    E.H.states[:] = z["pt__states"]
    E.H.loglikelihood[:] = z["pt__logll"]
    E.H.accept[:] = z["pt__accept"]
    E.H.swap_attempts[:] = z["pt__swap_attempts"]
    E.H.swap_accept[:] = z["pt__swap_accept"]
    E.latest_iter = 12
    E.random_state = np.random.default_rng(5).bit_generator.state
    E.ensemble_fields["_times"] = [z["pt__times"], z["pt__times"]]
    E.checkpoint(os.path.join(HERE, "ref_ensemble.pik"))
    print("wrote ref_ensemble.pik", os.path.getsize(os.path.join(HERE, "ref_ensemble.pik")))
