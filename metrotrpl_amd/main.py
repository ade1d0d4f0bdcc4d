"""Launcher with the reference's main.py flow (main.py:14-52): read the config script,
the initial conditions and the measurements, run `metro`, leave `CPU{jobid}-final.pik`
in the output directory.

    python -m metrotrpl_amd.main [script_head [jobid]]            one GPU
    python -m torch.distributed.run --nproc-per-node N -m metrotrpl_amd.main ...   chains sharded over N GPUs

`script_head` defaults to "mcmc", `jobid` to 0 (-> mcmc0.txt), as in the reference.
"""
import sys

import numpy as np

from .bayes_io import get_data, get_initpoints, read_config_script_file
from .metropolis import metro


def run(script_head="mcmc", jobid=0, **kwargs):
    script_path = f"{script_head}{jobid}.txt"
    export_path = f"CPU{jobid}-final.pik"
    sim_info, param_info, meas_fields, MCMC_fields = read_config_script_file(script_path)
    np.random.seed(jobid)
    iniPar = get_initpoints(MCMC_fields["init_cond_path"], meas_fields)
    e_data = get_data(MCMC_fields["measurement_path"], meas_fields, MCMC_fields)
    ens = metro(sim_info, iniPar, e_data, MCMC_fields, param_info, verbose=False, export_path=export_path,
                logger_name=f"Ensemble{jobid}", **kwargs)
    print(f"{jobid} Finished - {export_path}")
    return ens


if __name__ == "__main__":
    head = sys.argv[1] if len(sys.argv) > 1 else "mcmc"
    job = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    run(head, job)
