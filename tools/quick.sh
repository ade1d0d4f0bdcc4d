#!/bin/bash
# tools/quick.sh TAG [pytest -k expr] -- dev loop on the GPU box: key parity tests + the config-2 bench line (no CPU / PT legs)
T=$1; K=${2:-"solver or accuracy or tolerances or team or min_y"}
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "$K" > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/${T}_tests.log; tail -4 gpurun_out/${T}_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-pt > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${T}_bench.json"))
c=d["config"]
print("as-drawn x3 %.0f/s %.3f ms | steps %.1f rhs %.1f max %d" % (d["value"], d["ms_per_step"], c["mean_integrator_steps"], c["mean_rhs_evals"], c["max_integrator_steps"]))
for k in ("single_stream","sorted_by_measured_cost","fp64_linear_algebra"):
    if d.get(k): print("%-24s %.0f/s %.3f ms ok=%s" % (k, d[k]["solves_per_sec"], d[k]["ms_per_step"], d[k]["all_solves_ok"]))
a=d["accuracy"]; print("acc", ["%.2e"%x for x in a["max_rel_err_per_decade_0_to_8"]], a["all_ok"])
PY
