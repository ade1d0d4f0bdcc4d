#!/bin/bash
# tools/ab.sh TAG [reps] -- same-box A/B of metrotrpl_amd/lib/libbase.so (A) against the current build (B): the config-2 bench line
# without CPU / PT / api legs, interleaved; then the accuracy gates on B
cd $GRAFT_REPO_ROOT
T=$1; N=${2:-2}
for i in $(seq 1 $N); do
for v in A B; do
  if [ $v = A ]; then export MTRPL_LIB_PATH=$GRAFT_REPO_ROOT/metrotrpl_amd/lib/libbase.so; else unset MTRPL_LIB_PATH; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-pt --no-api > gpurun_out/${T}_$v$i.json 2> gpurun_out/${T}_$v$i.err
  python - <<PY
import json
d=json.load(open("gpurun_out/${T}_$v$i.json"))
c=d["config"]; a=d["accuracy"]
print("$v$i value %.0f | single %.0f (%.3f ms) | sorted %.0f | fp64 %.0f | one4096 %.0f | steps %.1f rhs %.1f max %d | acc %.2e %.2e ok=%s" % (d["value"], d["single_stream"]["solves_per_sec"], d["single_stream"]["ms_per_step"], d["sorted_by_measured_cost"]["solves_per_sec"], d["fp64_linear_algebra"]["solves_per_sec"], d["one_launch"]["4096"]["solves_per_sec"], c["mean_integrator_steps"], c["mean_rhs_evals"], c["max_integrator_steps"], a["max_rel_err_decades_0_to_5"], a["max_rel_err_decades_6_to_8"], a["all_ok"]))
PY
done; done
unset MTRPL_LIB_PATH
if [ "$3" != "notests" ]; then
python -m pytest tests -m gpu -q -x -k "solver or accuracy or heldout or tolerances or min_y or properties" > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/${T}_tests.log; tail -3 gpurun_out/${T}_tests.log
fi
