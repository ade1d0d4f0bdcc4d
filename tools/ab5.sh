#!/bin/bash
# tools/ab5.sh TAG [reps] -- same-box A/B of libbase.so against the current build on the config-5 line (nx = 512)
cd $GRAFT_REPO_ROOT
T=$1; N=${2:-2}
for i in $(seq 1 $N); do
for v in A B; do
  if [ $v = A ]; then export MTRPL_LIB_PATH=$GRAFT_REPO_ROOT/metrotrpl_amd/lib/libbase.so; else unset MTRPL_LIB_PATH; fi
  timeout -k 10 300 python bench.py --workload config5 > gpurun_out/${T}_$v$i.json 2> gpurun_out/${T}_$v$i.err
  python - <<PY
import json
d=json.load(open("gpurun_out/${T}_$v$i.json"))
c=d["config"]
print("$v$i config5 value %.0f (%.3f ms) | single %.0f (%.3f ms) | steps %.1f max %d ok=%s" % (d["value"], d["ms_per_step"], d["single_stream"]["solves_per_sec"], d["single_stream"]["ms_per_step"], c["mean_integrator_steps"], c["max_integrator_steps"], c["all_solves_ok"]))
PY
done; done
unset MTRPL_LIB_PATH
python -m pytest tests/test_gpu_team.py tests/test_gpu_heldout_large.py tests/test_gpu_streams.py -m gpu -q -x > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/${T}_tests.log; tail -3 gpurun_out/${T}_tests.log
