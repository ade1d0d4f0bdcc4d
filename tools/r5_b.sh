#!/bin/bash
cd $GRAFT_REPO_ROOT
T=r05_b
python -m pytest tests/test_gpu_heldout_large.py tests/test_gpu_team.py tests/test_gpu_streams.py -m gpu -q -x -s > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/${T}_tests.log; grep -v "^$" gpurun_out/${T}_tests.log | tail -60
for q in 8 16; do
GPU_MAX_HW_QUEUES=$q timeout -k 10 400 python bench.py --no-cpu-baseline --no-pt > gpurun_out/${T}_bench_q$q.json 2> gpurun_out/${T}_bench_q$q.err; echo "bench exit $?"
python - <<PY
import json
d=json.load(open("gpurun_out/${T}_bench_q$q.json"))
a=d["api"]
print("queues $q value %.0f single %.0f | api solve_batch pinned %.0f pageable %.0f python %.0f | dense %.0f" % (d["value"], d["single_stream"]["solves_per_sec"], a["solve_batch"]["pinned_out"]["solves_per_sec"], a["solve_batch"]["pageable_out"]["solves_per_sec"], a["solve_batch"]["python_batch_solve_batch"]["solves_per_sec"], a["dense_sampler"]["samples_per_sec"]))
PY
done
