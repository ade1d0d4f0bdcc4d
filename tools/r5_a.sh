#!/bin/bash
# round 5, first GPU call: the pipelined host path + lanes (tests), then the bench line with the api leg
cd $GRAFT_REPO_ROOT
T=r05_a
python -m pytest tests/test_gpu_streams.py tests/test_gpu_dense_sampler.py tests/test_gpu_c_client.py tests/test_gpu_pt_kernels.py -m gpu -q -x > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/${T}_tests.log; tail -5 gpurun_out/${T}_tests.log
timeout -k 10 400 python bench.py --no-cpu-baseline --no-pt > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench exit $?"
python - <<PY
import json
d=json.load(open("gpurun_out/${T}_bench.json"))
print("value %.0f single %.0f" % (d["value"], d["single_stream"]["solves_per_sec"]))
print(json.dumps(d.get("api"), indent=1))
PY
