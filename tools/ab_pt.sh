#!/bin/bash
# tools/ab_pt.sh TAG ENVVAR [chains] [reps] -- same-box A/B of the parallel-tempering line: A with ENVVAR=1 (an experiment
# switch of pt_bench.py that turns something OFF), B without; interleaved
cd $GRAFT_REPO_ROOT
T=$1; V=$2; C=${3:-64}; N=${4:-2}
for i in $(seq 1 $N); do
for v in A B; do
  if [ $v = A ]; then export $V=1; else unset $V; fi
  timeout -k 10 300 python bench.py --workload pt --chains $C --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/${T}_$v$i.json 2> gpurun_out/${T}_$v$i.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/${T}_$v$i.json")); c=d["config"]
print("$v$i ($V=%s) %.3f ms per iteration | %.0f proposals/s | exchange_ms %s | accept %.3f" % ("1" if "$v"=="A" else "-", d["ms_per_step"], d["value"], c.get("exchange_ms"), c.get("acceptance_rate", float("nan"))))
PY
done; done
unset $V
