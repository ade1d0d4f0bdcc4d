#!/bin/bash
# tools/opt_sweep.sh TAG "k=v k2=v2" "k=v" ... -- the config-2 bench line (quick) per option set, run-time-option kernels
cd $GRAFT_REPO_ROOT
T=$1; shift
for o in "$@"; do
  args=""; for kv in $o; do [ "$kv" != "none" ] && args="$args --opt $kv"; done
  [ "$o" = "none" ] && args="--opt max_steps=2000001"
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-pt --no-api $args > gpurun_out/${T}_o.json 2> gpurun_out/${T}_o.err
  python - <<PY
import json
d=json.load(open("gpurun_out/${T}_o.json"))
c=d["config"]; a=d["accuracy"]
print("%-40s value %.0f | single %.0f | steps %.1f rhs %.1f max %d | acc %s ok=%s" % ("$o", d["value"], (d["single_stream"] or {}).get("solves_per_sec",0), c["mean_integrator_steps"], c["mean_rhs_evals"], c["max_integrator_steps"], " ".join("%.1e"%x for x in a["max_rel_err_per_decade_0_to_8"]), a["all_ok"]))
PY
done
