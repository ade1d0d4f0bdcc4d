#!/bin/bash
# tools/amp_sweep.sh TAG "k=v,k2=v2" ... -- one policy setting after the other through everything that gates accuracy: the config-2
# line (steps, slowest solve, per-decade table), the config-2 / config-3 tail table, the three nx <= 128 held-out sets, the nx > 128 one
cd $GRAFT_REPO_ROOT
T=$1; shift
for o in none "$@"; do
  echo "=== $o"
  tools/opt_sweep.sh $T "$(echo $o | tr ',' ' ')"
  if [ "$o" = none ]; then s=shipped; else s=$o; fi
  timeout -k 10 200 python tools/tail_table.py --settings "$s" 2>/dev/null | grep -v amdgpu.ids | grep -i "max\|worst\|decade" | head -12
done
timeout -k 10 300 python tools/heldout_both.py "$@" 2>/dev/null | grep -v amdgpu.ids
timeout -k 10 300 python tools/heldout_table.py --large "$@" 2>/dev/null | grep -v amdgpu.ids | grep "steps"
