#!/bin/bash
# tools/ab.sh LIB_A LIB_B [batches] -- the config-2 launch time of two builds of the library, alternating (dev tool,
# run on the GPU box): python tools/variant_bench.py under MTRPL_LIB_PATH
B=${3:-1024,384}
for rep in 1 2; do
  for lib in "$1" "$2"; do
    echo "== $lib"
    MTRPL_LIB_PATH=$PWD/$lib python tools/variant_bench.py --variants 2 --batches $B --reps 8 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   B=%d  %.3f ms  %.0f solves/s  steps %.0f/%d' % (d['B'], d['ms'], d['solves_per_s'], d['steps_mean'], d['steps_max']))"
  done
done
