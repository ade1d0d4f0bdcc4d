#!/bin/bash
# tools/c5_check.sh TAG -- after a change to the SPIKE backend: its parity tests, the config-5 contract line, and the two
# HBM counter passes of its kernel (each in its own run); stops at the first step that fails
T=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
O=gpurun_out
python -m pytest tests -m gpu -q -x -k "team or heldout_large or spike" > $O/${T}_tests.log 2>&1
rc=$?; tail -3 $O/${T}_tests.log; [ $rc = 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload config5 --steps 10 --warmup 2 > $O/${T}_c5.json 2> $O/${T}_c5.err || exit 1
cut -c1-700 $O/${T}_c5.json
C5="python3 bench.py --workload config5 --steps 4 --warmup 1"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${T}_write -- $C5 > $O/${T}_w.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${T}_fetch -- $C5 > $O/${T}_f.log 2>&1 || exit 1
python tools/pmc_summary.py "solve_kernel_spike<" $O/${T}_spike4.json $O/${T}_fetch $O/${T}_write | head -12
