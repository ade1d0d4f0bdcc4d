#!/bin/bash
# tools/pmc.sh TAG -- rocprofv3 counter passes (each group in its own run, --kernel-trace only) for the two builds of
# the wave kernel: one 1024-solve launch at a time (wave1) and one 2048-solve launch at a time (wave2)
T=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
O=gpurun_out
B1="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --streams 1 --quick"
B2="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --streams 1 --quick --batch 2048 --opt kernel_variant=3"
for W in 1 2; do
  if [ $W = 1 ]; then B=$B1; K="solve_kernel_wave1<"; else B=$B2; K="solve_kernel_wave2<"; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc${T}w${W}_fetch -- $B > $O/pmc${T}w${W}_1.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc${T}w${W}_write -- $B > $O/pmc${T}w${W}_2.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc${T}w${W}_sq -- $B > $O/pmc${T}w${W}_3.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc${T}w${W}_sq2 -- $B > $O/pmc${T}w${W}_4.log 2>&1
  python tools/pmc_summary.py "$K" $O/hbm_traffic_${T}_wave${W}.json $O/pmc${T}w${W}_fetch $O/pmc${T}w${W}_write $O/pmc${T}w${W}_sq $O/pmc${T}w${W}_sq2 | grep -A14 '"derived"'
done
