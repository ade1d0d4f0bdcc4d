#!/bin/bash
# tools/final_profiles.sh TAG -- the bench / rocprofv3 / PMC set that profiles/ holds for a round (run on the GPU box via
# gpurun AFTER the last code commit; everything lands in gpurun_out/, tools/collect_profiles.sh copies it to profiles/)
# SECTIONS=wave,c5,pt,acc (default: all) limits the run to some of its parts.
T=$1
SEC=${SECTIONS:-wave,c5,pt,acc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# bench.py sets this itself, but under rocprofv3 the runtime is up before Python starts
export GPU_MAX_HW_QUEUES=16
O=gpurun_out
if [[ $SEC == *wave* ]]; then
# ---- counters of the two builds of the wave kernel (each group in its own pass) ----
tools/pmc.sh $T > $O/${T}_pmc.log 2>&1
tools/pmc_mix.sh $T > $O/${T}_pmc_mix.log 2>&1
# the bench line quotes VALU instructions / busy cycles / HBM bytes per solve from the committed summaries: put this
# build's in place first (tools/collect_profiles.sh makes the same copy in the repository)
cp $O/hbm_traffic_${T}_wave1.json profiles/hbm_traffic.json
cp $O/hbm_traffic_${T}_wave2.json profiles/hbm_traffic_wave2.json
cp $O/hbm_traffic_${T}_mix_wave2.json profiles/hbm_traffic_mix_wave2.json
cp $O/${T}_issue_cost.txt profiles/${T}_issue_cost.txt
# ---- the contract line (all legs: single stream, as given, sorted, fp64 linear algebra, PT) ----
timeout -k 10 500 python bench.py --steps 30 --warmup 3 > $O/${T}_bench_config2.json 2> $O/${T}_bench.err
# ... and over the K the driver times (its command line: --gpus 1 --steps 20 --warmup 5)
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pt > $O/${T}_bench_config2_k20.json 2>> $O/${T}_bench.err
# ---- kernel statistics of the timed kind of launch, and of one launch at a time ----
BENCH="python3 bench.py --steps 20 --warmup 1 --no-cpu-baseline --quick"
BENCH1="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --streams 1 --quick"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- $BENCH > $O/prof_$T.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}s1 -- $BENCH1 > $O/prof_${T}s1.log 2>&1
cp $O/prof_$T/*/*_kernel_stats.csv $O/${T}_kernel_stats_bench_config2.csv
cp $O/prof_${T}s1/*/*_kernel_stats.csv $O/${T}_kernel_stats_bench_config2_one_stream.csv
fi
if [[ $SEC == *c5* ]]; then
# ---- config 5 (nx = 512): counters of the SPIKE kernel first (the contract line quotes its traffic from the summary), then
# the contract line and the kernel statistics ----
C5="python3 bench.py --workload config5 --steps 4 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc${T}c5_fetch -- $C5 > $O/pmc${T}c5_1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc${T}c5_write -- $C5 > $O/pmc${T}c5_2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc${T}c5_sq -- $C5 > $O/pmc${T}c5_3.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc${T}c5_sq2 -- $C5 > $O/pmc${T}c5_4.log 2>&1
python tools/pmc_summary.py "solve_kernel_spike<" $O/hbm_traffic_${T}_spike4.json $O/pmc${T}c5_fetch $O/pmc${T}c5_write $O/pmc${T}c5_sq $O/pmc${T}c5_sq2 > $O/${T}_pmc_team.log 2>&1
cp $O/hbm_traffic_${T}_spike4.json profiles/${T}_pmc_spike4.json
timeout -k 10 300 python bench.py --workload config5 --steps 10 --warmup 2 > $O/${T}_bench_config5.json 2>> $O/${T}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}c5 -- $C5 > $O/prof_${T}c5.log 2>&1
cp $O/prof_${T}c5/*/*_kernel_stats.csv $O/${T}_kernel_stats_bench_config5.csv
fi
if [[ $SEC == *pt* ]]; then
# ---- the parallel-tempering ensembles: contract lines, kernel statistics, counters of the likelihood kernel ----
timeout -k 10 300 python bench.py --workload pt --chains 512 --steps 40 --warmup 10 > $O/${T}_bench_pt512_1gpu.json 2>> $O/${T}_bench.err
timeout -k 10 300 python bench.py --workload pt --chains 64 --steps 40 --warmup 10 --no-cpu-baseline > $O/${T}_bench_pt64_1gpu.json 2>> $O/${T}_bench.err
# ... and the blocks one GPU of four and of two would hold
timeout -k 10 300 python bench.py --workload pt --chains 128 --steps 40 --warmup 10 --no-cpu-baseline > $O/${T}_bench_pt128_1gpu.json 2>> $O/${T}_bench.err
timeout -k 10 300 python bench.py --workload pt --chains 256 --steps 40 --warmup 10 --no-cpu-baseline > $O/${T}_bench_pt256_1gpu.json 2>> $O/${T}_bench.err
timeout -k 10 300 python bench.py --workload pt --chains 64 --steps 40 --warmup 10 --no-cpu-baseline --groups 1 --prefetch off --decide host > $O/${T}_bench_pt64_1gpu_plain.json 2>> $O/${T}_bench.err
# ... and with the decisions on the host, as round 3 took them (two groups where prefetching, one otherwise)
timeout -k 10 300 python bench.py --workload pt --chains 64 --steps 40 --warmup 10 --no-cpu-baseline --decide host > $O/${T}_bench_pt64_1gpu_host.json 2>> $O/${T}_bench.err
timeout -k 10 300 python bench.py --workload pt --chains 512 --steps 40 --warmup 10 --no-cpu-baseline --decide host > $O/${T}_bench_pt512_1gpu_host.json 2>> $O/${T}_bench.err
PT="python3 bench.py --workload pt --chains 512 --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}pt -- $PT > $O/prof_${T}pt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}pt64 -- python3 bench.py --workload pt --chains 64 --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_${T}pt64.log 2>&1
cp $O/prof_${T}pt/*/*_kernel_stats.csv $O/${T}_kernel_stats_bench_pt512.csv
cp $O/prof_${T}pt64/*/*_kernel_stats.csv $O/${T}_kernel_stats_bench_pt64.csv
# ... the launch cycles of the groups from longer traces (per hardware queue: solve / likelihood kernel times, the gap from
# one solve to the next; per stretch between two exchanges: when each queue starts and ends), and the host's share
rocprofv3 --kernel-trace --output-format csv -d $O/prof_${T}pt64l -- python3 bench.py --workload pt --chains 64 --steps 60 --warmup 10 --no-cpu-baseline > $O/prof_${T}pt64l.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/prof_${T}pt512l -- python3 bench.py --workload pt --chains 512 --steps 30 --warmup 10 --no-cpu-baseline > $O/prof_${T}pt512l.log 2>&1
for n in 64 512; do
  { python tools/pt_timeline.py $O/prof_${T}pt${n}l/*/*_kernel_trace.csv; python tools/pt_segments.py $O/prof_${T}pt${n}l/*/*_kernel_trace.csv; } > $O/${T}_pt${n}_timeline.txt 2>/dev/null
done
{ timeout -k 10 100 python tools/pt_host_phases.py 64 100; PT_DECIDE=host PT_GROUPS=2 timeout -k 10 100 python tools/pt_host_phases.py 64 100; } > $O/${T}_pt64_host_phases.txt 2>/dev/null
# (the counters characterise the likelihood kernel by itself: one group, one 3072-trace launch of twelve-wave blocks at a
# time; the one-wave blocks of the three-group ensemble produce the same table, tests/test_gpu_loglik_wave_blocks.py)
PTL="python3 bench.py --workload pt --chains 512 --steps 5 --warmup 1 --no-cpu-baseline --groups 1 --decide host"
export MTRPL_PARTS=1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc${T}pt_sq -- $PTL > $O/pmc${T}pt_3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc${T}pt_fetch -- $PTL > $O/pmc${T}pt_1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc${T}pt_write -- $PTL > $O/pmc${T}pt_2.log 2>&1
unset MTRPL_PARTS
python tools/pmc_summary.py "loglik_kernel<" $O/hbm_traffic_${T}_loglik.json $O/pmc${T}pt_fetch $O/pmc${T}pt_write $O/pmc${T}pt_sq > $O/${T}_pmc_loglik.log 2>&1
fi
if [[ $SEC == *c5* ]]; then
# ---- config 5: fp32 / fp64 tolerance sweep on this build (the fp32 instantiation is the team kernel's) ----
timeout -k 10 400 python tools/config5_sweep.py $O/${T}_config5_sweep.json > $O/${T}_config5_sweep.txt 2>&1
# ---- where the cycles of a step go (-DMTRPL_STAMPS diagnostic build: s_memtime per phase), SPIKE at nx = 512 and the one-wave kernel
{ timeout -k 10 300 python tools/stamps.py --config5 --variant 0 --batch 512; timeout -k 10 300 python tools/stamps.py --variant 6,3 --batch 1024; } 2>/dev/null | grep -v amdgpu.ids > $O/${T}_step_phases.txt
fi
if [[ $SEC == *acc* ]]; then
# ---- the held-out converged set, where the steps of a solve go ----
timeout -k 10 200 python tools/heldout_table.py > $O/${T}_heldout.txt 2>/dev/null
{ echo "== SPIKE backend (default)"; timeout -k 10 200 python tools/heldout_table.py --large; echo "== team backend (kernel_variant 1)"; timeout -k 10 200 python tools/heldout_table.py --large kernel_variant=1; } > $O/${T}_heldout_large.txt 2>/dev/null
MTRPL_LIB_PATH=$PWD/metrotrpl_amd/lib/libmetrotrpl_amd_trace.so timeout -k 10 200 python tools/bind_trace.py 2>/dev/null | grep -v amdgpu.ids > $O/${T}_bind_trace.txt
{ timeout -k 10 100 python tools/pt_chain_granularity.py 10 64; timeout -k 10 100 python tools/pt_chain_granularity.py 10 512; } 2>/dev/null | grep -v amdgpu.ids > $O/${T}_pt_chain_granularity.txt
{ timeout -k 10 100 python tools/api_sweep.py 8192; for q in 8 16; do for a in "0 0" "3 2" "8 2"; do GPU_MAX_HW_QUEUES=$q timeout -k 10 100 python tools/api_queues.py $a; done; done; } 2>/dev/null | grep -v amdgpu.ids > $O/${T}_api_streams.txt
MTRPL_LIB_PATH=$PWD/metrotrpl_amd/lib/libmetrotrpl_amd_trace.so timeout -k 10 200 python tools/startup_trace.py > $O/${T}_startup_trace.txt 2>/dev/null
# ---- accuracy table, variant agreement, solve latency, PCIe-inclusive rate ----
timeout -k 10 200 python tools/tail_table.py --json $O/${T}_tail_decades.json --settings shipped > $O/${T}_tail_decades.txt 2>/dev/null
timeout -k 10 300 python tools/variant_agreement.py 8192 > $O/${T}_variant_agreement.txt 2>/dev/null
timeout -k 10 100 python tools/solve_latency.py > $O/${T}_solve_latency.txt 2>/dev/null
timeout -k 10 200 python tools/slow_solves.py > $O/${T}_slow_solves.txt 2>/dev/null
fi
cut -c1-260 $O/${T}_bench_*.json
head -4 $O/${T}_kernel_stats_bench_config2.csv | cut -c1-200
cat $O/${T}_solve_latency.txt
