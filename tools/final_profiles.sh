#!/bin/bash
# tools/final_profiles.sh TAG -- the bench / rocprofv3 / PMC set that profiles/ holds for a round (run on the GPU box via gpurun)
T=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
# the command whose line is the headline (two streams, longest solves first) and the same steps on one stream
BENCH="python3 bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-single-stream"
BENCH1="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --streams 1 --sort-by-cost 0"
timeout -k 10 400 python bench.py --steps 30 --warmup 3 > $O/${T}_bench_config2.json 2> $O/${T}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- $BENCH > $O/prof_$T.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}s1 -- $BENCH1 > $O/prof_${T}s1.log 2>&1
# counters: one launch at a time on the device (--streams 1), each group in its own pass
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc${T}_fetch -- $BENCH1 > $O/pmc${T}1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc${T}_write -- $BENCH1 > $O/pmc${T}2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc${T}_sq -- $BENCH1 > $O/pmc${T}3.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc${T}_sq2 -- $BENCH1 > $O/pmc${T}4.log 2>&1
python tools/pmc_summary.py "solve_kernel_wave1<" $O/hbm_traffic_$T.json $O/pmc${T}_fetch $O/pmc${T}_write $O/pmc${T}_sq $O/pmc${T}_sq2 | tail -12
# the same counters for the two-solves-per-SIMD kernel of the timed region (one 2048-solve launch at a time)
BENCH2="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --streams 1 --sort-by-cost 0 --batch 2048 --opt kernel_variant=3"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc${T}w2_fetch -- $BENCH2 > $O/pmc${T}w2_1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc${T}w2_write -- $BENCH2 > $O/pmc${T}w2_2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc${T}w2_sq -- $BENCH2 > $O/pmc${T}w2_3.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc${T}w2_sq2 -- $BENCH2 > $O/pmc${T}w2_4.log 2>&1
python tools/pmc_summary.py "solve_kernel_wave2<" $O/hbm_traffic_wave2_$T.json $O/pmc${T}w2_fetch $O/pmc${T}w2_write $O/pmc${T}w2_sq $O/pmc${T}w2_sq2 | tail -12
timeout -k 10 300 python bench.py --workload pt --chains 512 --steps 20 --warmup 10 > $O/${T}_bench_pt512_1gpu.json 2>> $O/${T}_bench.err
timeout -k 10 300 python bench.py --workload pt --chains 64 --steps 40 --warmup 10 --no-cpu-baseline > $O/${T}_bench_pt64_1gpu.json 2>> $O/${T}_bench.err
timeout -k 10 300 python bench.py --workload pt --chains 64 --steps 40 --warmup 10 --no-cpu-baseline --groups 1 > $O/${T}_bench_pt64_1gpu_groups1.json 2>> $O/${T}_bench.err
timeout -k 10 300 python bench.py --workload pt --chains 64 --steps 40 --warmup 10 --no-cpu-baseline --groups 1 --prefetch off > $O/${T}_bench_pt64_1gpu_plain.json 2>> $O/${T}_bench.err
timeout -k 10 100 python bench.py --no-cpu-baseline --steps 24 --streams 1 --sort-by-cost 0 > $O/${T}_bench_streams1.json 2>> $O/${T}_bench.err
timeout -k 10 100 python bench.py --no-cpu-baseline --steps 24 --streams 2 > $O/${T}_bench_streams2.json 2>> $O/${T}_bench.err
timeout -k 10 100 python bench.py --no-cpu-baseline --steps 24 --streams 3 --opt kernel_variant=6 > $O/${T}_bench_streams3_one_per_simd.json 2>> $O/${T}_bench.err
timeout -k 10 100 python bench.py --no-cpu-baseline --steps 24 --streams 1 --sort-by-cost 0 --opt kernel_variant=2 > $O/${T}_bench_streams1_fp64_blocks.json 2>> $O/${T}_bench.err
timeout -k 10 300 python tools/variant_agreement.py 8192 > $O/${T}_variant_agreement.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}pt -- python3 bench.py --workload pt --chains 512 --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_${T}pt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}pt64 -- python3 bench.py --workload pt --chains 64 --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_${T}pt64.log 2>&1
cp $O/prof_$T/*/*_kernel_stats.csv $O/${T}_kernel_stats_bench_config2.csv
cp $O/prof_${T}s1/*/*_kernel_stats.csv $O/${T}_kernel_stats_bench_config2_one_stream.csv
cp $O/prof_${T}pt/*/*_kernel_stats.csv $O/${T}_kernel_stats_bench_pt512.csv
cp $O/prof_${T}pt64/*/*_kernel_stats.csv $O/${T}_kernel_stats_bench_pt64.csv
timeout -k 10 100 python tools/sched_trace.py 64 > $O/${T}_sched_trace_pt64.txt 2>/dev/null
timeout -k 10 200 python tools/tail_table.py --json $O/${T}_tail_decades.json --settings shipped > $O/${T}_tail_decades.txt 2>/dev/null
timeout -k 10 200 python tools/stamps.py --variant 6,2,4 --batch 1024 > $O/${T}_stamps.txt 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $O/microbench tools/microbench.hip 2>/dev/null && timeout -k 5 120 $O/microbench 256 > $O/${T}_microbench.txt 2>&1; rm -f $O/microbench
timeout -k 10 600 python tools/config5_sweep.py $O/${T}_config5_sweep.json > $O/${T}_config5_sweep.txt 2>/dev/null
timeout -k 10 100 python tools/solve_latency.py > $O/${T}_solve_latency.txt 2>/dev/null
cut -c1-200 $O/${T}_bench_*.json
head -3 $O/${T}_kernel_stats_bench_config2.csv | cut -c1-200
cat $O/${T}_solve_latency.txt
