#!/bin/bash
# tools/collect_profiles.sh TAG -- copy what tools/final_profiles.sh left in gpurun_out/ into profiles/ (tracked)
T=$1
O=gpurun_out
for f in bench_config2.json bench_config2_k20.json bench_config5.json bench_pt512_1gpu.json bench_pt64_1gpu.json bench_pt128_1gpu.json bench_pt256_1gpu.json bench_pt64_1gpu_plain.json bench_pt64_1gpu_host.json bench_pt512_1gpu_host.json \
         kernel_stats_bench_config2.csv kernel_stats_bench_config2_one_stream.csv kernel_stats_bench_config5.csv \
         kernel_stats_bench_pt512.csv kernel_stats_bench_pt64.csv tail_decades.json tail_decades.txt variant_agreement.txt \
         solve_latency.txt slow_solves.txt issue_cost.txt pt64_timeline.txt pt512_timeline.txt pt64_host_phases.txt step_phases.txt config5_sweep.json config5_sweep.txt heldout.txt heldout_large.txt bind_trace.txt pt_chain_granularity.txt api_streams.txt startup_trace.txt; do
  [ -s $O/${T}_$f ] && cp $O/${T}_$f profiles/${T}_$f
done
for k in wave1 wave2 spike4 loglik mix_wave2; do [ -s $O/hbm_traffic_${T}_$k.json ] && cp $O/hbm_traffic_${T}_$k.json profiles/${T}_pmc_$k.json; done
# the two summaries bench.py reads its counter-derived fields from
[ -s $O/hbm_traffic_${T}_wave1.json ] && cp $O/hbm_traffic_${T}_wave1.json profiles/hbm_traffic.json
[ -s $O/hbm_traffic_${T}_wave2.json ] && cp $O/hbm_traffic_${T}_wave2.json profiles/hbm_traffic_wave2.json
[ -s $O/hbm_traffic_${T}_mix_wave2.json ] && cp $O/hbm_traffic_${T}_mix_wave2.json profiles/hbm_traffic_mix_wave2.json
ls -la profiles | grep ${T}_
# the table of profiles/README.md that quotes these files is GENERATED from them
python tools/profiles_readme.py ${T}
