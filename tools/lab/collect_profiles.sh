#!/bin/bash
# copies the evidence of tools/lab/final_profiles.sh and side_evidence.sh (gpurun_out/prof_*, gpurun_out/side) into profiles/ under a round prefix:
#   tools/lab/collect_profiles.sh r05        (build container, after the GPU calls)
r=$1
for d in gpurun_out/prof_*/; do
  tag=$(basename $d | sed 's/^prof_//')
  [ -f $d/kernel_stats.csv ] && cp $d/kernel_stats.csv profiles/${r}_kernel_stats_${tag}.csv
  [ -f $d/traffic.json ] && cp $d/traffic.json profiles/${r}_pmc_hbm_traffic_${tag}.json
  [ -f $d/mfma_util.json ] && cp $d/mfma_util.json profiles/${r}_pmc_mfma_util_${tag}.json
  [ -f $d/timeline.txt ] && cp $d/timeline.txt profiles/${r}_timeline_${tag}.txt
done
[ -f gpurun_out/prof_small/phase_ticks.txt ] && cp gpurun_out/prof_small/phase_ticks.txt profiles/${r}_small_phase_ticks.txt
[ -f gpurun_out/prof_bench/bench_line.json ] && grep '^{' gpurun_out/prof_bench/bench_line.json > profiles/${r}_bench_line_under_rocprofv3.json
mv profiles/${r}_kernel_stats_bench.csv profiles/${r}_kernel_stats_bench_py.csv 2>/dev/null
s=gpurun_out/side
for f in evalloop_sizes evalloop_small_ab evalloop_beyond64 trainbench configbench batchbench densebench raggedbench nutsbench_ticks ratelab potrflab selftest fetchlab evalloop_2d sub16_ab fitrate; do
  [ -f $s/$f.txt ] && grep -v "amdgpu.ids" $s/$f.txt > profiles/${r}_$f.txt
done
[ -f $s/smallbench.json ] && cp $s/smallbench.json profiles/${r}_smallbench.json
[ -f $s/bench_line.json ] && grep '^{' $s/bench_line.json > profiles/${r}_bench_line.json
[ -f $s/bench_line_total_batch512_n2048.json ] && grep '^{' $s/bench_line_total_batch512_n2048.json > profiles/${r}_bench_line_total_batch512_n2048.json
cat gpurun_out/prof_lib_sha.txt $s/lib_sha.txt 2>/dev/null
