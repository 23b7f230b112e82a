# the round's evidence on the final library: kernel stats, PMC traffic, MfmaUtil, timeline per workload (tools/profile.sh), and
# the kernel statistics of bench.py itself.  tools/lab/final_profiles.sh [a|b]: the first / second half (one GPU call each)
set -e
cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
part=${1:-abc}
if [[ $part == *a* ]]; then
bash tools/profile.sh single_n4096 4096 5 1
bash tools/profile.sh cfg4_n8192_d2 8192 3 1 3 1 2
bash tools/profile.sh cfg5_tick_8x2048 2048 5 1 4 8
bash tools/profile.sh batch64_n2048 2048 3 1 4 64
mkdir -p gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o stats -- python3 bench.py --steps 20 --warmup 3 --no-extra --no-cpu > gpurun_out/prof_bench/bench_line.json 2> gpurun_out/prof_bench/bench.err
s=$(find gpurun_out/prof_bench -name 'stats_kernel_stats.csv' | head -1); cp "$s" gpurun_out/prof_bench/kernel_stats.csv
find gpurun_out/prof_bench -name '*_kernel_trace.csv' -delete
fi
if [[ $part == *b* ]]; then
bash tools/profile.sh batch512_n2048 2048 2 1 4 512
bash tools/profile.sh batch256_n4096 4096 1 1 4 256
# the ragged batch of tools/raggedbench.py (512 light curves, N ~ U{1024..2048}: one trimmed launch set): kernel statistics and timeline
mkdir -p gpurun_out/prof_ragged512
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ragged512 -o stats -- python3 tools/lab/ragprof.py 512 1024 2048 3 > gpurun_out/prof_ragged512/stats.log 2>&1
s=$(find gpurun_out/prof_ragged512 -name 'stats_kernel_stats.csv' | head -1); cp "$s" gpurun_out/prof_ragged512/kernel_stats.csv
k=$(find gpurun_out/prof_ragged512 -name 'stats_kernel_trace.csv' | head -1); python3 tools/timeline.py "$k" > gpurun_out/prof_ragged512/timeline.txt
find gpurun_out/prof_ragged512 -name '*_kernel_trace.csv' -delete
fi
sha256sum pgmuvi_amd/libpgmuvi_hip.so > gpurun_out/prof_lib_sha.txt
if [[ $part == *c* ]]; then
# round 6: the one-launch path (k_small) -- kernel statistics of the reference's published workload through train_native and of
# plain evaluations at N = 17 / 89 / 128, and the in-kernel phase timeline of the lab build with -DPGM_SMALL_STAMPS (tools/variants/stamps)
mkdir -p gpurun_out/prof_small
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small -o stats -- python3 tools/smallprof.py > gpurun_out/prof_small/stats.log 2>&1
s=$(find gpurun_out/prof_small -name 'stats_kernel_stats.csv' | head -1); head -4 "$s" > gpurun_out/prof_small/kernel_stats.csv
find gpurun_out/prof_small -name '*_kernel_trace.csv' -delete
if [ -f tools/variants/stamps/libpgmuvi_hip.so ]; then
  ( export LD_LIBRARY_PATH=$PWD/tools/variants/stamps:$LD_LIBRARY_PATH
    for a in "1 2 1 2" "17 2 1 2" "89 2 1 2" "89 2 0 2" "89 2 1 4" "128 2 1 2" "128 2 1 4"; do tools/evalloop $a 2>&1 | grep k_small | tail -1; done ) > gpurun_out/prof_small/phase_ticks.txt 2>&1
fi
fi
