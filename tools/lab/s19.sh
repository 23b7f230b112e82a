mkdir -p gpurun_out/s19
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s19/p -o st -- tools/evalloop 2048 2 1 4 512 > gpurun_out/s19/log.txt 2>&1
f=$(find gpurun_out/s19/p -name 'st_kernel_stats.csv' | head -1); cp "$f" gpurun_out/s19/stats_b512_n2048.csv
k=$(find gpurun_out/s19/p -name 'st_kernel_trace.csv' | head -1); python3 tools/timeline.py "$k" > gpurun_out/s19/timeline_b512_n2048.txt
rm -rf gpurun_out/s19/p
