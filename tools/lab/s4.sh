mkdir -p gpurun_out/s4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PGM_LEFT=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s4/p -o st -- tools/evalloop 2048 2 1 4 64 > gpurun_out/s4/log.txt 2>&1
f=$(find gpurun_out/s4/p -name 'st_kernel_trace.csv' | head -1); python3 tools/timeline.py "$f" > gpurun_out/s4/timeline_b64_n2048.txt
rm -rf gpurun_out/s4/p
PGM_LEFT=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s4/p -o st -- tools/evalloop 4096 2 1 4 64 > gpurun_out/s4/log.txt 2>&1
f=$(find gpurun_out/s4/p -name 'st_kernel_trace.csv' | head -1); python3 tools/timeline.py "$f" > gpurun_out/s4/timeline_b64_n4096.txt
rm -rf gpurun_out/s4/p
