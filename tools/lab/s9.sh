mkdir -p gpurun_out/s9
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s9/p -o st -- tools/evalloop 4096 3 1 > gpurun_out/s9/log.txt 2>&1
f=$(find gpurun_out/s9/p -name 'st_kernel_trace.csv' | head -1); python3 tools/timeline.py "$f" > gpurun_out/s9/timeline_n4096.txt
rm -rf gpurun_out/s9/p
PGM_PLAN_DUMP=1 tools/evalloop 4096 1 1 > gpurun_out/s9/plan.txt 2>&1
TARGET=0.95 timeout -k 10 600 python tools/nutsconv.py > gpurun_out/s9/nutsconv.txt 2>&1
