#!/bin/bash
# queues of the last evaluation's launches under the side queue: tools/lab/sidetrace.sh <tag> <n> "<env settings>"  (rocprofv3 kernel trace + lab/overlap.py)
tag=$1; n=$2; shift 2
R=$GRAFT_REPO_ROOT
for kv in $1; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/st_$tag -o t -- $R/tools/evalloop $n 4 1 > $R/gpurun_out/st_$tag.log 2>&1 || exit 1
k=$(find $R/gpurun_out/st_$tag -name '*kernel_trace.csv' | head -1)
python3 $R/tools/lab/overlap.py "$k" 80 > $R/gpurun_out/st_$tag.txt
