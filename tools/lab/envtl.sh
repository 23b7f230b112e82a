#!/bin/bash
# launch timeline of one evalloop command under environment switches: tools/lab/envtl.sh <tag> "<evalloop args>" "<env settings>"
tag=$1; args=$2
R=$GRAFT_REPO_ROOT
for kv in $3; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$tag -o t -- $R/tools/evalloop $args > $R/gpurun_out/tl_$tag.log 2>&1 || exit 1
k=$(find $R/gpurun_out/tl_$tag -name '*kernel_trace.csv' | head -1)
python3 $R/tools/timeline.py "$k" > $R/gpurun_out/tl_$tag.txt
