#!/bin/bash
# kernel timelines of one evalloop command under two builds: tools/lab/tl.sh <variant> <evalloop args>
v=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in tools/variants/$v pgmuvi_amd; do
  tag=$(echo $d | tr '/' '_')_$1
  export LD_LIBRARY_PATH=$R/$d:$LD_LIBRARY_PATH
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$tag -o t -- $R/tools/evalloop "$@" > $R/gpurun_out/tl_$tag.log 2>&1 || exit 1
  k=$(find $R/gpurun_out/tl_$tag -name '*kernel_trace.csv' | head -1)
  python3 $R/tools/timeline.py "$k" > $R/gpurun_out/tl_$tag.txt
done
