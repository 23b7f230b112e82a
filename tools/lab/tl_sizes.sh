#!/bin/bash
# kernel timelines of the launch sequence at mid sizes: tools/lab/tl_sizes.sh "256 512 1024" [q] [d]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in $1; do
  tag=n${n}_q${2:-4}_d${3:-1}
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tls_$tag -o t -- $R/tools/evalloop $n 50 1 ${2:-4} 1 ${3:-1} > $R/gpurun_out/tls_$tag.log 2>&1 || exit 1
  k=$(find $R/gpurun_out/tls_$tag -name '*kernel_trace.csv' | head -1)
  python3 $R/tools/timeline.py "$k" > $R/gpurun_out/tls_$tag.txt
done
