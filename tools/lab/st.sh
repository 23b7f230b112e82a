#!/bin/bash
# kernel statistics of one evalloop command: tools/lab/st.sh <tag> <evalloop args>   (PGM_LIBDIR=tools/variants/<name>: that build)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$R/${PGM_LIBDIR:-pgmuvi_amd}:$LD_LIBRARY_PATH
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_$tag -o s -- $R/tools/evalloop "$@" > $R/gpurun_out/st_$tag.log 2>&1 || exit 1
f=$(find $R/gpurun_out/st_$tag -name '*kernel_stats.csv' | head -1)
tail -3 $R/gpurun_out/st_$tag.log
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]: print("%-60s calls %6s total_us %10.1f avg_us %9.2f  %5s%%" % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3, r['Percentage']))
PY
