#!/bin/bash
# per-kernel totals of one evalloop command under two builds of the library, same box: tools/lab/abstats.sh <variant> <evalloop args>
v=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in tools/variants/$v pgmuvi_amd tools/variants/$v pgmuvi_amd; do
  export LD_LIBRARY_PATH=$PWD/$d
  rm -rf gpurun_out/abst; mkdir -p gpurun_out/abst
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abst -o s -- tools/evalloop "$@" > gpurun_out/abst/log.txt 2>&1
  echo "== $d"; grep "ms/" gpurun_out/abst/log.txt
  python3 - $(find gpurun_out/abst -name "s_kernel_stats.csv") <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    n = r["Name"]; i = n.find("k_"); n = n[i:i + 28] if i >= 0 else n[:28]
    print(f"   {n:30s} calls {r['Calls']:>4s}  total {int(r['TotalDurationNs']) / 1e6:9.3f} ms  avg {float(r['AverageNs']) / 1e3:9.1f} us")
PY
done
rm -rf gpurun_out/abst
