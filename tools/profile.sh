#!/bin/bash
# tools/profile.sh <tag> <evalloop args...>   (GPU box)  ->  gpurun_out/prof_<tag>/{stats,fetch,write,tcc}* and gpurun_out/prof_<tag>/traffic.json
# Kernel-trace statistics and the PMC passes of one evalloop command, each counter set in a run of its own (gpurun refuses
# --pmc combined with the trace domains other than --kernel-trace).
set -e
tag=$1; shift
cd "$(dirname "$0")/.."
d=gpurun_out/prof_$tag
mkdir -p $d
export PGM_PROFILE_WORKLOAD="tools/evalloop $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o stats -- tools/evalloop "$@" > $d/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $d -o fetch -- tools/evalloop "$@" > $d/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $d -o write -- tools/evalloop "$@" > $d/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $d -o tcc -- tools/evalloop "$@" > $d/tcc.log 2>&1
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $d -o mfma -- tools/evalloop "$@" > $d/mfma.log 2>&1
m=$(find $d -name 'mfma_counter_collection.csv' | head -1)
python3 tools/pmc_counter.py "$m" MfmaUtil > $d/mfma_util.json
f=$(find $d -name 'fetch_counter_collection.csv' | head -1); w=$(find $d -name 'write_counter_collection.csv' | head -1); t=$(find $d -name 'tcc_counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py "$f" "$w" $t > $d/traffic.json
s=$(find $d -name 'stats_kernel_stats.csv' | head -1)
cp "$s" $d/kernel_stats.csv
k=$(find $d -name 'stats_kernel_trace.csv' | head -1)
python3 tools/timeline.py "$k" > $d/timeline.txt
# keep the merge-back small: the raw per-dispatch tables go
find $d -name '*_counter_collection.csv' -delete; find $d -name '*_kernel_trace.csv' -delete
tail -2 $d/stats.log
