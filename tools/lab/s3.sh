mkdir -p gpurun_out/s3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "1 2048" "0 2048" "1 4096" "0 4096"; do set -- $cfg
  PGM_LEFT=$1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s3/p_$1_$2 -o st -- tools/evalloop $2 4 1 4 64 > gpurun_out/s3/log_$1_$2.txt 2>&1
  f=$(find gpurun_out/s3/p_$1_$2 -name 'st_kernel_stats.csv' | head -1); cp "$f" gpurun_out/s3/stats_left$1_n$2.csv
  rm -rf gpurun_out/s3/p_$1_$2
done
