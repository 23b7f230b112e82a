mkdir -p gpurun_out/s14
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base noepi; do
  if [ "$v" = base ]; then d=pgmuvi_amd; else d=tools/variants/$v; fi
  for a in "2048 3 1 4 64" "4096 3 1"; do
    tag=$(echo $a | tr ' ' '_')
    LD_LIBRARY_PATH=$PWD/$d:$LD_LIBRARY_PATH rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s14/p -o st -- tools/evalloop $a > gpurun_out/s14/log.txt 2>&1
    f=$(find gpurun_out/s14/p -name 'st_kernel_stats.csv' | head -1); cp "$f" gpurun_out/s14/stats_${v}_$tag.csv
    rm -rf gpurun_out/s14/p
  done
done
