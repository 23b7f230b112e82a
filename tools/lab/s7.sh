mkdir -p gpurun_out/s7
o=gpurun_out/s7/ab.txt
for v in base staged fillstaged fill6; do
  if [ "$v" = base ]; then d=pgmuvi_amd; else d=tools/variants/$v; fi
  echo "== $v" >> $o
  for a in "4096 50 1" "2048 50 1" "1024 100 1" "2048 20 1 4 64" "4096 5 1 4 64" "2048 50 1 4 8" "4096 10 1 4 8" "8192 5 1"; do
    PGM_LEFT=0 LD_LIBRARY_PATH=$PWD/$d:$LD_LIBRARY_PATH timeout -k 5 120 tools/evalloop $a >> $o 2>&1
  done
done
echo "== base LEFT=1" >> $o
for a in "2048 20 1 4 64" "4096 5 1 4 64"; do PGM_LEFT=1 tools/evalloop $a >> $o 2>&1; done
