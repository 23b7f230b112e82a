mkdir -p gpurun_out/s20
o=gpurun_out/s20/ab.txt
for v in base prev; do
  if [ "$v" = base ]; then d=pgmuvi_amd; else d=tools/variants/$v; fi
  echo "== $v" >> $o
  for a in "4096 50 1" "2048 50 1" "1024 100 1" "2048 5 1 4 512" "2048 20 1 4 64" "4096 3 1 4 256" "8192 5 1 3 1 2" "2048 30 1 4 8"; do
    LD_LIBRARY_PATH=$PWD/$d:$LD_LIBRARY_PATH timeout -k 5 120 tools/evalloop $a >> $o 2>&1
  done
done
tools/selftest > gpurun_out/s20/selftest.txt 2>&1; echo "selftest rc=$?" >> gpurun_out/s20/selftest.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s20/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/s20/pytest.txt
