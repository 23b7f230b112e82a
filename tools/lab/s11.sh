mkdir -p gpurun_out/s11
o=gpurun_out/s11/ab.txt
for v in base staged; do
  if [ "$v" = base ]; then d=pgmuvi_amd; else d=tools/variants/$v; fi
  echo "== $v" >> $o
  for a in "4096 50 1" "2048 50 1" "1024 100 1" "256 200 1" "2048 20 1 4 64" "4096 5 1 4 64" "2048 50 1 4 8" "256 50 1 4 1024" "8192 5 1"; do
    LD_LIBRARY_PATH=$PWD/$d:$LD_LIBRARY_PATH timeout -k 5 120 tools/evalloop $a >> $o 2>&1
  done
done
for kc in 3 4 5 6 8 10; do echo "KC=$kc" >> $o; PGM_LAUUM_KC=$kc tools/evalloop 4096 50 1 >> $o 2>&1; done
tools/selftest > gpurun_out/s11/selftest.txt 2>&1; echo "selftest rc=$?" >> gpurun_out/s11/selftest.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s11/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/s11/pytest.txt
