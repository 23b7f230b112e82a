mkdir -p gpurun_out/s26
o=gpurun_out/s26/ab.txt
for a in "2048 30 1 4 8" "2048 30 1 4 4" "2048 30 1 4 2" "1024 50 1 4 8" "1024 50 1 4 16" "4096 10 1 4 4" "4096 10 1 4 2" "512 50 1 4 16" "512 50 1 4 32"; do
  echo "== $a : auto / PANEL=0 / PANEL=4" >> $o
  tools/evalloop $a >> $o 2>&1; PGM_PANEL=0 tools/evalloop $a >> $o 2>&1; PGM_PANEL=4 tools/evalloop $a >> $o 2>&1
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lomb or scargle or ls_" > gpurun_out/s26/pytest_ls.txt 2>&1; echo "rc=$?" >> gpurun_out/s26/pytest_ls.txt
