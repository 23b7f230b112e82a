mkdir -p gpurun_out/s10
o=gpurun_out/s10/ab.txt
for c in 1 0; do echo "COMBINE=$c" >> $o
for a in "4096 50 1" "3000 50 1" "2048 50 1" "1024 100 1" "5120 20 1" "8192 5 1"; do PGM_COMBINE=$c tools/evalloop $a >> $o 2>&1; done; done
for kc in 2 3 4 6 8; do echo "COMBINE=1 KC=$kc" >> $o; PGM_LAUUM_KC=$kc tools/evalloop 4096 50 1 >> $o 2>&1; done
tools/selftest > gpurun_out/s10/selftest.txt 2>&1; echo "selftest rc=$?" >> gpurun_out/s10/selftest.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s10/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/s10/pytest.txt
