mkdir -p gpurun_out/s8
o=gpurun_out/s8/ab.txt
for a in "2048 20 1 4 64" "4096 5 1 4 64" "1024 50 1 4 64" "256 50 1 4 1024" "2048 20 1 4 16"; do tools/evalloop $a >> $o 2>&1; done
echo "STRIPS=0" >> $o
for a in "2048 20 1 4 64" "4096 5 1 4 64"; do PGM_STRIPS=0 tools/evalloop $a >> $o 2>&1; done
tools/selftest > gpurun_out/s8/selftest.txt 2>&1; echo "selftest rc=$?" >> gpurun_out/s8/selftest.txt
timeout -k 10 600 python tools/nutsconv.py > gpurun_out/s8/nutsconv.txt 2>&1; echo "rc=$?" >> gpurun_out/s8/nutsconv.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s8/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/s8/pytest.txt
