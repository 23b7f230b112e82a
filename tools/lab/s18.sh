mkdir -p gpurun_out/s18
o=gpurun_out/s18/ab.txt
tools/evalloop 4096 20 1 >> $o 2>&1
tools/evalloop 8192 5 1 3 1 2 >> $o 2>&1
tools/evalloop 2048 20 1 4 8 >> $o 2>&1
tools/evalloop 2048 5 1 4 512 >> $o 2>&1
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s18/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/s18/pytest.txt
timeout -k 10 400 python bench.py > gpurun_out/s18/bench.json 2> gpurun_out/s18/bench.err; echo "bench rc=$?" >> gpurun_out/s18/bench.err
