mkdir -p gpurun_out/s28
timeout -k 10 300 python tools/lab/soak.py > gpurun_out/s28/soak.txt 2>&1; echo "soak rc=$?" >> gpurun_out/s28/soak.txt
PGM_FUZZ_CASES=120 PGM_FUZZ_SEED=7 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -s > gpurun_out/s28/fuzz.txt 2>&1; echo "fuzz rc=$?" >> gpurun_out/s28/fuzz.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s28/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/s28/pytest.txt
