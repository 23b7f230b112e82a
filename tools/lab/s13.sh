mkdir -p gpurun_out/s13
timeout -k 10 300 python tools/trainbench.py > gpurun_out/s13/trainbench.txt 2>&1
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s13/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/s13/pytest.txt
