mkdir -p gpurun_out/s12
timeout -k 10 300 python tools/trainbench.py > gpurun_out/s12/trainbench.txt 2>&1
timeout -k 10 200 python tools/trainprof.py 1024 300 > gpurun_out/s12/trainprof_1024.txt 2>&1
timeout -k 10 200 python tools/loopgap.py 1024 > gpurun_out/s12/loopgap_1024.txt 2>&1
