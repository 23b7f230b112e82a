mkdir -p gpurun_out/s2
tools/lab/pipelab > gpurun_out/s2/pipelab.txt 2>&1
for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg; echo "LEFT=$1 STRIPS=$2" >> gpurun_out/s2/ab.txt
  PGM_LEFT=$1 PGM_STRIPS=$2 tools/evalloop 2048 20 1 4 64 >> gpurun_out/s2/ab.txt 2>&1
  PGM_LEFT=$1 PGM_STRIPS=$2 tools/evalloop 4096 5 1 4 64 >> gpurun_out/s2/ab.txt 2>&1
  PGM_LEFT=$1 PGM_STRIPS=$2 tools/evalloop 1024 20 1 4 64 >> gpurun_out/s2/ab.txt 2>&1
  PGM_LEFT=$1 PGM_STRIPS=$2 tools/evalloop 256 20 1 4 1024 >> gpurun_out/s2/ab.txt 2>&1
  PGM_LEFT=$1 PGM_STRIPS=$2 tools/evalloop 2048 20 1 4 16 >> gpurun_out/s2/ab.txt 2>&1
done
echo "UPD_BIG_MIN=1024" >> gpurun_out/s2/ab.txt
PGM_UPD_BIG_MIN=1024 tools/evalloop 2048 20 1 4 64 >> gpurun_out/s2/ab.txt 2>&1
tools/selftest > gpurun_out/s2/selftest.txt 2>&1; echo "selftest rc=$?" >> gpurun_out/s2/selftest.txt
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/s2/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/s2/pytest.txt
