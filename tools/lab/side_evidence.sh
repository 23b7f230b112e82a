#!/bin/bash
# the side measurements quoted in DESIGN.md / profiles/README.md on the current library (one GPU call)
cd $GRAFT_REPO_ROOT
o=gpurun_out/side; mkdir -p $o
export LD_LIBRARY_PATH=$PWD/pgmuvi_amd:$LD_LIBRARY_PATH
( for n in 17 89 128 256 512 1000 1024 1500 2048 2560 3000 3584 4096 4608 5120 6144 8192; do tools/evalloop $n 20 1; done
  tools/evalloop 4096 20 0
  for n in 256 1024 4096; do tools/evalloop $n 20 1 0; done ) > $o/evalloop_sizes.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > $o/bench_line.json 2> $o/bench.err
python3 bench.py --total-batch 512 --npoints 2048 --steps 5 --warmup 2 > $o/bench_line_total_batch512_n2048.json 2> $o/bench_total.err
python3 tools/trainbench.py > $o/trainbench.txt 2>&1
python3 tools/smallbench.py > $o/smallbench.json 2> $o/smallbench.err
( for sw in 1 0; do echo "PGM_SMALL=$sw"; for n in 1 17 64 89 128; do PGM_SMALL=$sw tools/evalloop $n 2000 1 2; done; PGM_SMALL=$sw tools/evalloop 89 2000 1 4; PGM_SMALL=$sw tools/evalloop 128 2000 1 4; PGM_SMALL=$sw tools/evalloop 89 2000 0 2; done ) > $o/evalloop_small_ab.txt 2>&1
for n in 8320 12288 16384; do tools/evalloop $n 3 1; done > $o/evalloop_beyond64.txt 2>&1
python3 tools/configbench.py > $o/configbench.txt 2>&1
python3 tools/batchbench.py > $o/batchbench.txt 2>&1
python3 tools/densebench.py > $o/densebench.txt 2>&1
python3 tools/raggedbench.py 512 1024 2048 3 > $o/raggedbench.txt 2>&1
python3 tools/raggedbench.py 512 200 2300 3 >> $o/raggedbench.txt 2>&1
for a in "64 1024 2048" "128 300 1500" "32 500 3000" "16 1024 2048" "12 1024 2048" "8 1024 2048"; do python3 tools/raggedbench.py $a 3; done >> $o/raggedbench.txt 2>&1
( CHAINS=1 SAMPLES=10 WARMUP=20 python3 tools/nutsbench.py 2>/dev/null | awk "NR<=2"; CHAINS=8 SAMPLES=10 WARMUP=20 python3 tools/nutsbench.py 2>/dev/null | awk "NR<=2" ) > $o/nutsbench_ticks.txt 2>&1
tools/selftest 1024 > $o/selftest.txt 2>&1
[ -x tools/lab/ratelab ] && tools/lab/ratelab > $o/ratelab.txt 2>&1
[ -x tools/lab/potrflab ] && tools/lab/potrflab 2000 > $o/potrflab.txt 2>&1
[ -x tools/lab/fetchlab ] && tools/lab/fetchlab > $o/fetchlab.txt 2>&1
# two input dimensions: the one launch against the launch sequence up to 128 points, the launch sequence beyond
ev2() { echo "$1: $($2 tools/evalloop $3 $4 1 $5 1 2 | tail -1 | sed 's/  gsum.*//')  [n reps grad q batch d = $3 $4 1 $5 1 2]"; }
( for sw in 2 0; do ev2 "PGM_SMALL=$sw" "env PGM_SMALL=$sw" 64 2000 2; ev2 "PGM_SMALL=$sw" "env PGM_SMALL=$sw" 106 2000 3; ev2 "PGM_SMALL=$sw" "env PGM_SMALL=$sw" 128 2000 4; done
  ev2 "default" "env" 225 2000 2; ev2 "default" "env" 250 2000 3; ev2 "default" "env" 512 1000 3; ev2 "default" "env" 1000 500 3; ev2 "default" "env" 2048 100 3; ev2 "default" "env" 4096 30 3 ) > $o/evalloop_2d.txt 2>&1
# the inverse/gradient launch of light curves of two to four block rows: sixteenth tiles (default) against quarter tiles
( for n in 192 256 384 512 640; do for s in 0 20; do echo -n "PGM_LAUUM_SUB16=$s "; PGM_LAUUM_SUB16=$s tools/evalloop $n 1000 1 4 1 1 | tail -1; done; done
  echo "== two input dimensions (q=3)"
  for n in 225 250 512; do for s in 0 20; do echo -n "PGM_LAUUM_SUB16=$s "; PGM_LAUUM_SUB16=$s tools/evalloop $n 1000 1 3 1 2 | tail -1; done; done ) > $o/sub16_ab.txt 2>&1
python3 tools/lab/fitrate.py 2>/dev/null | grep "^n=" > $o/fitrate.txt
sha256sum pgmuvi_amd/libpgmuvi_hip.so > $o/lib_sha.txt
