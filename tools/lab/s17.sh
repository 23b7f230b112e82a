mkdir -p gpurun_out/s17
o=gpurun_out/s17/ab.txt
for b in 64 128 256 512; do tools/evalloop 2048 5 1 4 $b >> $o 2>&1; done
for b in 64 128 256; do tools/evalloop 4096 3 1 4 $b >> $o 2>&1; done
for b in 32 48 8 16 24; do tools/evalloop 2048 10 1 4 $b >> $o 2>&1; done
