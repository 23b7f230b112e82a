mkdir -p gpurun_out/s16
o=gpurun_out/s16/ab.txt
for cfg in "0 4" "1 4" "0 8" "1 8" "1 16" "1 6"; do set -- $cfg
  echo "INLEFT=$1 PANEL=$2" >> $o
  for a in "2048 20 1 4 64" "4096 5 1 4 64" "1024 50 1 4 64"; do PGM_INLEFT=$1 PGM_PANEL=$2 tools/evalloop $a >> $o 2>&1; done
  for big in 512 1024; do echo " UPD_BIG_MIN=$big" >> $o; PGM_UPD_BIG_MIN=$big PGM_INLEFT=$1 PGM_PANEL=$2 tools/evalloop 2048 20 1 4 64 >> $o 2>&1; done
done
