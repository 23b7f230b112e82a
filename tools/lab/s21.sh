mkdir -p gpurun_out/s21
o=gpurun_out/s21/ab.txt
for d in 4 6 8 16; do echo "PANEL=$d" >> $o
  PGM_PANEL=$d tools/evalloop 2048 5 1 4 512 >> $o 2>&1
  PGM_PANEL=$d tools/evalloop 4096 2 1 4 256 >> $o 2>&1
done
echo "PANEL=8 INLEFT=0" >> $o; PGM_INLEFT=0 PGM_PANEL=8 tools/evalloop 2048 5 1 4 512 >> $o 2>&1
echo "PANEL=4 INLEFT=0" >> $o; PGM_INLEFT=0 PGM_PANEL=4 tools/evalloop 2048 5 1 4 512 >> $o 2>&1
