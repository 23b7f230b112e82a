mkdir -p gpurun_out/s22
o=gpurun_out/s22/ab.txt
tools/evalloop 4096 50 1 >> $o 2>&1
for kc in 0 4 8 16; do echo "LAUUM_SUB=2000 KC=$kc" >> $o; if [ $kc = 0 ]; then PGM_LAUUM_SUB=2000 tools/evalloop 4096 50 1 >> $o 2>&1; else PGM_LAUUM_SUB=4000 PGM_LAUUM_KC=$kc tools/evalloop 4096 50 1 >> $o 2>&1; fi; done
echo "3000:" >> $o; tools/evalloop 3000 50 1 >> $o 2>&1; PGM_LAUUM_SUB=4000 tools/evalloop 3000 50 1 >> $o 2>&1
echo "2048:" >> $o; tools/evalloop 2048 50 1 >> $o 2>&1; PGM_LAUUM_SUB=4000 tools/evalloop 2048 50 1 >> $o 2>&1
