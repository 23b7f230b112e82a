mkdir -p gpurun_out/s23
o=gpurun_out/s23/ab.txt
tools/evalloop 4096 50 1 >> $o 2>&1
for kc in 12 14 16 20 32; do echo "KC=$kc" >> $o; PGM_LAUUM_KC=$kc tools/evalloop 4096 50 1 >> $o 2>&1; done
for n in 1536 2048 2560 3000 3584; do echo "n=$n default / LAUUM_SUB=4000" >> $o; tools/evalloop $n 50 1 >> $o 2>&1; PGM_LAUUM_SUB=4000 tools/evalloop $n 50 1 >> $o 2>&1; done
PGM_PLAN_DUMP=1 tools/evalloop 2048 1 1 2>&1 | tail -3 >> $o
