mkdir -p gpurun_out/s27
o=gpurun_out/s27/ab.txt
echo "default" >> $o; tools/evalloop 2048 30 1 4 8 >> $o 2>&1
for bt in 0 32 64 128 192; do echo "BT=$bt" >> $o; PGM_BT=$bt tools/evalloop 2048 30 1 4 8 >> $o 2>&1; done
echo "LAZY=0" >> $o; PGM_LAZY=0 tools/evalloop 2048 30 1 4 8 >> $o 2>&1
echo "PAIRS=0" >> $o; PGM_PAIRS=0 tools/evalloop 2048 30 1 4 8 >> $o 2>&1
echo "LOOKAHEAD=99" >> $o; PGM_LOOKAHEAD=99 tools/evalloop 2048 30 1 4 8 >> $o 2>&1
PGM_PLAN_DUMP=1 tools/evalloop 2048 1 1 4 8 2>&1 | grep "plan k" | head -16 >> $o
