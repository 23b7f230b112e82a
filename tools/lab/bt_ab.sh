#!/bin/bash
# tools/lab/bt_ab.sh: config 5's tick (8 chains x N=2048 through pgm_pot_*) and the bare batched evaluation under different budgets of
# update tiles in the row-solve launch's tail (PGM_BT; default 96; 0 = every planned tile rides a k_diag launch).  Three rounds, interleaved.
cd $GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$PWD/pgmuvi_amd:$LD_LIBRARY_PATH
for round in 1 2 3; do
  for bt in 0 32 64 96 160 256; do
    echo "round $round PGM_BT=$bt: $(PGM_BT=$bt tools/evalloop 2048 30 1 4 8 2>&1 | tail -1 | cut -c1-60) | $(PGM_BT=$bt CHAINS=8 SAMPLES=2 WARMUP=2 python3 tools/nutsbench.py 2>/dev/null | head -1)"
  done
done
