#!/bin/bash
# A/B of the sixteenth-tile inverse/gradient launch (PGM_LAUUM_SUB16 = most work items of a call that takes it; 0 = never)
R=$GRAFT_REPO_ROOT; cd $R
for n in 192 256 384 512 640; do
  for s in 0 20; do
    echo -n "SUB16=$s "; PGM_LAUUM_SUB16=$s timeout -k 5 120 tools/evalloop $n 1000 1 4 1 1 | tail -1
  done
done
echo "== 2-D (q=3)"
for n in 225 250 512; do
  for s in 0 20; do
    echo -n "SUB16=$s "; PGM_LAUUM_SUB16=$s timeout -k 5 120 tools/evalloop $n 1000 1 3 1 2 | tail -1
  done
done
echo "== PF=32"
export LD_LIBRARY_PATH=$R/tools/variants/pf32:$LD_LIBRARY_PATH
for n in 256 512 1024; do echo -n "PF32 SUB16=200 "; PGM_LAUUM_SUB16=200 timeout -k 5 120 tools/evalloop $n 1000 1 4 1 1 | tail -1; done
