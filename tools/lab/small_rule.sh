#!/bin/bash
# The table behind small_ok(): one launch (k_small, PGM_SMALL=2: whatever the shape) against the launch sequence (PGM_SMALL=0) for ONE
# light curve, tools/evalloop <n reps need_grad q batch d>, ms per evaluation; work = n^2 x (mixture, dimension) pairs / 1000.
R=$GRAFT_REPO_ROOT; cd $R
echo "# library sha $(sha256sum pgmuvi_amd/libpgmuvi_hip.so | cut -c1-16)"
row() { a=$(PGM_SMALL=2 tools/evalloop $1 1000 1 $2 1 $3 | tail -1 | sed 's/.*: \([0-9.]*\) ms.*/\1/'); b=$(PGM_SMALL=0 tools/evalloop $1 1000 1 $2 1 $3 | tail -1 | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  echo "n=$1 q=$2 d=$3 work=$(( $1 * $1 * $2 * $3 / 1000 )): one launch $a ms, launch sequence $b ms"; }
for d in 1 2; do for n in 48 64 80 96 106 112 120 128; do for q in 1 2 3 4 6 8 16; do [ $((q * d)) -le 16 ] && row $n $q $d; done; done; done
echo "# batches of 8 and 64 light curves"
rowb() { a=$(PGM_SMALL=2 tools/evalloop $1 200 1 $2 $4 $3 | tail -1 | sed 's/.*: \([0-9.]*\) ms.*/\1/'); b=$(PGM_SMALL=0 tools/evalloop $1 200 1 $2 $4 $3 | tail -1 | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  echo "n=$1 q=$2 d=$3 batch=$4: one launch $a ms, launch sequence $b ms"; }
for B in 2 8 64; do rowb 128 4 1 $B; rowb 128 8 1 $B; rowb 128 4 2 $B; rowb 106 3 2 $B; done
