#!/bin/bash
for n in 2560 3000 3584 4096; do
  echo "== n=$n default"; tools/evalloop $n 100 1 | cut -c1-50
  for le in 160 240 400 480; do echo -n "LAZY_END=$le: "; PGM_LAZY_END=$le tools/evalloop $n 100 1 | cut -c1-50; done
  for bh in 64 192; do echo -n "BH=$bh: "; PGM_BH=$bh tools/evalloop $n 100 1 | cut -c1-50; done
  for bt in 48 144; do echo -n "BT=$bt: "; PGM_BT=$bt tools/evalloop $n 100 1 | cut -c1-50; done
  echo -n "LAZY=0: "; PGM_LAZY=0 tools/evalloop $n 100 1 | cut -c1-50
  echo -n "EARLY=0: "; PGM_EARLY=0 tools/evalloop $n 100 1 | cut -c1-50
done
