#!/bin/bash
for a in "2048 30" "2560 30" "3000 20" "3584 20" "4096 20" "2048 10 1 4 8" "1024 10 1 4 8" "4096 5 1 4 3"; do
  for t in 0 1 33 65 129 2000; do echo -n "evalloop $a PGM_TRSM16=$t: "; PGM_TRSM16=$t timeout -k 5 60 tools/evalloop $a | cut -c1-60; done
done
