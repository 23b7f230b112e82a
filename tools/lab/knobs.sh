#!/bin/bash
# tools/lab/knobs.sh: short light curves under the schedule switches that matter there (one GPU call)
for n in 640 1000 1500 2048 3000; do
  for sub in 128 320 700 2000; do echo "N=$n PGM_LAUUM_SUB=$sub"; PGM_LAUUM_SUB=$sub timeout -k 5 60 tools/evalloop $n 30 1 | cut -c1-40; done
  for et in 0 12 32; do echo "N=$n PGM_EARLY_T=$et"; PGM_EARLY_T=$et timeout -k 5 60 tools/evalloop $n 30 1 | cut -c1-40; done
done
