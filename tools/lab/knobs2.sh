#!/bin/bash
for n in 2560 3584 4096; do
  for sub in 128 700 2000; do echo "N=$n PGM_LAUUM_SUB=$sub"; PGM_LAUUM_SUB=$sub timeout -k 5 60 tools/evalloop $n 30 1 | cut -c1-40; done
done
for n in 2048 2560 3000 3584 4096; do
  for et in 12 32 64 128; do echo "N=$n PGM_EARLY_T=$et"; PGM_EARLY_T=$et timeout -k 5 60 tools/evalloop $n 30 1 | cut -c1-40; done
done
