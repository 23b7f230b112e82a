#!/bin/bash
# tools/lab/sizes.sh [reps]: ms per evaluation from C for the sizes the rounds' tables quote (one light curve, then batches)
r=${1:-20}
for n in 89 256 512 1000 2048 3000 4096; do timeout -k 5 120 tools/evalloop $n $r 1 || exit 1; done
timeout -k 5 120 tools/evalloop 2048 3 1 4 64 || exit 1
timeout -k 5 120 tools/evalloop 2048 3 1 4 8 || exit 1
