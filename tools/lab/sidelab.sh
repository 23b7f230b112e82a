#!/bin/bash
# Same-session A/B of the side queue of the inverse pass (PGM_SIDE_*): tools/lab/sidelab.sh <n> <reps> "<env settings>"...
# Every setting is run twice, alternating with the others; evalloop prints ms per evaluation, the value and two gradient fingerprints.
n=$1; reps=$2; shift 2
for pass in 1 2; do
  for cfg in "$@"; do
    echo "== n=$n pass $pass : ${cfg:-default}"
    env $cfg timeout -k 5 120 tools/evalloop $n $reps 1 || exit 1
  done
done
