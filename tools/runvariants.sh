#!/bin/bash
# tools/runvariants.sh <n> <reps> <variant>...   : ms/eval of tools/evalloop against each tools/variants/<variant>/libpgmuvi_hip.so
n=$1; reps=$2; shift 2
for v in "$@"; do
  if [ "$v" = base ]; then d=pgmuvi_amd; else d=tools/variants/$v; fi
  echo "== $v"
  LD_LIBRARY_PATH=$PWD/$d:$LD_LIBRARY_PATH timeout -k 5 120 tools/evalloop $n $reps 1 || exit 1
done
