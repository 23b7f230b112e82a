#!/bin/bash
# same-session A/B of two builds of the library: tools/lab/ab.sh <variant> "<evalloop args>"...
v=$1; shift
for a in "$@"; do
  for d in tools/variants/$v pgmuvi_amd tools/variants/$v pgmuvi_amd; do
    echo "== $d : $a"
    LD_LIBRARY_PATH=$PWD/$d:$LD_LIBRARY_PATH timeout -k 5 120 tools/evalloop $a || exit 1
  done
done
