#!/bin/bash
# the default (PGM_SMALL=1: small_ok's table) against both forced paths, one light curve: the default should be the faster of the two
R=$GRAFT_REPO_ROOT; cd $R
t() { PGM_SMALL=$1 tools/evalloop $2 1000 1 $3 1 $4 | tail -1 | sed 's/.*: \([0-9.]*\) ms.*/\1/'; }
for d in 1 2; do for n in 80 96 106 112 120 128; do for q in 1 2 3 4 6 8; do [ $((q * d)) -le 16 ] && echo "n=$n q=$q d=$d: default $(t 1 $n $q $d)  one launch $(t 2 $n $q $d)  launch sequence $(t 0 $n $q $d)"; done; done; done
