#!/bin/bash
# Same-session A/B of environment switches of the library: tools/lab/envab.sh "<evalloop args>" "<env settings>"...
# Every setting is run twice, alternating with the others ("" = the defaults); evalloop prints ms per evaluation, the value and
# (one light curve) two gradient fingerprints.
args=$1; shift
for pass in 1 2; do
  for cfg in "$@"; do
    echo "== evalloop $args | pass $pass : ${cfg:-default}"
    env $cfg timeout -k 5 120 tools/evalloop $args || exit 1
  done
done
