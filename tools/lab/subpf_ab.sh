#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for n in 640 768 1024 1536 2048 2560 3000; do
  echo -n "PF8  "; tools/evalloop $n 500 1 4 1 1 | tail -1
  echo -n "PF16 "; LD_LIBRARY_PATH=$R/tools/variants/subpf16:$LD_LIBRARY_PATH tools/evalloop $n 500 1 4 1 1 | tail -1
done
