cd $GRAFT_REPO_ROOT
rowb() { a=$(PGM_SMALL=2 tools/evalloop $1 200 1 $2 $4 $3 | tail -1 | sed 's/.*: \([0-9.]*\) ms.*/\1/'); b=$(PGM_SMALL=0 tools/evalloop $1 200 1 $2 $4 $3 | tail -1 | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  echo "n=$1 q=$2 d=$3 batch=$4: one launch $a ms, launch sequence $b ms"; }
for B in 12 16 24 32 48; do rowb 128 4 1 $B; rowb 128 8 1 $B; rowb 128 4 2 $B; rowb 106 3 2 $B; rowb 112 8 1 $B; rowb 96 3 2 $B; done
