#!/bin/bash
# tools/lab/mkvariant.sh <name> [git-rev]: builds the library of a git revision (default HEAD) into tools/variants/<name>/ for tools/lab/ab.sh
set -e
name=$1; rev=${2:-HEAD}
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/pgmuvi_amd/csrc $tmp/include $root/tools/variants/$name
for f in $(git -C $root ls-tree --name-only $rev pgmuvi_amd/csrc/); do git -C $root show $rev:$f > $tmp/$f; done
git -C $root show $rev:include/pgmuvi_hip.h > $tmp/include/pgmuvi_hip.h
make -C $tmp/pgmuvi_amd/csrc OUT=$root/tools/variants/$name/libpgmuvi_hip.so > /dev/null
rm -rf $tmp
ls -la $root/tools/variants/$name/
