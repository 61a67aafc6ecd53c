#!/bin/bash
# Developer helper (here): builds the library of a git revision into ab_tmp/<name>.so (for tools/ab.sh on the GPU box).
# usage: tools/mk_ab.sh <git-rev> <name>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/mk_ab_$2
rm -rf $T && mkdir -p $T $R/ab_tmp
git -C $R archive $1 cova_amd/csrc include | tar -x -C $T
make -C $T/cova_amd/csrc -j8 > $T/build.log 2>&1 || { tail -20 $T/build.log; exit 1; }
cp $T/cova_amd/libcovahip.so $R/ab_tmp/$2.so
ls -la $R/ab_tmp/$2.so
