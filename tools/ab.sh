#!/bin/bash
# Run ON THE GPU BOX: A/B of library builds with tools/quick_bench.py (carrier-frame and stacked entry).
# usage: ab.sh <lib.so> [<lib.so> ...]   ("cur" = the library in the tree)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/cova_amd/libcovahip.so /tmp/ab_cur.so
for v in "$@"; do
    if [ "$v" = cur ]; then cp /tmp/ab_cur.so $R/cova_amd/libcovahip.so; else cp "$v" $R/cova_amd/libcovahip.so; fi
    for e in frames stack; do
        echo "== $v $e"
        QB_INPUT=$e timeout 120 python3 $R/tools/quick_bench.py ${AB_BATCH:-256} 30 2>&1 | grep -E "us/batch|_mfma" | tr '\n' ' ' | sed 's/  */ /g'
        echo
    done
done
cp /tmp/ab_cur.so $R/cova_amd/libcovahip.so
