#!/bin/bash
# Run ON THE GPU BOX: A/B of library builds with tools/quick_bench.py on ONE box (boxes differ by ~8 %, more than most
# kernel changes): carrier-frame entry with one and two lanes; AB_ENTRIES="frames stack" adds the stacked entry.
# usage: ab.sh <lib.so> [<lib.so> ...]   ("cur" = the library in the tree); name a build twice to see the run-to-run spread
#   e.g. (builds kept in an untracked ab_tmp/): tools/ab.sh ab_tmp/old.so ab_tmp/new.so ab_tmp/old.so ab_tmp/new.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/cova_amd/libcovahip.so /tmp/ab_cur.so
for v in "$@"; do
    if [ "$v" = cur ]; then cp /tmp/ab_cur.so $R/cova_amd/libcovahip.so; else cp "$v" $R/cova_amd/libcovahip.so; fi
    for e in ${AB_ENTRIES:-frames}; do
        for lanes in ${AB_LANES:-1 2}; do
            echo "== $v $e lanes=$lanes"
            QB_INPUT=$e QB_LANES=$lanes timeout -k 10 120 python3 $R/tools/quick_bench.py ${AB_BATCH:-256} ${AB_STEPS:-300} 2>&1 | grep -E "us/batch|_mfma|fused" | tr '\n' ' ' | sed 's/  */ /g'
            echo
        done
    done
done
cp /tmp/ab_cur.so $R/cova_amd/libcovahip.so
