#!/bin/bash
# Run ON THE GPU BOX: element bench + chain bench for library builds, alternating ("cur" = the tree's). usage: chain_ab.sh <lib.so|cur> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/cova_amd/libcovahip.so /tmp/ab_cur.so
for v in "$@"; do
    if [ "$v" = cur ]; then cp /tmp/ab_cur.so $R/cova_amd/libcovahip.so; else cp "$R/$v" $R/cova_amd/libcovahip.so; fi
    echo -n "$v element: "; timeout -k 10 200 bash $R/tools/element_bench.sh 20000 8 1 2>/dev/null | grep -o '"frames_per_s_through_elements": [0-9.]*'
    echo -n "$v chain records: "; CHAINBENCH_RECORDS=1 timeout -k 10 200 bash $R/tools/chain_bench.sh 60000 16 2>/dev/null | grep -o '"frames_per_s_full_chain": [0-9.]*'
    echo -n "$v native: "; timeout -k 10 200 bash $R/tools/native_chain.sh 4000 16 8 2>/dev/null | grep -o '"frames_per_s_native_chain": [0-9.]*, .*"cpu_us_per_frame": [0-9.]*'
done
cp /tmp/ab_cur.so $R/cova_amd/libcovahip.so
