#!/bin/bash
# Run ON THE GPU BOX: A/B of kernel-chain switches (covahip_blobnet_set_impl names, QB_IMPL) of ONE library build with
# tools/quick_bench.py, alternating, carrier-frame entry, one and two lanes.
# usage: ab_impl.sh <impl> [<impl> ...]    e.g. tools/ab_impl.sh mfma enc23_separate mfma enc23_separate
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
    for lanes in ${AB_LANES:-1 2}; do
        echo "== $v lanes=$lanes"
        QB_IMPL=$v QB_INPUT=frames QB_LANES=$lanes timeout -k 10 120 python3 $R/tools/quick_bench.py ${AB_BATCH:-256} ${AB_STEPS:-300} 2>&1 | grep -E "us/batch|_mfma|fused" | tr '\n' ' ' | sed 's/  */ /g'
        echo
    done
done
