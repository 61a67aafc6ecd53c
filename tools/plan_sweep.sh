#!/bin/bash
# Run ON THE GPU BOX: encoder band plans (covahip_blobnet_set_enc_plan) through tools/quick_bench.py, carrier-frame entry.
# usage: plan_sweep.sh "1:8:2" "2:8:2,3:2:2" ...   (level:nbands:nbuf, comma-separated sets allowed)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for plan in "" "$@"; do
    echo "== plan '${plan}'"
    QB_PLAN="$plan" QB_INPUT=${QB_INPUT:-frames} timeout 120 python3 $R/tools/quick_bench.py ${AB_BATCH:-256} 30 2>&1 | grep -E "us/batch|enc|rror" | tr '\n' ' ' | sed 's/  */ /g'
    echo
done
