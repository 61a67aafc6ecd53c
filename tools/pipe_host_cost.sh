#!/bin/bash
# Run ON THE GPU BOX: tools/pipe_host_cost over lanes x slots for one or more library builds ("cur" = the tree's); PHC_BLOB=1: blob-like weights
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "
import sys; sys.path.insert(0, '$R')
from cova_amd import weights as W
open('/tmp/w_noise.bin', 'wb').write(W.to_bytes(W.random_init(1234)))
open('/tmp/w_blob.bin', 'wb').write(W.to_bytes(W.blob_like(7)))"
cp $R/cova_amd/libcovahip.so /tmp/ab_cur.so
for v in "$@"; do
    if [ "$v" = cur ]; then cp /tmp/ab_cur.so $R/cova_amd/libcovahip.so; else cp "$R/$v" $R/cova_amd/libcovahip.so; fi
    for w in noise blob; do
        for cfg in ${PHC_CFGS:-"1 2" "2 3" "3 3" "3 4" "3 6" "4 6"}; do
            echo -n "$v $w: "; timeout -k 10 120 $R/tools/pipe_host_cost /tmp/w_$w.bin ${PHC_STEPS:-1500} $cfg
        done
    done
done
cp /tmp/ab_cur.so $R/cova_amd/libcovahip.so
