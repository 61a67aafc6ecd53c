#!/bin/bash
# Run ON THE GPU BOX: aggregate frames/s of the carrier-frame hot path with the batch split over N contexts / streams
# (tools/stream_probe.c) and L lanes per context; spec = "B contexts lanes".
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "
import sys; sys.path.insert(0, '$R')
from cova_amd import weights as W
open('/tmp/covahip_w1234.bin', 'wb').write(W.to_bytes(W.random_init(1234)))"
for spec in ${SPECS:-"64 1 1" "128 1 1" "256 1 1" "512 1 1" "1024 1 1" "256 1 2" "256 1 3" "256 1 4" "128 1 2" "512 1 2" "256 2 1" "256 2 2"}; do
  set -- $spec
  timeout -k 10 120 $R/tools/stream_probe /tmp/covahip_w1234.bin $1 $2 300 $3 || exit 1
done
