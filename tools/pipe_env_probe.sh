#!/bin/bash
# Run ON THE GPU BOX: the pinned pipeline (tools/pipe_host_cost, 3 lanes x 6 slots) under runtime environment switches that move
# what the GPU fetches across the host link per launch (kernel arguments, queue packets) off it.
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "
import sys; sys.path.insert(0, '$R')
from cova_amd import weights as W
open('/tmp/w_blob.bin', 'wb').write(W.to_bytes(W.blob_like(7)))
open('/tmp/w_noise.bin', 'wb').write(W.to_bytes(W.random_init(1234)))"
for rep in 1 2; do
  for e in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "ROC_USE_FGS_KERNARG=0" "GPU_MAX_HW_QUEUES=8" "HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=8"; do
    for w in blob noise; do
      echo -n "$e $w: "; env $e timeout -k 10 120 $R/tools/pipe_host_cost /tmp/w_$w.bin 1500 3 6
    done
  done
done
