#!/bin/bash
# Run ON THE GPU BOX: the resident step (tools/pipe_host_cost slots = 0: covahip_filter_forward_frames_packed in a loop, C driver)
# with 1 .. 4 lanes under runtime environment switches (hardware queues per process, ...).
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/tools/pipe_host_cost.sh --build-only
python3 -c "
import sys; sys.path.insert(0, '$R')
from cova_amd import weights as W
open('/tmp/w_noise.bin', 'wb').write(W.to_bytes(W.random_init(1234)))"
for rep in 1 2; do
for e in "X=1" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=3" "GPU_MAX_HW_QUEUES=6" "GPU_MAX_HW_QUEUES=8" "AMD_DIRECT_DISPATCH=0" "HIP_FORCE_DEV_KERNARG=1"; do
  for l in 1 2 3 4; do
    echo -n "$e lanes=$l: "; env $e timeout -k 10 120 $R/tools/pipe_host_cost /tmp/w_noise.bin 2000 $l 0
  done
done
done
