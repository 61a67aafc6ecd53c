#!/bin/bash
# Run ON THE GPU BOX: bboxcc over objects per frame x first-pass run capacity of the wave kernel:
# ns/frame, fraction of the 8 TB/s HBM peak and the fraction of frames that overflow pass 1 / pass 2.
# cap 0 = automatic (128, or 512 once a quarter of the previous call's frames had more than 128 runs).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-bboxcc_caps}
mkdir -p $OUT
for cap in 0 128 256 512; do
  SWEEP_CAP=$cap SWEEP_KINDS=${KINDS:-obj6,obj20,obj50,obj100,blobs,noise} SWEEP_BATCHES=${BATCHES:-256,65536} \
    SWEEP_OUT=$OUT/sweep_cap$cap.json timeout -k 10 300 python3 $R/tools/bboxcc_sweep.py || exit 1
done
