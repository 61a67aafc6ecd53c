#!/bin/bash
# Run ON THE GPU BOX: BASELINE config 4 as a throughput number (gst_element_driver chainbench): S streams of 1080p carrier
# frames -> blobnetfilter batch-size=256 -> per-stream cova (embedded SORT + GoP frame filter, the experiment's parameters:
# maxage 60 / minhits 30 / iou 0.1, cc-threshold 1; experiment/cova/config.yaml:59,67) -> counting sink.  Blob-like weights
# (cova_amd.weights.blob_like): a trained BlobNet's masks, a few boxes per frame.
# usage: chain_bench.sh [frames per stream] [streams]
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-6000}
S=${2:-16}
python3 -c "
import sys; sys.path.insert(0, '$R')
from cova_amd import weights as W
open('/tmp/covahip_wblob.bin', 'wb').write(W.to_bytes(W.blob_like(7)))"
export GST_PLUGIN_PATH=$R/gst GST_PLUGIN_SYSTEM_PATH=/opt/conda/lib/gstreamer-1.0 LD_LIBRARY_PATH=/opt/conda/lib
export GST_REGISTRY=/tmp/covahip_gst_registry.bin LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6 GST_DEBUG=1
$R/gst/gst_element_driver chainbench "blobnetfilter model-weights-file=/tmp/covahip_wblob.bin batch-size=256 batched-push-timeout=0 cc-threshold=1 max-boxes=256" \
  "sort-maxage=60 sort-minhits=30 sort-iou=0.1" $S 1920 1088 $N
