#!/bin/bash
# Run ON THE GPU BOX: frames/s through the GStreamer batching element (gst/gstblobnetfilter.c): 8 decoder-branch threads
# push 1080p carrier frames (68x120 macroblock records) into blobnetfilter batch-size=256; boxes come out per stream.
# usage: element_bench.sh [frames per stream] [streams] [cc-threshold] [weights: random | blob]
# (random BlobNet weights give ~500 one-macroblock boxes per frame at cc-threshold 1, a few at 30; "blob" =
#  cova_amd.weights.blob_like, nothing on muxbench's object-free noise frames: the floor of the per-frame host cost)
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-4000}
S=${2:-8}
CC=${3:-1}
KIND=${4:-random}
python3 -c "
import sys; sys.path.insert(0, '$R')
from cova_amd import weights as W
open('/tmp/covahip_w1234.bin', 'wb').write(W.to_bytes(W.blob_like(7) if '$KIND' == 'blob' else W.random_init(1234)))"
export GST_PLUGIN_PATH=$R/gst GST_PLUGIN_SYSTEM_PATH=/opt/conda/lib/gstreamer-1.0 LD_LIBRARY_PATH=/opt/conda/lib
export GST_REGISTRY=/tmp/covahip_gst_registry.bin LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6 GST_DEBUG=1
$R/gst/gst_element_driver muxbench "blobnetfilter model-weights-file=/tmp/covahip_w1234.bin batch-size=256 batched-push-timeout=0 cc-threshold=$CC max-boxes=2048" $S 1920 1088 $N
