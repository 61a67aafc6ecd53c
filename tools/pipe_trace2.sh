#!/bin/bash
# Run ON THE GPU BOX: kernel + memory-copy time line of tools/pipe_host_cost (C driver of the pinned pipeline).
# usage: pipe_trace2.sh <outdir> <tag> <noise|blob> <lanes> <slots> [steps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(realpath -m $1); tag=$2
mkdir -p $OUT
python3 -c "
import sys; sys.path.insert(0, '$R')
from cova_amd import weights as W
open('/tmp/w_noise.bin', 'wb').write(W.to_bytes(W.random_init(1234)))
open('/tmp/w_blob.bin', 'wb').write(W.to_bytes(W.blob_like(7)))"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptrace_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ptrace_$tag -- $R/tools/pipe_host_cost /tmp/w_$3.bin ${6:-80} $4 $5 > $OUT/trace_$tag.log 2>&1
for f in $(find /tmp/ptrace_$tag -name "*kernel_trace.csv" -o -name "*memory_copy_trace.csv"); do cp $f $OUT/${tag}_$(basename $f | sed 's/^[0-9]*_//'); done
grep frames_per_s $OUT/trace_$tag.log
