#!/bin/bash
# Run ON THE GPU BOX: BASELINE config 4 through the C-ABI alone (tools/native_chain_bench.c: covahip_pipe + per-stream
# covahip_gopfilter, no GStreamer), blob-like weights.  usage: native_chain.sh [batches] [streams] [threads]
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "
import sys; sys.path.insert(0, '$R')
from cova_amd import weights as W
open('/tmp/covahip_wblob.bin', 'wb').write(W.to_bytes(W.blob_like(7)))"
[ -x $R/tools/native_chain_bench ] || gcc -O2 -fopenmp -I$R/include $R/tools/native_chain_bench.c -o $R/tools/native_chain_bench -L$R/cova_amd -lcovahip -Wl,-rpath,$R/cova_amd -lm
$R/tools/native_chain_bench /tmp/covahip_wblob.bin ${1:-4000} ${2:-16} ${3:-16}
