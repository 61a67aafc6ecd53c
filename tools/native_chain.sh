#!/bin/bash
# Run ON THE GPU BOX: BASELINE config 4 through the C-ABI alone (tools/native_chain_bench.c: covahip_pipe + per-stream
# covahip_gopfilter, no GStreamer), blob-like weights.  usage: native_chain.sh [batches] [streams] [threads]
#        native_chain.sh --build-only   (what __graft_entry__.build() calls)
# The binary is rebuilt whenever its source, the C-ABI header or the library is newer than it: a binary that is stale against the
# current ABI never runs.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
BIN=$R/tools/native_chain_bench
if [ ! -x $BIN ] || [ $R/tools/native_chain_bench.c -nt $BIN ] || [ $R/include/covahip.h -nt $BIN ] || [ $R/cova_amd/libcovahip.so -nt $BIN ]; then
    gcc -O2 -fopenmp -I$R/include $R/tools/native_chain_bench.c -o $BIN -L$R/cova_amd -lcovahip '-Wl,-rpath,$ORIGIN/../cova_amd' -lm || exit 1
fi
[ "$1" = "--build-only" ] && exit 0
python3 -c "
import sys; sys.path.insert(0, '$R')
from cova_amd import weights as W
open('/tmp/covahip_wblob.bin', 'wb').write(W.to_bytes(W.blob_like(7)))"
$BIN /tmp/covahip_wblob.bin ${1:-4000} ${2:-16} ${3:-16}
