#!/bin/bash
# developer helper: time the kernels with parts of the encoder kernels disabled
for d in 0 1 2 3 4; do echo "== COVAHIP_DBG=$d"; COVAHIP_DBG=$d python tools/quick_bench.py 256 10 2>&1 | grep -E "enc[0123]_mfma"; done
