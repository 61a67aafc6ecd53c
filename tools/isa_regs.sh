#!/bin/bash
# Developer helper: device ISA of blobnet_mfma.hip into /tmp/isa/mfma.s and a table of VGPRs / scratch bytes per kernel.
mkdir -p /tmp/isa
cd "$(dirname "$0")/../cova_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-honor-nans -mllvm -pragma-unroll-threshold=1000000 -I../../include -I. \
    -S --cuda-device-only blobnet_mfma.hip -o /tmp/isa/mfma.s "$@" 2>&1 | grep -E "error|remark" -A5 | head -30
python3 - <<'PY'
import re
txt = open("/tmp/isa/mfma.s").read()
ks = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", txt)
for n, sc, v in ks:
    print("  %-62s scratch %4s vgpr %4s" % (re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n)[:62], sc, v))
PY
