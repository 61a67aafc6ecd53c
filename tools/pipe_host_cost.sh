#!/bin/bash
# Run ON THE GPU BOX: tools/pipe_host_cost over lanes x slots for one or more library builds ("cur" = the tree's); PHC_BLOB=1: blob-like weights
#   pipe_host_cost.sh --build-only      (what __graft_entry__.build() calls; rebuilt whenever source / header / library are newer)
#   pipe_host_cost.sh --bench            one JSON line for bench.py: 3 lanes x 6 slots and the same batch resident, noise + blob-like weights
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
BIN=$R/tools/pipe_host_cost
if [ ! -x $BIN ] || [ $R/tools/pipe_host_cost.c -nt $BIN ] || [ $R/include/covahip.h -nt $BIN ] || [ $R/cova_amd/libcovahip.so -nt $BIN ]; then
    gcc -O2 -I$R/include $R/tools/pipe_host_cost.c -o $BIN -L$R/cova_amd -lcovahip '-Wl,-rpath,$ORIGIN/../cova_amd' || exit 1
fi
[ "$1" = "--build-only" ] && exit 0
python3 -c "
import sys; sys.path.insert(0, '$R')
from cova_amd import weights as W
open('/tmp/w_noise.bin', 'wb').write(W.to_bytes(W.random_init(1234)))
open('/tmp/w_blob.bin', 'wb').write(W.to_bytes(W.blob_like(7)))"
if [ "$1" = "--bench" ]; then
    python3 - <<PY
import json, subprocess
out = {}
for w in ("noise", "blob"):
    for name, cfg in (("pipe_3_lanes_6_slots", ("3", "6")), ("resident_3_lanes", ("3", "0"))):
        r = subprocess.run(["$BIN", "/tmp/w_%s.bin" % w, "1500", *cfg], capture_output=True, text=True, timeout=120)
        try:
            out["%s_%s" % (w, name)] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            out["%s_%s" % (w, name)] = {"error": (r.stderr or r.stdout)[-200:]}
for w in ("noise", "blob"):
    a, b = out.get(w + "_pipe_3_lanes_6_slots", {}), out.get(w + "_resident_3_lanes", {})
    if "frames_per_s" in a and "frames_per_s" in b:
        out[w + "_pcie_inclusive_over_resident"] = round(a["frames_per_s"] / b["frames_per_s"], 3)
print(json.dumps(out))
PY
    exit 0
fi
cp $R/cova_amd/libcovahip.so /tmp/ab_cur.so
for v in "$@"; do
    if [ "$v" = cur ]; then cp /tmp/ab_cur.so $R/cova_amd/libcovahip.so; else cp "$R/$v" $R/cova_amd/libcovahip.so; fi
    for w in noise blob; do
        IFS=';' read -ra CFGS <<< "${PHC_CFGS:-1 2;2 3;3 3;3 4;3 6;4 6}"   # "lanes slots" pairs
        for cfg in "${CFGS[@]}"; do
            echo -n "$v $w: "; timeout -k 10 120 $R/tools/pipe_host_cost /tmp/w_$w.bin ${PHC_STEPS:-1500} $cfg
        done
    done
done
cp /tmp/ab_cur.so $R/cova_amd/libcovahip.so
