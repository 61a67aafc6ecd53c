"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh into profiles/<tag>/traffic.json.

Per kernel, per launch:  hbm_bytes = 2 * FETCH_SIZE * 1024  +  WRITE_SIZE * 1024
(FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
streaming reads -- MI355X_MICROARCH.md, HBM section -- so it is doubled; WRITE_SIZE is exact for
16-byte stores.  The narrow 2/8-byte epilogue stores of these kernels are outside the calibrated
access widths, so the write side is indicative.)"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

tag = sys.argv[1]
dst = sys.argv[2] if len(sys.argv) > 2 else f"profiles/{tag}"
prefix = sys.argv[3] if len(sys.argv) > 3 else "stack_"   # "frames_" / "stack_": the entry point's passes of collect_profiles.sh
src = f"gpurun_out/{tag}"


def collect(sub, counter):
    acc = defaultdict(list)
    for f in glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
            acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = collect(prefix + "pmc_fetch", "FETCH_SIZE")
write = collect(prefix + "pmc_write", "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    if "rocclr" in k:
        continue
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out[k] = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
              "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024)}
json.dump(out, open(f"{dst}/traffic{'_frames' if prefix == 'frames_' else ''}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
