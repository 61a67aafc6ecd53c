"""Developer helper: resolves the raw samples of `CHAINBENCH_PROF=1 CHAINBENCH_PROF_RAW=<file>` (gst/gst_element_driver.c: SIGPROF
sampling, "library offset" per sample) to function names with nm -- static functions included -- for the libraries that exist
in this tree under the same relative path (libcovahip.so, libgstcova.so, the driver); others are counted per library."""
import bisect
import collections
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOCAL = {"libcovahip.so": os.path.join(ROOT, "cova_amd", "libcovahip.so"), "libgstcova.so": os.path.join(ROOT, "gst", "libgstcova.so"),
         "gst_element_driver": os.path.join(ROOT, "gst", "gst_element_driver")}
tables = {}


def table(path):
    if path not in tables:
        syms = []
        out = subprocess.run(["nm", "-C", "--defined-only", "-n", path], capture_output=True, text=True).stdout
        if not out.strip():   # stripped: the dynamic symbol table is what is left
            out = subprocess.run(["nm", "-C", "-D", "--defined-only", "-n", path], capture_output=True, text=True).stdout
        for line in out.splitlines():
            parts = line.split(None, 2)
            if len(parts) == 3 and parts[1] in "tTwWiu":
                syms.append((int(parts[0], 16), parts[2]))
        tables[path] = syms
    return tables[path]


cnt = collections.Counter()
n = 0
by_thread = collections.Counter()
for line in open(sys.argv[1]):
    parts = line.split()
    lib, off = parts[0], parts[1]
    thread = parts[2] if len(parts) > 2 else "?"
    by_thread[thread] += 1
    n += 1
    base = os.path.basename(lib)
    path = LOCAL.get(base) if base in LOCAL and os.path.exists(LOCAL[base]) else (os.path.realpath(lib) if os.path.exists(lib) else None)
    if path and not os.environ.get("PROF_LOCAL_ONLY"):      # (the build container and the GPU boxes run the same image: system libraries resolve too)
        t = table(path) or [(0, "*")]
        i = bisect.bisect_right(t, (int(off, 16), "￿")) - 1
        cnt[(thread, base, t[i][1][:70] if i >= 0 else "?")] += 1
    else:
        cnt[(thread, base, "*")] += 1
print(f"{n} samples")
print("by thread name:", ", ".join(f"{t} {100.0 * c / n:.1f} %" for t, c in by_thread.most_common()))
for (thread, lib, fn), c in cnt.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    print(f"  {100.0 * c / n:5.1f} %  {thread:16s} {lib:24s} {fn}")
