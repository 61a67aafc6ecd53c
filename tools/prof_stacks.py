"""Developer helper: call chains of the chain bench's sampling profiler (CHAINBENCH_PROF=1 CHAINBENCH_PROF_STACKS=1
CHAINBENCH_PROF_RAW=<file>, gst/gst_element_driver.c).  Per sample the raw file holds "lib offset thread | lib offset symbol | ..."
(innermost first).  Prints the share of samples per thread name and, per thread name, the most frequent chains.
usage: prof_stacks.py <raw> [chains per thread] [frames per chain]"""
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
        for line in out.splitlines():
            parts = line.split(None, 2)
            if len(parts) == 3 and parts[1] in "tTwWiu":
                syms.append((int(parts[0], 16), parts[2]))
        tables[path] = syms
    return tables[path]


def name(lib, off, sym):
    base = os.path.basename(lib)
    if base in LOCAL and os.path.exists(LOCAL[base]):
        t = table(LOCAL[base])
        i = bisect.bisect_right([a for a, _ in t], int(off, 16)) - 1
        if i >= 0:
            return t[i][1].split("(")[0][-40:]
    return (sym if sym != "-" else base)[:40]


nchains = int(sys.argv[2]) if len(sys.argv) > 2 else 8
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 7
per_thread = collections.defaultdict(collections.Counter)
tot = 0
for line in open(sys.argv[1]):
    parts = line.rstrip("\n").split(" | ")
    head = parts[0].split()
    thread = head[2] if len(head) > 2 else "?"
    frames = []
    for fr in parts[1:]:
        f = fr.split()
        if len(f) >= 2:
            frames.append(name(f[0], f[1], f[2] if len(f) > 2 else "-"))
    # the first frames are the signal trampoline / handler: drop up to the first frame that is not ours
    while frames and ("prof_handler" in frames[0] or "restore_rt" in frames[0] or "libc.so" in frames[0] and len(frames) > depth):
        frames.pop(0)
    per_thread[thread][" < ".join(frames[:depth])] += 1
    tot += 1
print(f"{tot} samples")
for thread, c in sorted(per_thread.items(), key=lambda kv: -sum(kv[1].values())):
    n = sum(c.values())
    print(f"== {thread}: {100.0 * n / tot:.1f} % of the samples")
    for chain, k in c.most_common(nchains):
        print(f"   {100.0 * k / tot:5.1f} %  {chain}")
