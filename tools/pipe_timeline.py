"""Developer helper (here): steady-state summary + excerpt of a rocprofv3 kernel / memory-copy trace of the pinned pipeline
(tools/pipe_trace.sh, tools/pipe_trace2.sh).  usage: pipe_timeline.py <dir> <tag> [excerpt_us]"""
import collections
import csv
import sys

d, tag = sys.argv[1], sys.argv[2]
span = float(sys.argv[3]) if len(sys.argv) > 3 else 400.0


def short(n):
    for k in ("enc0p", "enc1_mfma", "enc23", "dec012", "dec3cc", "pack_kernel", "scan_kernel", "fillBuffer", "copyBuffer", "bboxcc"):
        if k in n:
            return k
    return n[:20]


K = list(csv.DictReader(open(f"{d}/{tag}_kernel_trace.csv")))
C = list(csv.DictReader(open(f"{d}/{tag}_memory_copy_trace.csv")))
ev = []
for r in K:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), "q" + r["Queue_Id"] + "s" + r["Stream_Id"]))
for r in C:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "H2D" if "HOST_TO" in r["Direction"] else "D2H", "s" + r["Stream_Id"]))
ev.sort()
packs = [e for e in ev if e[2] == "pack_kernel"]
pk = packs[len(packs) // 2:]
print(tag, "batches", len(packs), "steady-state period per batch (us):", round((pk[-1][1] - pk[0][1]) / (len(pk) - 1) / 1e3, 1))
t0 = pk[0][0]
dur = collections.defaultdict(list)
for e in ev:
    if e[0] >= t0:
        dur[e[2]].append((e[1] - e[0]) / 1e3)
for k, v in dur.items():
    print(f"   {k:12s} n={len(v):3d} avg {sum(v)/len(v):6.1f} us  max {max(v):6.1f}")
# busy fraction: union of kernel intervals
iv = sorted((e[0], e[1]) for e in ev if e[0] >= t0 and e[2] not in ("H2D", "D2H"))
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("   some kernel running:", round(100 * busy / (iv[-1][1] - iv[0][0]), 1), "% of the steady-state span")
base = pk[len(pk) // 2][0]
for e in ev:
    if base - span / 2 * 1e3 <= e[0] <= base + span / 2 * 1e3:
        print(f"   {(e[0]-base)/1e3:8.1f} {(e[1]-base)/1e3:8.1f} {e[2]:12s} {e[3]}")
