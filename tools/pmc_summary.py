"""Summarise a rocprofv3 --pmc counter_collection.csv: mean per-dispatch value per kernel and counter."""
import csv
import glob
import sys
import re
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0][:40]
        rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(rows):
    print(k)
    for c in sorted(rows[k]):
        v = rows[k][c]
        print(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
