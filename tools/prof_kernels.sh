#!/bin/bash
# Run ON THE GPU BOX: per-kernel average durations (rocprofv3 --kernel-trace --stats) of quick_bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/quick_bench.py 256 100 noprofile > $OUT/qb.txt 2> $OUT/trace.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    n = r["Name"][:60]
    if "mfma" in n or "bboxcc" in n:
        print(f'{n:62s} {float(r["AverageNs"])/1e3:8.1f} us  x{r["Calls"]}')
        tot += float(r["AverageNs"]) / 1e3
print("sum", round(tot, 1))
P
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/qb.txt | head -2
