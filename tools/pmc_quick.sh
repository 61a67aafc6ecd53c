#!/bin/bash
# Run ON THE GPU BOX: kernel durations (rocprofv3 --kernel-trace --stats) and SQ counters of tools/quick_bench.py.
# usage: pmc_quick.sh <tag> [stack|frames] [batch]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_quick}
export QB_INPUT=${2:-stack}
B=${3:-256}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/quick_bench.py $B 100 noprofile > $OUT/qb.txt 2> $OUT/trace.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_a -- python3 $R/tools/quick_bench.py $B 10 noprofile > /dev/null 2> $OUT/pmc_a.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_b -- python3 $R/tools/quick_bench.py $B 10 noprofile > /dev/null 2> $OUT/pmc_b.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    n = r["Name"][:70]
    if "mfma" in n or "bboxcc" in n:
        print(f'{n:72s} {float(r["AverageNs"])/1e3:8.1f} us  x{r["Calls"]}')
        tot += float(r["AverageNs"]) / 1e3
print("sum", round(tot, 1))
P
find $OUT -name "*kernel_trace.csv" -delete
python3 $R/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
cat $OUT/qb.txt | head -2
