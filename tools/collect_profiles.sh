#!/bin/bash
# Run ON THE GPU BOX (through gpurun): produces the per-round evidence under gpurun_out/<tag>/
#   bench.json                 python bench.py (default steps, every leg; timed entry = carrier frames, two lanes)
#   <entry>[_lanesN]_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the timed workload only (bench.py --no-extra-legs):
#                              entry = frames (the default, what `value` is) with two lanes (kernels of two steps share the
#                              chip: in-situ durations) and with one lane (each launch alone), and the stacked entry
#   <entry>_pmc_fetch / _pmc_write   FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section), one lane
#   <entry>_pmc_sq             SQ counters of the same command
set -u
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R && python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
for SPEC in "frames 3" "frames 1" "stack 1"; do
  read ENTRY LANES <<< "$SPEC"
  NAME=${ENTRY}_lanes${LANES}
  ARGS="--no-cpu-baseline --no-extra-legs --entry $ENTRY --lanes $LANES"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${NAME}_trace -- python3 $R/bench.py --steps 100 --warmup 10 $ARGS > $OUT/${NAME}_bench_under_rocprof.json 2> $OUT/${NAME}_trace.err
  f=$(find $OUT/${NAME}_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${NAME}_kernel_stats.csv
  [ "$LANES" != 1 ] && continue
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${ENTRY}_pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 5 $ARGS > /dev/null 2> $OUT/${ENTRY}_pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${ENTRY}_pmc_write -- python3 $R/bench.py --steps 20 --warmup 5 $ARGS > /dev/null 2> $OUT/${ENTRY}_pmc_write.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/${ENTRY}_pmc_sq -- python3 $R/bench.py --steps 20 --warmup 5 $ARGS > /dev/null 2> $OUT/${ENTRY}_pmc_sq.err
  python3 $R/tools/pmc_summary.py $OUT/${ENTRY}_pmc_sq > $OUT/${ENTRY}_pmc_sq.txt
done
find $OUT -name "*kernel_trace.csv" -delete   # large; the stats summary is what is kept
ls $OUT | head -40
