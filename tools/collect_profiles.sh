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
# round 6: bboxcc over objects per frame at B = 65,536, the LDS bank-conflict share per kernel, the pinned
# pipeline's host cost over lanes x slots and its kernel + memory-copy time line (C driver)
SWEEP_KINDS=blobs,obj6,obj20,obj50,obj100 SWEEP_BATCHES=65536 SWEEP_OUT=$OUT/bboxcc_objects_B65536.json python3 $R/tools/bboxcc_sweep.py > $OUT/bboxcc_objects_B65536.txt 2>&1
PHC_CFGS="3 3;3 6" bash $R/tools/pipe_host_cost.sh cur > $OUT/pipe_host_cost.txt 2>&1
bash $R/tools/pipe_trace2.sh $OUT/pipe_trace c_driver_3x6_blob blob 3 6 > $OUT/pipe_trace.log 2>&1
python3 $R/tools/pipe_timeline.py $OUT/pipe_trace c_driver_3x6_blob 600 > $OUT/pipe_timeline_c_driver_3x6_blob.txt 2>&1
find $OUT -name "*kernel_trace.csv" -not -path "*pipe_trace*" -delete   # large; the stats summary is what is kept
ls $OUT | head -40
