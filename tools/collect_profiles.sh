#!/bin/bash
# Run ON THE GPU BOX (through gpurun): produces the per-round evidence under gpurun_out/<tag>/
#   bench.json            python bench.py (default steps, every leg)
#   kernel_stats.csv      rocprofv3 --kernel-trace --stats of the timed workload only (bench.py --no-extra-legs)
#   pmc_fetch / pmc_write FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section)
#   pmc_sq                SQ counters of the same command
#   frames_*              the same for the carrier-frame entry point (tools/quick_bench.py, QB_INPUT=frames)
set -u
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R && python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extra-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 100 --warmup 10 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 5 $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 20 --warmup 5 $ARGS > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 20 --warmup 5 $ARGS > /dev/null 2> $OUT/pmc_sq.err
export QB_INPUT=frames
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/frames_trace -- python3 $R/tools/quick_bench.py 256 100 noprofile > $OUT/frames_qb.txt 2> $OUT/frames_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/frames_pmc_fetch -- python3 $R/tools/quick_bench.py 256 20 noprofile > /dev/null 2> $OUT/frames_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/frames_pmc_write -- python3 $R/tools/quick_bench.py 256 20 noprofile > /dev/null 2> $OUT/frames_pmc_write.err
unset QB_INPUT
find $OUT -name "*kernel_trace.csv" -delete   # large; the stats summary is what is kept
python3 $R/tools/pmc_summary.py $OUT/pmc_sq > $OUT/pmc_sq.txt
for d in trace frames_trace; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv; done
ls $OUT | head -40
