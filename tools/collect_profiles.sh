#!/bin/bash
# Run ON THE GPU BOX (through gpurun): produces the per-round evidence under gpurun_out/<tag>/
#   bench.json            python bench.py (default steps)
#   kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command
#   pmc_fetch / pmc_write FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section)
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R && python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq.err
find $OUT -name "*kernel_trace.csv" -delete   # large; the stats summary is what is kept
ls -R $OUT | head -40
