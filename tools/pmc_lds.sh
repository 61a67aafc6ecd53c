#!/bin/bash
# Run ON THE GPU BOX: LDS bank-conflict share per kernel (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) of quick_bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_lds}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc -- python3 $R/tools/quick_bench.py 256 20 noprofile > /dev/null 2> $OUT/pmc.err
python3 $R/tools/pmc_summary.py $OUT/pmc
