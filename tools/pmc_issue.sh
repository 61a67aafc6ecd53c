#!/bin/bash
# Run ON THE GPU BOX: where the waves' cycles go, per kernel of tools/quick_bench.py (two counter passes):
# active-instruction cycles by class, waits by class, MFMA busy / co-execution, LDS and TA FIFO-full cycles.
# usage: pmc_issue.sh <tag> [stack|frames] [batch]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_issue}
export QB_INPUT=${2:-frames}
B=${3:-256}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_a -- python3 $R/tools/quick_bench.py $B 10 noprofile > /dev/null 2> $OUT/pmc_a.err
rocprofv3 --pmc SQ_INST_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_b -- python3 $R/tools/quick_bench.py $B 10 noprofile > /dev/null 2> $OUT/pmc_b.err
python3 $R/tools/pmc_summary.py $OUT > $OUT/pmc_issue.txt
cat $OUT/pmc_issue.txt
