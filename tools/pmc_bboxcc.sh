#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the bboxcc kernels on the batch sweep's sparse-blob masks (B = 65,536).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_bboxcc}
CAP=${2:-0}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SWEEP_CAP=$CAP SWEEP_KINDS=${3:-blobs} SWEEP_BATCHES=65536 SWEEP_OUT=$OUT/sweep.json
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_a -- python3 $R/tools/bboxcc_sweep.py > /dev/null 2> $OUT/pmc_a.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_b -- python3 $R/tools/bboxcc_sweep.py > /dev/null 2> $OUT/pmc_b.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/bboxcc_sweep.py > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/bboxcc_sweep.py > /dev/null 2> $OUT/pmc_write.err
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
