#!/bin/bash
# PMC counters of kernels matching $1 under tools/quick_bench.py $2...
export TMPDIR=/tmp
REGEX=$1; shift
OUT=gpurun_out/pmc_dev; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
  --kernel-include-regex "$REGEX" --output-format csv -d $OUT/a -- python3 tools/quick_bench.py "$@" --iters 2 > /dev/null 2> $OUT/a.err || { tail $OUT/a.err; exit 1; }
rocprofv3 --pmc SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR \
  --kernel-include-regex "$REGEX" --output-format csv -d $OUT/b -- python3 tools/quick_bench.py "$@" --iters 2 > /dev/null 2> $OUT/b.err || { tail $OUT/b.err; exit 1; }
python3 tools/pmc_summary.py $OUT/a $OUT/b
