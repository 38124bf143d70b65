#!/bin/bash
# counters of the predictor kernels under configs[3]; every pass under its own timeout
export TMPDIR=/tmp
A="--size 4096 --ws 32 --passes 3 --mode CWS --batch 16 --iters 2"
OUT=gpurun_out/pmc_pred; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-include-regex "predict" --output-format csv -d $OUT/p$i -- python3 tools/quick_bench.py $A > /dev/null 2> $OUT/p$i.err || { echo "pass $i failed"; tail -3 $OUT/p$i.err; exit 1; }
  echo "pass $i done"
done
python3 tools/pmc_summary.py $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 > $OUT/summary.txt; cat $OUT/summary.txt
