#!/bin/bash
# L1 / L2 request counts of the shifted 32x32 pass at a power-of-two row pitch against a padded one
export TMPDIR=/tmp
for w in 2048 2064; do
  OUT=gpurun_out/pitch_pmc_$w; rm -rf $OUT; mkdir -p $OUT
  A="--size 2048 --width $w --ws 64 --passes 2 --mode CWS --batch 64 --iters 2"
  timeout -k 10 150 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-include-regex "xcorr_tile" --output-format csv -d $OUT/p1 -- python3 tools/quick_bench.py $A > /dev/null 2> $OUT/p1.err || { tail -3 $OUT/p1.err; exit 1; }
  timeout -k 10 150 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-include-regex "xcorr_tile" --output-format csv -d $OUT/p2 -- python3 tools/quick_bench.py $A > /dev/null 2> $OUT/p2.err || { tail -3 $OUT/p2.err; exit 1; }
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "xcorr_tile" --output-format csv -d $OUT/p3 -- python3 tools/quick_bench.py $A > /dev/null 2> $OUT/p3.err || { tail -3 $OUT/p3.err; exit 1; }
  echo "== width $w"; python3 tools/pmc_summary.py $OUT/p1 $OUT/p2 $OUT/p3
done
