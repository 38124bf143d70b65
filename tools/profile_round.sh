#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel stats of the same command, and the
# HBM-traffic counters (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, as MI355X_MICROARCH.md
# prescribes).  Outputs go to gpurun_out/$1/ ; copy the summaries into profiles/ afterwards.
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R="--kernel-include-regex xcorr|predict"
BENCH="python3 bench.py --steps 6 --warmup 2"
python3 bench.py --steps 8 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv $R -d $OUT/stats -- $BENCH --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv $R -d $OUT/pmc_fetch -- $BENCH --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv $R -d $OUT/pmc_write -- $BENCH --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv $R -d $OUT/pmc_insts -- $BENCH --no-cpu-baseline > /dev/null 2> $OUT/pmc_insts.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv $R -d $OUT/pmc_l2 -- $BENCH --no-cpu-baseline > /dev/null 2> $OUT/pmc_l2.err
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_insts $OUT/pmc_l2 > $OUT/pmc_summary.txt
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_insts $OUT/pmc_l2
grep -E "xcorr|predict|Name" $OUT/kernel_stats.csv | cut -c1-150
cat $OUT/pmc_summary.txt
tail -c 2500 $OUT/bench.json
