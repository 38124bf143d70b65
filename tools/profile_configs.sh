#!/bin/bash
# rocprofv3 kernel statistics + quick_bench lines for the BASELINE.json configs other than the headline
# one (configs[2] DWS, configs[3] 4096^2 32->16->8, configs[4] 128->64).  Outputs: gpurun_out/$1/.
set -o pipefail
TAG=${1:-cfgs}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R="--kernel-include-regex xcorr|predict|finalize"
run() {   # name, quick_bench args
    local name=$1; shift
    python3 tools/quick_bench.py "$@" 2>&1 | grep -E "pairs/s|us/pair:" > $OUT/$name.txt
    rocprofv3 --kernel-trace --stats --output-format csv $R -d $OUT/$name.prof -- python3 tools/quick_bench.py "$@" > /dev/null 2> $OUT/$name.err
    cp $OUT/$name.prof/*/*_kernel_stats.csv $OUT/${name}_kernel_stats.csv
    rm -rf $OUT/$name.prof
    cat $OUT/$name.txt; cut -c1-110 $OUT/${name}_kernel_stats.csv
}
run cfg2_dws --size 2048 --ws 64 --passes 2 --mode DWS --batch 256
run cfg3_4096_32_16_8 --size 4096 --ws 32 --passes 3 --mode CWS --batch 16
run cfg4_128_64 --size 2048 --ws 128 --passes 2 --mode CWS --batch 64
