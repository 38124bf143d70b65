#!/bin/bash
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv --kernel-include-regex "predict|finalize" -d gpurun_out/predprof -- python3 tools/quick_bench.py --size 4096 --ws 32 --passes 3 --mode CWS --batch 16 > /dev/null 2> gpurun_out/predprof.err
cat gpurun_out/predprof/*/*_kernel_stats.csv | cut -c1-200
