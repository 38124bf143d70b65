#!/bin/bash
export TMPDIR=/tmp
for lib in torchpiv_amd/libtorchpiv_hip.so tools/diag/lib_f64s1.so tools/diag/lib_f64s2.so tools/diag/lib_f64s4.so tools/diag/lib_f64s7.so; do
  rm -rf gpurun_out/pmc_dev; mkdir -p gpurun_out/pmc_dev
  TPIV_LIB=$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU --kernel-include-regex "xcorr_f64" --output-format csv -d gpurun_out/pmc_dev/a -- python3 bench.py --pmc-child --pmc off --no-cpu-baseline --steps 2 --warmup 1 --distinct 8 --precision reference > /dev/null 2> gpurun_out/pmc_dev/err.txt
  echo "== $lib"; python3 tools/pmc_summary.py gpurun_out/pmc_dev/a | grep -E "BANK|ACTIVE|INSTS"
  TPIV_LIB=$lib python3 bench.py --precision reference --no-cpu-baseline --pmc off --steps 20 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   pass1 ms', round(r['kernel_ms']['pass1_xcorr'],2))"
done
