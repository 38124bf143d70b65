#!/bin/bash
# same-box A/B of the float64 pass-1 kernels: tools/diag/lib_ab.so (A) against the in-tree library (B);
# configs[1] (64x64, bench.py) and configs[4] (128x128, quick_bench.py)
for lib in tools/diag/lib_ab.so torchpiv_amd/libtorchpiv_hip.so tools/diag/lib_ab.so torchpiv_amd/libtorchpiv_hip.so; do
TPIV_LIB=$lib python3 bench.py --no-cpu-baseline --pmc off --no-fast --no-e2e --steps 40 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(r['value']), {k: round(v,3) for k,v in r['kernel_ms'].items()})"
TPIV_LIB=$lib python3 tools/quick_bench.py --size 2048 --ws 128 --passes 2 --mode CWS --batch 64 --precision f64 2>&1 | grep -E "us/pair:" | sed -e "s/.*us\/pair/   cfg4 f64 us\/pair/"
done
