#!/bin/bash
# same-box A/B of tools/diag/lib_ab.so (A) against the in-tree library (B): configs[1] bench line + configs[3]
for lib in tools/diag/lib_ab.so torchpiv_amd/libtorchpiv_hip.so tools/diag/lib_ab.so torchpiv_amd/libtorchpiv_hip.so; do
TPIV_LIB=$lib python3 bench.py --no-cpu-baseline --pmc off --steps 60 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(r['value']), {k: round(v,3) for k,v in r['kernel_ms'].items()})"
TPIV_LIB=$lib python3 tools/quick_bench.py --size 4096 --ws 32 --passes 3 --mode CWS --batch 16 2>&1 | grep -E "us/pair:" | sed -e "s/.*us\/pair/   cfg3 us\/pair/"
done
