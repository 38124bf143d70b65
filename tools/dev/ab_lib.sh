#!/bin/bash
# same-box A/B: tools/diag/lib_ab.so (A) against the in-tree library (B); args: extra bench.py flags
for lib in tools/diag/lib_ab.so torchpiv_amd/libtorchpiv_hip.so tools/diag/lib_ab.so torchpiv_amd/libtorchpiv_hip.so; do
TPIV_LIB=$lib python3 bench.py --no-cpu-baseline --pmc off --steps 60 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(r['value']), {k: round(v,3) for k,v in r['kernel_ms'].items()})"
done
