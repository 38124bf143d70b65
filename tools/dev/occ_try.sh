#!/bin/bash
for occ in 0 4 2; do
TPIV_LIB=tools/diag/lib_exp.so TPIV_OCC=$occ python3 bench.py --no-cpu-baseline --pmc off --steps 60 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('occ $occ', round(r['value']), {k: round(v,3) for k,v in r['kernel_ms'].items()})"
done
