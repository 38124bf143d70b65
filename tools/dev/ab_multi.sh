#!/bin/bash
# same-box comparison of several library builds: configs[1] bench line
for lib in "$@"; do
TPIV_LIB=$lib python3 bench.py --no-cpu-baseline --pmc off --steps 60 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(r['value']), {k: round(v,3) for k,v in r['kernel_ms'].items()})"
done
