#!/bin/bash
set -o pipefail
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 || exit 1
for prec in fast reference; do
python3 bench.py --precision $prec --no-cpu-baseline --steps 50 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$prec', round(r['value']), 'pairs/s', {k: round(v,3) for k,v in r['kernel_ms'].items()}, 'frac', round(r['roofline']['frac'],4), [ (k, round(v['valu_issue']['insts_per_launch']/1e9,3)) for k,v in r['kernels'].items() if v.get('valu_issue')])"
done
