#!/bin/bash
set -o pipefail
python3 -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "reference" 2>&1 | tail -3 || exit 1
python3 bench.py --precision reference --no-cpu-baseline --pmc off --steps 30 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reference mode:', round(r['value']), 'pairs/s', r['kernel_ms'], r['roofline']['frac'])"
