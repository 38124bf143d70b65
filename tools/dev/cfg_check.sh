#!/bin/bash
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
python3 tools/quick_bench.py --size 4096 --ws 32 --passes 3 --mode CWS --batch 16 2>&1 | grep -E "pairs/s|us/pair"
python3 bench.py --no-cpu-baseline --pmc off --steps 60 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg1', round(r['value']), {k: round(v,3) for k,v in r['kernel_ms'].items()})"
