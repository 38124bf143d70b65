#!/bin/bash
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
for w8 in 1 0; do
TPIV_W8=$w8 python3 tools/quick_bench.py --size 4096 --ws 32 --passes 3 --mode CWS --batch 16 2>&1 | grep -E "pairs/s|us/pair" | sed "s/^/W8=$w8 /"
done
