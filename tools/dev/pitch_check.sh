#!/bin/bash
# does a power-of-two row pitch cost anything?  same height, widths around 2048 / 4096
for m in CWS DWS; do
for w in 2048 2064; do
  echo "2048 x $w $m"; python3 tools/quick_bench.py --size 2048 --width $w --ws 64 --passes 2 --mode $m --batch 64 2>&1 | grep -E "us/pair:" | sed -e "s/.*us\/pair/us\/pair/"
done
done
for w in 4096 4112; do
  echo "4096 x $w"; python3 tools/quick_bench.py --size 4096 --width $w --ws 32 --passes 3 --mode CWS --batch 16 2>&1 | grep -E "us/pair:" | sed -e "s/.*us\/pair/us\/pair/"
done
