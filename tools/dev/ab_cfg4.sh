#!/bin/bash
for lib in tools/diag/lib_ab.so torchpiv_amd/libtorchpiv_hip.so tools/diag/lib_ab.so torchpiv_amd/libtorchpiv_hip.so; do
echo $lib; TPIV_LIB=$lib python3 tools/quick_bench.py --size 2048 --ws 128 --passes 2 --mode CWS --batch 64 2>&1 | grep -E "us/pair:" | sed -e "s/.*us\/pair/   cfg4 us\/pair/"
done
