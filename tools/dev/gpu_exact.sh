# GPU check of precision="exact": its test file, then the quick bench of configs[1] / [3] / [4] geometries
mkdir -p gpurun_out
L=gpurun_out/r4_exact2.log
timeout -k 10 600 python -m pytest tests/test_gpu_exact.py -x -q > $L 2>&1; rc=$?; echo "pytest rc=$rc" >> $L
tail -30 $L
[ $rc -eq 0 ] || exit $rc
for prec in exact f64 fast; do
  timeout -k 10 300 python tools/quick_bench.py --size 2048 --ws 64 --passes 2 --mode CWS --batch 256 --distinct 8 --precision $prec >> $L 2>&1 || exit 1
  timeout -k 10 300 python tools/quick_bench.py --size 4096 --ws 32 --passes 3 --mode CWS --batch 16 --distinct 4 --precision $prec >> $L 2>&1 || exit 1
  timeout -k 10 300 python tools/quick_bench.py --size 2048 --ws 128 --passes 2 --mode CWS --batch 64 --distinct 8 --precision $prec >> $L 2>&1 || exit 1
done
grep "^size\|per-kernel\|^exact" $L | cut -c1-230
