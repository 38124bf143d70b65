# GPU check of precision="exact": its test file, then the quick bench of configs[1] at the three precisions
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_exact.py -x -q > gpurun_out/r4_exact1.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r4_exact1.log
tail -30 gpurun_out/r4_exact1.log
[ $rc -eq 0 ] || exit $rc
for prec in exact f64 fast; do
  timeout -k 10 300 python tools/quick_bench.py --size 2048 --ws 64 --passes 2 --mode CWS --batch 256 --distinct 8 --precision $prec >> gpurun_out/r4_exact1.log 2>&1 || exit 1
done
tail -12 gpurun_out/r4_exact1.log
