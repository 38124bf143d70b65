# fallback share and pass-1 time of precision="exact" against the width of the decision band: TPIV_EXACT_BAND_SCALE multiplies the
# proven band 2 Gamma (1 + 1/16) E+ (1 = the shipped band; < 1 is NOT safe), TPIV_EXACT_BAND_RANGE adds a floor relative to the map's range
mkdir -p gpurun_out
L=gpurun_out/r4_band2.log
timeout -k 10 300 python tools/research/exact_band.py > $L 2>&1 || exit 1
for band in 0.5 1 2 4; do
  for noise in 0 8; do
    echo "== band $band noise $noise" >> $L
    TPIV_EXACT_BAND_SCALE=$band timeout -k 10 300 python tools/quick_bench.py --size 2048 --ws 64 --passes 2 --mode CWS --batch 256 --distinct 16 --noise $noise --precision exact >> $L 2>&1 || exit 1
  done
done
grep -v amdgpu.ids $L | grep "==\|exact:\|per-kernel\|pairs/s\|family\|e-0\|e+0"
