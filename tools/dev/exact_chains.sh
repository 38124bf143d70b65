# precision "exact" against "f64" and "fast" at first-pass sizes other than 32 / 64 / 128: quick_bench
# lines per size and the two chains the verdict names.  Output: gpurun_out/$1/chains.log
OUT=gpurun_out/${1:-chains}
mkdir -p $OUT
L=$OUT/chains.log
: > $L
qb() { echo "== $*" >> $L; timeout -k 10 200 python3 tools/quick_bench.py "$@" 2>&1 | grep -v amdgpu.ids >> $L || exit 1; }
for prec in exact f64 fast; do
  qb --size 4096 --ws 16 --passes 2 --mode CWS --batch 8 --distinct 2 --precision $prec        # 16/8 -> 8/4
  qb --size 2048 --ws 48 --passes 2 --mode CWS --scale 1.5 --batch 32 --distinct 4 --precision $prec   # 48/24 -> 32/16
done
for ws in 8 16 24 28 40 42 48 56 96; do
  sz=2048; b=16
  [ $ws -le 16 ] && { sz=4096; b=4; }
  for prec in exact f64; do
    qb --size $sz --ws $ws --passes 1 --batch $b --distinct 2 --precision $prec
  done
done
# 8/4 on frames with sensor noise (the default synthetic frames have a noise-free background: 8 % of their sparse 8x8 windows have flat maps)
for prec in exact f64; do qb --size 4096 --ws 8 --passes 1 --batch 4 --distinct 2 --noise 2 --precision $prec; done
grep "==\|pairs/s\|exact" $L
