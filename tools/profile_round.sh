#!/bin/bash
# Round profile on the GPU box.  Everything the bench line quotes can be re-derived from these outputs:
#   bench_n1.json               the driver's command (configs[1] at the default precision "exact" = the headline, with the
#                               float64-FFT and all-float32 runs beside it, the end_to_end block, live PMC counters and the
#                               CPU baseline)
#   bench_n1_reference.json     the same at precision "reference" (reference-order CWS staging)
#   bench_n1_config2.json       configs[2] stream
#   bench_gloo{2,4}_*.json      2- / 4-rank gloo launch rehearsals of --gpus N (the ranks time-slice ONE GPU: not a measurement)
#   bench_soak2000.json         2000 timed steps of the headline workload
#   rocprofv3_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the headline command
#   other_configs/              configs[3] / configs[4] at the three precisions: quick_bench lines, kernel stats, SQ counters
#   stamps_f64.txt              per-phase cycle shares of the float64 pass-1 kernel (stamped diagnostic build)
#   files_profile.txt           where the file path spends its time
#   bench_e2e_gloo2.json        the GENERATOR path sharded over 2 gloo ranks on the one GPU (bench.py --gpus 2 --e2e): launch,
#                               per-rank host budget and the end-of-run gather from device-resident fields -- a rehearsal
#   generic_chain.txt           64/32 -> 42/21 -> 28/14 (multipass_scale 1.5) with the three generations of the generic kernel
#   other_configs/exact_sizes.txt   precision "exact" against "f64" / "fast" at the first-pass sizes other than 32 / 64 / 128
#                               (quick_bench lines: 8, 16, 24 ... 96 and the chains 16/8 -> 8/4, 48/24 -> 32/16)
#   stamps_cfg1_cws.txt         per-phase cycle shares of the locating pass and the 32x32 CWS kernel (stamped build)
#   exact_adversarial.txt       tools/research/exact_adversarial.py: the float32 map's error against the proven bound
# PART=a: the bench lines and the kernel statistics; PART=b: other configs, stamps, file path, rehearsals (two gpurun calls)
# Outputs: gpurun_out/$1/ ; tools/collect_profile.py copies what is to be judged into profiles/<round>/.
set -o pipefail
TAG=${1:-r05}
OUT=gpurun_out/$TAG
mkdir -p $OUT/other_configs
export TMPDIR=/tmp
R="--kernel-include-regex xcorr|predict|finalize|postval"
PART=${2:-ab}
if [[ $PART == *a* ]]; then
python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { tail $OUT/bench_n1.err; exit 1; }
python3 bench.py --precision reference --no-cpu-baseline --no-fast --no-e2e > $OUT/bench_n1_reference.json 2> $OUT/bench_ref.err || exit 1
python3 bench.py --config 2 --no-cpu-baseline --no-e2e > $OUT/bench_n1_config2.json 2> $OUT/bench_c2.err || exit 1
TPIV_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 20 --warmup 2 --batch 64 --pmc off > $OUT/bench_gloo2_config1.json 2> $OUT/g2.err || exit 1
TPIV_DIST_BACKEND=gloo python3 bench.py --gpus 2 --config 2 --steps 3 --stream 1000 --batch 250 --pmc off > $OUT/bench_gloo2_config2.json 2> $OUT/g2c2.err || exit 1
TPIV_DIST_BACKEND=gloo python3 bench.py --gpus 4 --steps 10 --warmup 2 --batch 32 --pmc off > $OUT/bench_gloo4_config1.json 2> $OUT/g4.err || exit 1
# soak: 2000 timed steps of the headline workload (stability of the rate; no counters, no side measurements)
python3 bench.py --steps 2000 --warmup 5 --pmc off --no-cpu-baseline --no-e2e --no-fast > $OUT/bench_soak2000.json 2> $OUT/soak.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv $R -d $OUT/stats -- python3 bench.py --steps 50 --warmup 5 --pmc off --no-cpu-baseline --no-e2e > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $OUT/stats/*/*_kernel_stats.csv $OUT/rocprofv3_kernel_stats.csv && rm -rf $OUT/stats
fi
if [[ $PART == *b* ]]; then
cfg() {   # name, quick_bench args
    local name=$1; shift
    python3 tools/quick_bench.py "$@" 2>&1 | grep -E "pairs/s|us/pair:|^exact" > $OUT/other_configs/$name.txt
    rocprofv3 --kernel-trace --stats --output-format csv $R -d $OUT/other_configs/$name.prof -- python3 tools/quick_bench.py "$@" > /dev/null 2> $OUT/other_configs/$name.err
    cp $OUT/other_configs/$name.prof/*/*_kernel_stats.csv $OUT/other_configs/${name}_kernel_stats.csv; rm -rf $OUT/other_configs/$name.prof
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv --kernel-include-regex "xcorr|predict" -d $OUT/other_configs/$name.pmc -- python3 tools/quick_bench.py "$@" --iters 2 > /dev/null 2>> $OUT/other_configs/$name.err
    python3 tools/pmc_summary.py $OUT/other_configs/$name.pmc > $OUT/other_configs/${name}_pmc.txt; rm -rf $OUT/other_configs/$name.pmc
    cat $OUT/other_configs/$name.txt
}
cfg cfg3_4096_32_16_8_fast --size 4096 --ws 32 --passes 3 --mode CWS --batch 16
cfg cfg3_4096_32_16_8_f64 --size 4096 --ws 32 --passes 3 --mode CWS --batch 16 --precision f64
cfg cfg3_4096_32_16_8_exact --size 4096 --ws 32 --passes 3 --mode CWS --batch 16 --precision exact
cfg cfg4_128_64_fast --size 2048 --ws 128 --passes 2 --mode CWS --batch 64
cfg cfg4_128_64_f64 --size 2048 --ws 128 --passes 2 --mode CWS --batch 64 --precision f64
cfg cfg4_128_64_exact --size 2048 --ws 128 --passes 2 --mode CWS --batch 64 --precision exact
TPIV_LIB=tools/diag/libtorchpiv_hip_stamps.so python3 tools/stamp_f64.py > $OUT/stamps_f64.txt 2> $OUT/stamps.err || tail -3 $OUT/stamps.err
TPIV_LIB=tools/diag/libtorchpiv_hip_stamps.so python3 tools/stamp_profile.py CWS 2> /dev/null | grep -v amdgpu.ids > $OUT/stamps_cfg1_cws.txt
bash tools/dev/exact_chains.sh $TAG/chains > /dev/null 2>&1; cp $OUT/chains/chains.log $OUT/other_configs/exact_sizes.txt
python3 tools/research/exact_adversarial.py 2> /dev/null | grep -v amdgpu.ids > $OUT/exact_adversarial.txt
python3 tools/dev/files_profile.py 8 8 8 spots 2 32 2> /dev/null | grep -v amdgpu.ids > $OUT/files_profile.txt
TPIV_DIST_BACKEND=gloo python3 bench.py --gpus 2 --e2e --e2e-pairs 64 > $OUT/bench_e2e_gloo2.json 2> $OUT/e2e_g2.err || tail -3 $OUT/e2e_g2.err
python3 bench.py --e2e --e2e-pairs 128 > $OUT/bench_e2e_n1.json 2> $OUT/e2e_n1.err || tail -3 $OUT/e2e_n1.err
# first generation (plain DFTs), second (run-time two-factor form), third (compile-time instances, in-register transforms)
{ TPIV_GENERIC_CT=0 python3 tools/dev/generic_chain.py 16 1.5 CWS; TPIV_GENERIC_REG=0 python3 tools/dev/generic_chain.py 16 1.5 CWS; python3 tools/dev/generic_chain.py 16 1.5 CWS; python3 tools/dev/generic_chain.py 16 1.5 DWS; } 2> /dev/null | grep -v amdgpu.ids > $OUT/generic_chain.txt
fi
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "n_gpus", r.get("n_gpus"), round(r["value"]), r["unit"], r.get("dtype"), (r.get("roofline") or {}).get("kernel"), round((r.get("roofline") or {}).get("frac", 0), 4),
              "fast", round((r.get("fast") or {}).get("value", 0)))
    except Exception as e:
        print(f, "unreadable", e)
PY
