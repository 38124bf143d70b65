#!/bin/bash
# Bench lines beside the headline: reference precision, configs[2] stream, 2-rank gloo rehearsal of --gpus 2.
set -o pipefail
TAG=${1:-modes}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
show() { python3 - "$1" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "n_gpus", r["n_gpus"], round(r["value"]), r["unit"], r["dtype"], r["scaling"], "step median", round(r["step_ms"]["median"], 3),
      "ms; dominant", r["roofline"]["kernel"], "frac", round(r["roofline"]["frac"], 4), {k: round(v, 3) for k, v in r["kernel_ms"].items()})
PY
}
timeout -k 10 400 python3 bench.py --precision reference --no-cpu-baseline > $OUT/bench_n1_reference.json 2> $OUT/ref.err || { tail $OUT/ref.err; exit 1; }
show $OUT/bench_n1_reference.json
timeout -k 10 400 python3 bench.py --config 2 --no-cpu-baseline > $OUT/bench_n1_config2.json 2> $OUT/c2.err || { tail $OUT/c2.err; exit 1; }
show $OUT/bench_n1_config2.json
TPIV_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 2 --batch 64 --pmc off > $OUT/bench_gloo2_config1.json 2> $OUT/g2.err || { tail -n 30 $OUT/g2.err; exit 1; }
show $OUT/bench_gloo2_config1.json
TPIV_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --config 2 --steps 3 --stream 1000 --batch 250 --pmc off > $OUT/bench_gloo2_config2.json 2> $OUT/g2c2.err || { tail -n 30 $OUT/g2c2.err; exit 1; }
show $OUT/bench_gloo2_config2.json
