#!/bin/bash
# One GPU-box call: the -m gpu suite, then the bench lines of this round.  Outputs under gpurun_out/$1/.
set -o pipefail
TAG=${1:-chk}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1
rc=$?
tail -n 15 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { tail -n 20 $OUT/bench_n1.err; exit 1; }
python3 - $OUT/bench_n1.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("config1 fast:", round(r["value"]), "pairs/s", "step median", round(r["step_ms"]["median"], 3), "ms",
      "frac", round(r["roofline"]["frac"], 4), "kernel", r["roofline"]["kernel"], "traffic", r["roofline"]["traffic"],
      "counters:", r["roofline"]["counters"][:80])
print({k: round(v, 3) for k, v in r["kernel_ms"].items()})
print("cpu:", r.get("cpu_baseline", {}).get("value"), r.get("cpu_baseline", {}).get("end_to_end"))
PY
