#!/bin/bash
# bench.py for a list of configs (arguments); JSON lines under gpurun_out/$R
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
R="${R:-r04}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT="gpurun_out/$R"
mkdir -p "$OUT"
for c in "$@"; do
  timeout -k 10 400 python bench.py --config "$c" ${GVI_BENCH_ARGS:-} > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err" || { echo "== $c FAILED"; tail -15 "$OUT/bench_$c.err"; continue; }
  python - <<PY
import json
d=json.load(open("$OUT/bench_$c.json"))
print("== $c ms/step", round(d["ms_per_step"], 5), "value %.3e" % d["value"], "kernel", d["roofline"]["kernel"][:60], "ms", round(d["roofline"]["kernel_ms"], 5),
      "frac", d["roofline"]["frac"], "stages", d.get("iteration_breakdown_us"), "accepted", d["accepted_steps"], "/", d["steps"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
done
