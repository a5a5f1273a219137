#!/bin/bash
# Iteration loop on the GPU box: GPU parity tests (GVI_TEST_K selects a subset, GVI_TEST_FILES the files), then the c3
# bench under each environment of GVI_BENCH_ENVS ("A=1 B=2|C=3|" -- an empty entry is the default environment), then a
# kernel trace of the default.  Output under gpurun_out/$R (R defaults to r03).
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
R="${R:-r04}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT="gpurun_out/$R"
mkdir -p "$OUT"
if [ "${GVI_SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 1000 python -m pytest ${GVI_TEST_FILES:-tests} -m gpu -x -q ${GVI_TEST_K:+-k "$GVI_TEST_K"} > "$OUT/pytest_iter.log" 2>&1; rc=$?
  tail -5 "$OUT/pytest_iter.log"
  [ $rc -ne 0 ] && exit $rc
fi
IFS='|' read -ra ENVS <<< "${GVI_BENCH_ENVS:-|}"
[ ${#ENVS[@]} -eq 0 ] && ENVS=("")
i=0
for e in "${ENVS[@]}"; do
  env $e timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline ${GVI_BENCH_ARGS:-} > "$OUT/bench_iter_$i.json" 2> "$OUT/bench_iter.err" || { tail -20 "$OUT/bench_iter.err"; exit 1; }
  python - <<PY
import json
d=json.load(open("$OUT/bench_iter_$i.json"))
print("[$e] ms/step", round(d["ms_per_step"], 5), "psi kernel ms", round(d["moments_kernel"]["ms"], 5), "final", d["final_cost"], "accepted", d["accepted_steps"], "passes", d["passes"]["full"], d["passes"]["cost_only"])
PY
  i=$((i+1))
done
if [ "${GVI_SKIP_TRACE:-0}" != "1" ]; then
  rm -rf "$OUT/trace"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline ${GVI_BENCH_ARGS:-} > "$OUT/trace.log" 2>&1 || { tail -20 "$OUT/trace.log"; exit 1; }
  find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_iter.csv"
  head -14 "$OUT/kernel_stats_iter.csv" | cut -c1-150
fi
