#!/bin/bash
# Round-end validation: full GPU suite, smoke(), default bench (+ cpu baseline), C2 / planar1k / C5 benches, kernel traces,
# 2-rank rehearsal.  Output under gpurun_out/$R/final (R defaults to r03).
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
R="${R:-r04}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O="gpurun_out/$R/final"
mkdir -p "$O"
if [ "${GVI_SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$O/pytest_gpu.log" 2>&1; rc=$?
  tail -4 "$O/pytest_gpu.log"
  [ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1 || { tail -20 "$O/smoke.log"; exit 1; }
tail -1 "$O/smoke.log"
timeout -k 10 300 python bench.py > "$O/bench.json" 2> "$O/bench.err" || { tail -20 "$O/bench.err"; exit 1; }
for c in c2 planar1k; do
  timeout -k 10 300 python bench.py --config $c --steps 500 --warmup 50 > "$O/bench_$c.json" 2> "$O/bench_$c.err" || { tail -20 "$O/bench_$c.err"; exit 1; }
done
timeout -k 10 500 python bench.py --config c5 --no-cpu-baseline > "$O/bench_c5.json" 2> "$O/bench_c5.err" || { tail -20 "$O/bench_c5.err"; exit 1; }
python - <<PY
import json
for c in ("", "_c2", "_planar1k", "_c5"):
    d=json.load(open("$O/bench%s.json" % c))
    print(c or "c3", "ms/step", round(d["ms_per_step"], 5), "value %.4g" % d["value"], "final", d["final_cost"], "kernel ms", round(d["roofline"]["kernel_ms"], 4),
          "frac", round(d["roofline"]["frac"], 3) if d["roofline"]["frac"] == d["roofline"]["frac"] else None, "stages", d.get("iteration_breakdown_us"),
          "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
for c in c3 c2 planar1k; do
  rm -rf "$O/trace_$c"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_$c" -- python3 bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline > "$O/trace_$c.log" 2>&1 || { tail -20 "$O/trace_$c.log"; exit 1; }
  find "$O/trace_$c" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/kernel_stats_$c.csv"
  rm -rf "$O/trace_$c"
done
rm -rf "$O/trace_c5"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_c5" -- python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > "$O/trace_c5.log" 2>&1 || tail -5 "$O/trace_c5.log"
find "$O/trace_c5" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/kernel_stats_c5.csv"
rm -rf "$O/trace_c5"
head -8 "$O/kernel_stats_c3.csv" | cut -c1-150
GVI_BENCH_REHEARSAL=1 GVI_BENCH_C5_CONFIG=c5small timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 60 --warmup 10 > "$O/bench_rehearsal2.log" 2>&1 || { tail -20 "$O/bench_rehearsal2.log"; exit 1; }
grep '^{' "$O/bench_rehearsal2.log" | tail -1 > "$O/bench_rehearsal2.json"
python - <<PY
import json
d=json.load(open("$O/bench_rehearsal2.json"))
print("rehearsal n=2 final", d["final_cost"], "ms/step", d["ms_per_step"], "c5_strong", d["c5_strong"]["ms_per_step"], d["c5_strong"]["accepted_steps"])
PY
