#!/bin/bash
# bench.py repeatability: "<config> <steps> <warmup>" triples (arguments, quoted), twice each
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
mkdir -p gpurun_out/r03/var
for rep in 1 2; do
  for csw in "$@"; do
    set -- $csw
    timeout -k 10 200 python bench.py --config $1 --steps $2 --warmup $3 --no-cpu-baseline ${GVI_BENCH_ARGS:-} > gpurun_out/r03/var/b_$1_$2_$3_$rep.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("gpurun_out/r03/var/b_$1_$2_$3_$rep.json"))
print("$1 steps $2 warmup $3 rep $rep: ms/step %.5f  ref-order %.5f  trials %.2f passes %d/%d stages %s" % (d["ms_per_step"], d["reference_pass_order"]["ms_per_step"], d["trials_per_step"], d["passes"]["full"], d["passes"]["cost_only"], {k: v["mean_us"] for k, v in d["iteration_breakdown_us"].items()}))
PY
  done
done
