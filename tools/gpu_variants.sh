#!/bin/bash
# A/B builds on the GPU box: bench c3 (or GVI_BENCH_ARGS) under every library of VARIANTS ("default" = the in-tree one;
# others = build/variants/libgvi_hip_<name>.so from tools/build_variant.py).  The "timing" build prints its phase stamps.
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
R="${R:-r04}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT="gpurun_out/$R"; mkdir -p "$OUT"
for v in ${VARIANTS:-default}; do
  lib=""; [ "$v" != "default" ] && lib="$ROOT/build/variants/libgvi_hip_$v.so"
  extra=""; [ "$v" = "timing" ] && extra="GVI_FUSED_DBG=8"
  env GVI_LIB_PATH="$lib" $extra ${GVI_ENV:-} timeout -k 10 200 python bench.py --steps ${STEPS:-300} --warmup 30 --no-cpu-baseline ${GVI_BENCH_ARGS:-} > "$OUT/var_$v.json" 2> "$OUT/var_$v.err" || { tail -20 "$OUT/var_$v.err"; exit 1; }
  python - <<PY
import json
d=json.load(open("$OUT/var_$v.json"))
print("[$v] ms/step", round(d["ms_per_step"], 5), "dominant kernel ms", round(d["moments_kernel"]["ms"], 5), "final", d["final_cost"], "breakdown", d.get("iteration_breakdown_us"))
PY
  [ "$v" = "timing" ] && grep "stamps" "$OUT/var_$v.err" | head -80
done
exit 0
