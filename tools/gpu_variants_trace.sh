#!/bin/bash
# rocprofv3 kernel trace of bench.py under every library of VARIANTS: average duration of the kernels matching KERNEL_RE
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
R="${R:-r04}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT="gpurun_out/$R"; mkdir -p "$OUT"
for v in ${VARIANTS:-default}; do
  lib=""; [ "$v" != "default" ] && lib="$ROOT/build/variants/libgvi_hip_$v.so"
  rm -rf "$OUT/trace_$v"
  GVI_LIB_PATH="$lib" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$v" -- python3 bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline ${GVI_BENCH_ARGS:-} > "$OUT/trace_$v.log" 2>&1 || { tail -20 "$OUT/trace_$v.log"; exit 1; }
  f=$(find "$OUT/trace_$v" -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "${KERNEL_RE:-fused|chain}" "$f" | cut -d, -f1-4 | cut -c1-170
  cp "$f" "$OUT/kernel_stats_$v.csv"; rm -rf "$OUT/trace_$v"
done
exit 0
