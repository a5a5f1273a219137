#!/bin/bash
# rocprofv3 kernel trace of bench.py under every environment setting of ENVS ("A=1 B=0|A=0 B=0": settings separated by |):
# average duration of the kernels matching KERNEL_RE.  (The variables are exported in a subshell: nothing stands between
# rocprofv3 and python3.)
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
R="${R:-r04}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT="gpurun_out/$R"; mkdir -p "$OUT"
IFS='|' read -ra SETS <<< "${ENVS:-}"
[ ${#SETS[@]} -eq 0 ] && SETS=("")
i=0
for e in "${SETS[@]}"; do
  rm -rf "$OUT/trace_env$i"
  ( for kv in $e; do export "$kv"; done
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_env$i" -- python3 bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline ${GVI_BENCH_ARGS:-} > "$OUT/trace_env$i.log" 2>&1 ) || { tail -20 "$OUT/trace_env$i.log"; exit 1; }
  f=$(find "$OUT/trace_env$i" -name "*kernel_stats.csv" | head -1)
  echo "== [$e] $(grep -o '"ms_per_step": [0-9.]*' "$OUT/trace_env$i.log" | head -1)"; grep -E "${KERNEL_RE:-fused|chain}" "$f" | cut -d, -f1-4 | cut -c1-170
  cp "$f" "$OUT/kernel_stats_env$i.csv"; rm -rf "$OUT/trace_env$i"
  i=$((i+1))
done
exit 0
