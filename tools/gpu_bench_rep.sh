#!/bin/bash
# Repeatability of a bench config: gpu_bench_rep.sh <config> <repetitions> [bench args...]
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
cfg="$1"; n="$2"; shift 2
mkdir -p gpurun_out/r04
for i in $(seq 1 "$n"); do
  timeout -k 10 300 python bench.py --config "$cfg" --no-cpu-baseline "$@" > "gpurun_out/r04/rep_${cfg}_$i.json" 2> gpurun_out/r04/rep.err || { tail -10 gpurun_out/r04/rep.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/r04/rep_${cfg}_$i.json')); print('$cfg', $i, 'ms/step', round(d['ms_per_step'],5), 'blocks', d['block_ms_per_step'])"
done
