#!/bin/bash
# Phase stamps of the fused factor launch (library built with GVI_BUILD_DEFINES=GVI_FUSED_TIMING): bench c3 with GVI_FUSED_DBG=8
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
cd "$ROOT"; mkdir -p gpurun_out/r04
GVI_FUSED_DBG=8 timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline ${GVI_BENCH_ARGS:-} > gpurun_out/r04/fused_t.json 2> gpurun_out/r04/fused_t.err
grep "stamps" gpurun_out/r04/fused_t.err | head -80
python -c "import json; d=json.load(open('gpurun_out/r04/fused_t.json')); print('ms/step', d['ms_per_step'])"
