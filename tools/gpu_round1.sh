#!/bin/bash
# One gpurun call: bench + rocprof kernel trace + PMC passes for the dominant kernel.
set -o pipefail
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof/trace.log 2>&1 || { tail -20 gpurun_out/prof/trace.log; exit 1; }
find gpurun_out/prof/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof/kernel_stats.csv
head -12 gpurun_out/prof/kernel_stats.csv
