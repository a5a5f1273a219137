#!/bin/bash
# One gpurun call: GPU test-suite, default bench line, rocprof kernel trace of the bench.
set -o pipefail
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.log 2>&1; rc=$?
tail -5 gpurun_out/r02/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err || { tail -20 gpurun_out/r02/bench.err; exit 1; }
cat gpurun_out/r02/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/trace -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r02/trace.log 2>&1 || { tail -20 gpurun_out/r02/trace.log; exit 1; }
find gpurun_out/r02/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02/kernel_stats.csv
head -16 gpurun_out/r02/kernel_stats.csv
