#!/bin/bash
# full GPU suite + default bench + PMC (traffic + SQ) of the dominant kernel + kernel trace
set -o pipefail
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.log 2>&1; rc=$?
tail -4 gpurun_out/r02/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err || { tail -20 gpurun_out/r02/bench.err; exit 1; }
cut -c1-400 gpurun_out/r02/bench.json
rm -rf gpurun_out/r02/pmc; bash tools/gpu_pmc_r02.sh > gpurun_out/r02/pmc_run.log 2>&1 || { tail -20 gpurun_out/r02/pmc_run.log; exit 1; }
tail -3 gpurun_out/r02/pmc_run.log
rm -rf gpurun_out/r02/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/trace -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r02/trace.log 2>&1 || { tail -20 gpurun_out/r02/trace.log; exit 1; }
find gpurun_out/r02/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02/kernel_stats.csv
head -12 gpurun_out/r02/kernel_stats.csv | cut -c1-130
