#!/bin/bash
# quick loop: scheduling/parity subset + default bench + kernel trace
set -o pipefail
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${GVI_TEST_K:-scheduling or ngd_iterations or chain_step or k8 or side_stream or headline or linesearch or prox or obstacle_chains}" > gpurun_out/r02/pytest_quick.log 2>&1; rc=$?
tail -5 gpurun_out/r02/pytest_quick.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02/bench_quick.json 2> gpurun_out/r02/bench_quick.err || { tail -20 gpurun_out/r02/bench_quick.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r02/bench_quick.json"))
print("c3 ms/step", d["ms_per_step"], "kernel ms", d["moments_kernel"]["ms"], "frac", d["roofline"]["frac"], "final", d["final_cost"], "ref order", d["reference_pass_order"]["ms_per_step"])
PY
timeout -k 10 200 python bench.py --config c2 --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/r02/bench_quick_c2.json 2> gpurun_out/r02/bench_quick_c2.err || { tail -20 gpurun_out/r02/bench_quick_c2.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r02/bench_quick_c2.json"))
print("c2 ms/step", d["ms_per_step"], "value", d["value"], "final", d["final_cost"])
PY
rm -rf gpurun_out/r02/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/trace -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r02/trace.log 2>&1 || { tail -20 gpurun_out/r02/trace.log; exit 1; }
find gpurun_out/r02/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02/kernel_stats.csv
head -14 gpurun_out/r02/kernel_stats.csv | cut -c1-130
