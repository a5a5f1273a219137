#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or asymmetric or quad_prior or fixed_prior or wide_factors or c3_full or scheduling or c5 or ngd_iterations" > gpurun_out/r02/pytest_sub.log 2>&1; rc=$?
tail -5 gpurun_out/r02/pytest_sub.log
[ $rc -ne 0 ] && exit $rc
for P in 0 1; do
  GVI_MIRROR=$P timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02/bench_mirror$P.json 2> gpurun_out/r02/bench_mirror$P.err || { tail -20 gpurun_out/r02/bench_mirror$P.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02/bench_mirror$P.json"))
print("mirror=$P ms/step", d["ms_per_step"], "kernel ms", d["moments_kernel"]["ms"], "frac", d["roofline"]["frac"], "final", d["final_cost"])
PY
done
timeout -k 10 300 python bench.py --config c5 --no-cpu-baseline > gpurun_out/r02/bench_c5.json 2> gpurun_out/r02/bench_c5.err || { tail -20 gpurun_out/r02/bench_c5.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r02/bench_c5.json"))
print("c5 ms/step", d["ms_per_step"], "kernel ms", d["moments_kernel"]["ms"], "frac", d["roofline"]["frac"], "value", d["value"])
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/trace -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r02/trace.log 2>&1 || { tail -20 gpurun_out/r02/trace.log; exit 1; }
find gpurun_out/r02/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02/kernel_stats.csv
head -14 gpurun_out/r02/kernel_stats.csv | cut -c1-160
