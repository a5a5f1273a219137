#!/bin/bash
# Full GPU test-suite + mirror A/B + C5 bench line.
set -o pipefail
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > gpurun_out/r02/pytest_gpu.log 2>&1; rc=$?
tail -25 gpurun_out/r02/pytest_gpu.log
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
cut -c1-1200 gpurun_out/r02/bench_c5.json
