#!/bin/bash
# A/B of the hand-pipelined full-moments body (GVI_SREG_PIPE) + the bit-identity test.
set -o pipefail
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or c3_full or quad_prior or fixed_prior or scheduling" > gpurun_out/r02/pytest_pipe.log 2>&1; rc=$?
tail -5 gpurun_out/r02/pytest_pipe.log
[ $rc -ne 0 ] && exit $rc
for P in 0 1; do
  GVI_SREG_PIPE=$P timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02/bench_pipe$P.json 2> gpurun_out/r02/bench_pipe$P.err || { tail -20 gpurun_out/r02/bench_pipe$P.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02/bench_pipe$P.json"))
print("pipe=$P ms/step", d["ms_per_step"], "kernel ms", d["moments_kernel"]["ms"], "frac", d["roofline"]["frac"])
PY
done
