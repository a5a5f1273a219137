#!/bin/bash
# iteration loop: parity subset (GVI_TEST_K), c3 bench under each environment of GVI_BENCH_ENVS ("A=1 B=2|C=3|" -- the
# empty entry is the default environment), kernel trace of the default
set -o pipefail
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_cpp_shim.py -m gpu -x -q ${GVI_TEST_K:+-k "$GVI_TEST_K"} > gpurun_out/r02/pytest_iter.log 2>&1; rc=$?
tail -5 gpurun_out/r02/pytest_iter.log
[ $rc -ne 0 ] && exit $rc
IFS='|' read -ra ENVS <<< "${GVI_BENCH_ENVS:-|}"
[ ${#ENVS[@]} -eq 0 ] && ENVS=("")
i=0
for e in "${ENVS[@]}"; do
  env $e timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/r02/bench_iter_$i.json 2> gpurun_out/r02/bench_iter.err || { tail -20 gpurun_out/r02/bench_iter.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02/bench_iter_$i.json"))
print("[$e] c3 ms/step", round(d["ms_per_step"], 5), "psi kernel ms", round(d["moments_kernel"]["ms"], 5), "final", d["final_cost"], "accepted", d["accepted_steps"], "passes", d["passes"]["full"], d["passes"]["cost_only"])
PY
  i=$((i+1))
done
rm -rf gpurun_out/r02/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/trace -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r02/trace.log 2>&1 || { tail -20 gpurun_out/r02/trace.log; exit 1; }
find gpurun_out/r02/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02/kernel_stats_iter.csv
head -12 gpurun_out/r02/kernel_stats_iter.csv | cut -c1-120
