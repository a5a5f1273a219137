#!/bin/bash
# sign-orbit kernel: parity suite with it as the default, then A/B benches against the lane-per-point kernels
set -o pipefail
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_cpp_shim.py -m gpu -q ${GVI_TEST_K:+-k "$GVI_TEST_K"} > gpurun_out/r02/pytest_orbit.log 2>&1; rc=$?
tail -8 gpurun_out/r02/pytest_orbit.log
[ $rc -ne 0 ] && exit $rc
for cfg in "4096 8" "8192 8" "4096 1" "4096 4" "4096 16" "2048 8"; do
  set -- $cfg; w=$1; cp=$2
  GVI_ORBIT_WAVES=$w GVI_ORBIT_COPIES=$cp timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02/bench_orbit_w${w}_c$cp.json 2> gpurun_out/r02/bench_orbit.err || { tail -20 gpurun_out/r02/bench_orbit.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02/bench_orbit_w${w}_c$cp.json"))
print("orbit waves $w copies $cp: c3 ms/step", d["ms_per_step"], "kernel ms", d["moments_kernel"]["ms"], "frac", d["roofline"]["frac"], "final", d["final_cost"], "variant", d["config"]["kernel_variant"], "chunks", d["config"]["chunks_per_factor"])
PY
done
GVI_ORBIT=0 timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02/bench_orbit_off.json 2> gpurun_out/r02/bench_orbit.err || { tail -20 gpurun_out/r02/bench_orbit.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r02/bench_orbit_off.json"))
print("orbit off: c3 ms/step", d["ms_per_step"], "kernel ms", d["moments_kernel"]["ms"], "frac", d["roofline"]["frac"], "final", d["final_cost"], "variant", d["config"]["kernel_variant"])
PY
timeout -k 10 400 python bench.py --config c5 --no-cpu-baseline > gpurun_out/r02/bench_orbit_c5.json 2> gpurun_out/r02/bench_orbit_c5.err || { tail -20 gpurun_out/r02/bench_orbit_c5.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r02/bench_orbit_c5.json"))
print("c5 ms/step", d["ms_per_step"], "value", d["value"], "kernel ms", d["moments_kernel"]["ms"], "final", d["final_cost"], "variant", d["config"]["kernel_variant"])
PY
rm -rf gpurun_out/r02/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/trace -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r02/trace.log 2>&1 || { tail -20 gpurun_out/r02/trace.log; exit 1; }
find gpurun_out/r02/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02/kernel_stats_orbit.csv
head -14 gpurun_out/r02/kernel_stats_orbit.csv | cut -c1-130
