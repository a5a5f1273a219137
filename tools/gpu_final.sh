#!/bin/bash
# round-end validation: full GPU suite, smoke(), default bench (+ cpu baseline), C2 / C5 benches, kernel trace, 2-rank rehearsal
set -o pipefail
mkdir -p gpurun_out/r02/final
O=gpurun_out/r02/final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -4 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("c3 ms/step", d["ms_per_step"], "value", d["value"], "kernel ms", d["moments_kernel"]["ms"], "frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"], "ref order", d["reference_pass_order"]["ms_per_step"])
PY
timeout -k 10 200 python bench.py --config c2 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err || { tail -20 $O/bench_c2.err; exit 1; }
timeout -k 10 400 python bench.py --config c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err || { tail -20 $O/bench_c5.err; exit 1; }
python - <<PY
import json
for c in ("c2", "c5"):
    d=json.load(open("$O/bench_%s.json" % c))
    print(c, "ms/step", d["ms_per_step"], "value", d["value"], "final", d["final_cost"], "variant", d["config"]["kernel_variant"])
PY
rm -rf $O/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -12 $O/kernel_stats.csv | cut -c1-130
GVI_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 60 --warmup 10 > $O/bench_rehearsal2.log 2>&1 || { tail -20 $O/bench_rehearsal2.log; exit 1; }
grep '^{' $O/bench_rehearsal2.log | tail -1 > $O/bench_rehearsal2.json
python - <<PY
import json
d=json.load(open("$O/bench_rehearsal2.json"))
print("rehearsal n=2 final", d["final_cost"], "ms/step", d["ms_per_step"])
PY
# steady-state iteration time p and restart cost R from two restart periods: t(r) = r p + R
for r in 15 30; do
  timeout -k 10 200 python bench.py --steps 600 --warmup 30 --restart-every $r --no-cpu-baseline > $O/bench_restart$r.json 2>/dev/null || exit 1
done
python - <<PY
import json
t = {r: json.load(open("$O/bench_restart%d.json" % r))["ms_per_step"] * r for r in (15, 30)}
p = (t[30] - t[15]) / 15.0
print("steady-state iteration p = %.2f us, restart R = %.1f us" % (1e3 * p, 1e3 * (t[15] - 15 * p)))
PY
