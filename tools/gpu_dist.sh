#!/bin/bash
# multi-rank paths on ONE GPU: in-library exchange tests, rehearsal of bench.py --gpus 2/4 (gloo callback), RCCL size-1 group
set -o pipefail
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_dist_gpu.py tests/test_dist_gloo.py -m gpu -x -q > gpurun_out/r02/pytest_dist.log 2>&1; rc=$?
tail -6 gpurun_out/r02/pytest_dist.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02/reh1.json 2> gpurun_out/r02/reh1.err || { tail -20 gpurun_out/r02/reh1.err; exit 1; }
for N in 2 4; do
  GVI_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02/reh$N.json 2> gpurun_out/r02/reh$N.err || { tail -30 gpurun_out/r02/reh$N.err; exit 1; }
done
GVI_FORCE_ALLREDUCE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02/rccl1.json 2> gpurun_out/r02/rccl1.err || { tail -30 gpurun_out/r02/rccl1.err; exit 1; }
python - <<'PY'
import json
for n in ("reh1", "reh2", "reh4", "rccl1"):
    d = json.loads([l for l in open(f"gpurun_out/r02/{n}.json") if l.startswith("{")][-1])
    print(n, "final_cost", d["final_cost"], "ms/step", d["ms_per_step"], "scaling", d["scaling"], "rccl_ranks", d.get("rccl_ranks"), d["config"]["sharding"][:90])
PY
