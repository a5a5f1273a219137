#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) for the dominant kernel + forced-collective check.
set -o pipefail
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc/$C -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/$C.log 2>&1 || { tail -20 gpurun_out/pmc/$C.log; exit 1; }
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc/SQ -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/SQ.log 2>&1 || tail -5 gpurun_out/pmc/SQ.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc/TCC -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/TCC.log 2>&1 || tail -5 gpurun_out/pmc/TCC.log
python3 tools/summarize_pmc.py gpurun_out/pmc > gpurun_out/pmc/summary.json; cat gpurun_out/pmc/summary.json
# distributed path on one GPU: RCCL group of size 1 with the all-reduce forced
GVI_FORCE_ALLREDUCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/dist1.json 2> gpurun_out/dist1.err; echo dist_exit=$?; tail -3 gpurun_out/dist1.err; cut -c1-300 gpurun_out/dist1.json
