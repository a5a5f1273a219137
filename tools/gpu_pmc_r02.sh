#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) for the dominant kernels of the bench.
set -o pipefail
mkdir -p gpurun_out/r02/pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail > gpurun_out/r02/pmc/list_avail.txt 2>&1
run() { # name, counters...
  local name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/r02/pmc/$name -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02/pmc/$name.log 2>&1 || { tail -5 gpurun_out/r02/pmc/$name.log; return 1; }
}
run SQ1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES || exit 1
run SQ2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS || echo SQ2 failed
run SQ3 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH || echo SQ3 failed
run FETCH FETCH_SIZE || echo FETCH failed
run WRITE WRITE_SIZE || echo WRITE failed
run TCC TCC_HIT_sum TCC_MISS_sum || echo TCC failed
python3 tools/summarize_pmc.py gpurun_out/r02/pmc > gpurun_out/r02/pmc/summary.json; head -c 3000 gpurun_out/r02/pmc/summary.json
rm -rf gpurun_out/r02/trace_c2
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/trace_c2 -- python3 bench.py --config c2 --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/r02/trace_c2.log 2>&1 || tail -5 gpurun_out/r02/trace_c2.log
find gpurun_out/r02/trace_c2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02/kernel_stats_c2.csv
head -14 gpurun_out/r02/kernel_stats_c2.csv | cut -c1-150
