#!/bin/bash
# PMC passes (separate runs, --kernel-trace only -- never with other trace domains) for the kernels of the bench.
# Usage: gpu_pmc.sh [bench args...]   output: gpurun_out/$R/pmc/summary.json   (R defaults to r03)
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
R="${R:-r04}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O="gpurun_out/$R/pmc"
mkdir -p "$O"
run() { # name, counters...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$O/$name" -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline $BENCH_ARGS > "$O/$name.log" 2>&1 || { tail -5 "$O/$name.log"; return 1; }
}
BENCH_ARGS="$*"
run SQ1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES || exit 1
run SQ2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS || echo SQ2 failed
run SQ3 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH || echo SQ3 failed
run LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN || echo LDS failed
run FETCH FETCH_SIZE || echo FETCH failed
run WRITE WRITE_SIZE || echo WRITE failed
run TCC TCC_HIT_sum TCC_MISS_sum || echo TCC failed
python3 tools/summarize_pmc.py "$O" > "$O/summary.json"; head -c 1500 "$O/summary.json"
