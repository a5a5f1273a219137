#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/pmc2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc2/A -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --variant ${1:-0} > gpurun_out/pmc2/A.log 2>&1 || tail -5 gpurun_out/pmc2/A.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc2/B -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --variant ${1:-0} > gpurun_out/pmc2/B.log 2>&1 || tail -5 gpurun_out/pmc2/B.log
python3 tools/summarize_pmc.py gpurun_out/pmc2 > gpurun_out/pmc2/summary.json
python3 - <<'PY'
import json
s=json.load(open('gpurun_out/pmc2/summary.json'))
for k,v in s.items():
    if 'true' in k or 'ELb1' in k:
        print(k[:70]); print({c:round(x['mean']) for c,x in v.items()})
PY
