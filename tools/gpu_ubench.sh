#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02/ubench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
./tools/ubench/fp64_pipes all > gpurun_out/r02/ubench/fp64_pipes.txt 2>&1 || { cat gpurun_out/r02/ubench/fp64_pipes.txt; exit 1; }
cat gpurun_out/r02/ubench/fp64_pipes.txt
for M in valu mfma8 mfma16 both; do
  timeout -k 10 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r02/ubench/pmc_$M -- ./tools/ubench/fp64_pipes $M > gpurun_out/r02/ubench/pmc_$M.log 2>&1 || { tail -5 gpurun_out/r02/ubench/pmc_$M.log; }
done
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for m in ("valu", "mfma8", "mfma16", "both"):
    for f in glob.glob(f"gpurun_out/r02/ubench/pmc_{m}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        # the LAST dispatch of the run = 4 waves/SIMD, long kernel
        last = max(int(r["Dispatch_Id"]) for r in rows)
        out[m] = {r["Counter_Name"]: float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == last}
json.dump(out, open("gpurun_out/r02/ubench/pmc_summary.json", "w"), indent=1)
for m, c in out.items():
    busy = c.get("SQ_BUSY_CYCLES", 0)
    print(m, {k: v for k, v in c.items()}, "mfma_busy/busy" , (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy) if busy else None)
PY
