#!/bin/bash
# LDS counters of the kernels of bench.py under every library of VARIANTS (see tools/gpu_variants.sh): one --pmc pass each
# (--kernel-trace only beside it).  Output: gpurun_out/$R/pmc_lds_<variant>/, a per-kernel summary on stdout.
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
R="${R:-r04}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for v in ${VARIANTS:-default}; do
  lib=""; [ "$v" != "default" ] && lib="$ROOT/build/variants/libgvi_hip_$v.so"
  O="gpurun_out/$R/pmc_lds_$v"; rm -rf "$O"; mkdir -p "$O"
  ( export GVI_LIB_PATH="$lib"
    timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d "$O" -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline ${GVI_BENCH_ARGS:-} > "$O.log" 2>&1 ) || { tail -5 "$O.log"; exit 1; }
  python3 - "$O" "$v" <<'PY'
import csv, glob, sys, collections
d, v = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "fused" in r["Kernel_Name"] or "orbit_pair" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(v, k, {n: round(sum(x) / len(x)) for n, x in sorted(c.items())})
PY
done
exit 0
