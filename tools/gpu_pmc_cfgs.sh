#!/bin/bash
# PMC passes for further bench configs: planar1k (hinge kernel) and c5 (orbit kernel <12, 6>); summaries under gpurun_out/<R>/pmc
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
cd "$ROOT"
R=r03_planar1k tools/gpu_pmc.sh --config planar1k > /dev/null 2>&1; echo "planar1k pmc rc $?"
R=r03_c5 tools/gpu_pmc.sh --config c5 > /dev/null 2>&1; echo "c5 pmc rc $?"
ls -la gpurun_out/r03_planar1k/pmc/summary.json gpurun_out/r03_c5/pmc/summary.json
