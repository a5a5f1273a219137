#!/bin/bash
# The A/B legs of the scheduling switches: the GPU suite under each of them (environment form)
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
mkdir -p gpurun_out/r04
for e in "GVI_PIPELINE=0" "GVI_FUSED=0" "GVI_CHAIN_WAVE=0 GVI_ASM_ON_LOAD=0 GVI_CHAIN_MERGE=0"; do
  env $e timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/legs.log 2>&1; rc=$?
  echo "[$e] rc=$rc $(tail -1 gpurun_out/r04/legs.log)"
  [ $rc -ne 0 ] && { tail -30 gpurun_out/r04/legs.log; exit $rc; }
done
exit 0
