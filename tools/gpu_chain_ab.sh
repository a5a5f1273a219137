#!/bin/bash
# chain kernels: A/B of harness builds (arguments: binaries under tools/ubench)
set -e
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
mkdir -p gpurun_out
{
  for B in "$@"; do
    echo "== $B"
    timeout -k 10 120 tools/ubench/$B 1025 6 300
    timeout -k 10 120 tools/ubench/$B 65 2 300
    timeout -k 10 120 tools/ubench/$B 7 6 300
  done
} > gpurun_out/chain_bench_ab.log 2>&1 || { tail -30 gpurun_out/chain_bench_ab.log; exit 1; }
grep "==\|T =\|us per\|OK\|FAIL" gpurun_out/chain_bench_ab.log
