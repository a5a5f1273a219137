#!/bin/bash
# chain kernels: quick check + timing of the headline shapes
set -e
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
mkdir -p gpurun_out
B=tools/ubench/chain_bench
{
  timeout -k 10 120 $B 1025 6 300
  timeout -k 10 120 $B 65 2 300
  timeout -k 10 120 $B 7 6 300
  timeout -k 10 120 $B 4097 12 50
  timeout -k 10 120 $B 1025 4 100
  timeout -k 10 120 $B 1025 8 100
} > gpurun_out/chain_bench_q.log 2>&1 || { tail -30 gpurun_out/chain_bench_q.log; exit 1; }
cat gpurun_out/chain_bench_q.log
