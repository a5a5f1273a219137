#!/bin/bash
# chain kernels: phase stamps (tools/ubench/chain_bench_t = chain_bench.hip built with -DGVI_CHAIN_TIMING)
set -e
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
mkdir -p gpurun_out
B=tools/ubench/chain_bench_t
{
  timeout -k 10 120 $B 7 6 50 0
  timeout -k 10 120 $B 7 2 50 0
  timeout -k 10 120 $B 1025 6 50 0
} > gpurun_out/chain_bench_t.log 2>&1 || { tail -30 gpurun_out/chain_bench_t.log; exit 1; }
cat gpurun_out/chain_bench_t.log
