#!/bin/bash
# chain kernels: stand-alone check + timing (tools/ubench/chain_bench.hip, built on the CPU side), all block sizes
set -e
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
mkdir -p gpurun_out
B=tools/ubench/chain_bench
{
  for n in 6 2 4 12 8 3 1 5 7 9 10 11 13 14 16; do timeout -k 10 120 $B 1025 $n 100; done
  timeout -k 10 120 $B 65 2 200
  timeout -k 10 120 $B 4097 12 50
  timeout -k 10 120 $B 40000 6 20 0
  timeout -k 10 120 $B 3000 14 20 0
  timeout -k 10 120 $B 7 6 20
  timeout -k 10 120 $B 7 5 20
  timeout -k 10 120 $B 1 6 20
  timeout -k 10 120 $B 2 4 20
  timeout -k 10 120 $B 33 6 20
  timeout -k 10 120 $B 32 6 20
  timeout -k 10 120 $B 1057 6 20
} > gpurun_out/chain_bench.log 2>&1 || { tail -30 gpurun_out/chain_bench.log; exit 1; }
grep "T =\|us per\|OK\|FAIL\|errors\|run-to" gpurun_out/chain_bench.log
