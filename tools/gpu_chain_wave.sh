#!/bin/bash
# Lane-per-node chain kernel (kernels_chain_wave.hpp) against the sequential host solver: T <= 65, n <= 2
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
cd "$ROOT"
for tn in "65 2" "64 2" "63 2" "33 2" "32 2" "17 2" "7 2" "3 2" "2 2" "1 2" "65 1" "20 1"; do
  set -- $tn
  echo "== T=$1 n=$2"
  timeout -k 10 120 tools/ubench/chain_bench $1 $2 200 0 2>&1 | tail -8 || exit 1
done
