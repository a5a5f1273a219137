#!/bin/bash
# GPU test-suite (new tests of round 2), C5 bench line.
set -o pipefail
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s -k "c5 or literal or thirty or temperature or recorder or joint_loop or empty_shards" > gpurun_out/r02/pytest_new.log 2>&1; rc=$?
tail -25 gpurun_out/r02/pytest_new.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --config c5 --no-cpu-baseline > gpurun_out/r02/bench_c5.json 2> gpurun_out/r02/bench_c5.err || { tail -20 gpurun_out/r02/bench_c5.err; exit 1; }
cut -c1-1500 gpurun_out/r02/bench_c5.json
