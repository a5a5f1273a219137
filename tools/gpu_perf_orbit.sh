#!/bin/bash
# tools/perf_orbit.py under every library of VARIANTS (see tools/gpu_variants.sh)
ROOT="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for v in ${VARIANTS:-default}; do
  lib=""; [ "$v" != "default" ] && lib="$ROOT/build/variants/libgvi_hip_$v.so"
  echo "== $v ${GVI_ENV:-}"
  env GVI_LIB_PATH="$lib" ${GVI_ENV:-} timeout -k 10 120 python tools/perf_orbit.py ${CONFIG:-c3} 2>&1 | tail -4 || exit 1
done
exit 0
