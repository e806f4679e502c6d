#!/bin/bash
# Phase clocks of the profiling build under a list of knob settings, both collections (one index build each, cached in /tmp).
# Usage: tools/r06_sweep.sh <tag> "<ENV=V ENV=V>" "<ENV=V ...>" ...
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; shift
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
export SGPU_INDEX_CACHE=/tmp SGPU_TEST_HOOKS=1
cd "$REPO"
for C in ${COLLECTIONS:-1 0}; do
  for S in "$@"; do
    echo "== collection $C  $S" | tee -a "$OUT/sweep.txt"
    env $S timeout 300 python tools/phase_profile.py --docs 8800000 --n-postings 2000 --queries 10000 --collection $C 2>&1 | grep -E "kernel ms|per query|phaseB|replay|stream/query|work/query|row_table|summary_dots" | tee -a "$OUT/sweep.txt"
  done
done
