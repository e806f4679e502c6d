#!/bin/bash
# Kernel leg of bench.py on both collections + the phase clocks, for one library (default: the product) - the A/B of the
# round's kernel work: run once per library on ONE box (boxes differ by +-2 %).
# Usage: tools/r06_ab.sh <tag> [extra env assignments, e.g. SGPU_STREAM=0]
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; shift
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
export SGPU_INDEX_CACHE=/tmp SGPU_TEST_HOOKS=1
for kv in "$@"; do export "$kv"; done
cd "$REPO"
LEAN="--no-entry --no-cpu --no-recall --no-latency --no-e2e --target-recall= --steps 20 --warmup 3"
python bench.py $LEAN > "$OUT/bench_survey.json" 2> "$OUT/bench_survey.err"
python bench.py $LEAN --collection clustered > "$OUT/bench_clustered.json" 2> "$OUT/bench_clustered.err"
if [ -f seismic_amd/libseismic_hip_prof.so ]; then
  python tools/phase_profile.py --docs 8800000 --n-postings 2000 --queries 10000 --collection 1 > "$OUT/phase_clustered.txt" 2>&1
  python tools/phase_profile.py --docs 8800000 --n-postings 2000 --queries 10000 --collection 0 > "$OUT/phase_survey.txt" 2>&1
fi
tail -c 1500 "$OUT/bench_survey.json"; echo; tail -c 1500 "$OUT/bench_clustered.json"; echo
grep -h "kernel ms\|per query" "$OUT"/phase_*.txt
