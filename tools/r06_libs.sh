#!/bin/bash
# The kernel leg of bench.py on both collections for several builds of the library on ONE box (boxes differ by +-2 %).
# Usage: tools/r06_libs.sh <tag> <lib suffix or "-" for the product> ...   (libseismic_hip_<suffix>.so: make exp NAME=<suffix>)
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; shift
OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"
export SGPU_INDEX_CACHE=/tmp SGPU_TEST_HOOKS=1
cd "$REPO"
LEAN="--no-entry --no-cpu --no-recall --no-latency --no-e2e --target-recall= --steps 20 --warmup 3"
for L in "$@"; do
  if [ "$L" = "-" ]; then unset SGPU_LIB; N=product; else export SGPU_LIB="$REPO/seismic_amd/libseismic_hip_$L.so"; N=$L; fi
  for C in survey clustered; do
    python bench.py $LEAN --collection $C > "$OUT/bench_${N}_$C.json" 2> "$OUT/bench_${N}_$C.err"
    python - "$OUT/bench_${N}_$C.json" $N $C <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print("%-10s %-9s kernel_ms %.3f frac %.3f identical %s" % (sys.argv[2], sys.argv[3], r["kernel_ms"], r["frac"], r["counted_pass_identical"]))
PY
  done
done | tee "$OUT/summary.txt"
