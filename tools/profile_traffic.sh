#!/bin/bash
# HBM traffic of one bench.py workload: a rocprofv3 kernel trace pass plus separate FETCH_SIZE and
# WRITE_SIZE --pmc passes (the short form of tools/profile_bench.sh, for the non-default workloads).
# Usage: tools/profile_traffic.sh <out-dir> [bench args...]; prints the pmc_traffic.json entry.
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-entry --no-cpu --no-recall --no-latency --no-e2e $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- $BENCH --no-accounting --steps 4 --warmup 1 > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err"
done
python "$REPO/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json"
find "$OUT" -name "*.db" -delete; find "$OUT" -type f -size +400k -delete   # (gpurun copies at most 64 MiB back: summaries only)
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
s = json.load(open(out + "/summary.json"))
line = json.loads(open(out + "/pmc_FETCH_SIZE.json").read().strip().splitlines()[-1])
def counter(tag, name):
    best = None
    for k, c in s["counters"].get(tag, {}).items():
        if k.startswith("seismic_search_kernel") and name in c and ", false," in k:   # the timed (uncounted) variant
            if best is None or c[name]["dispatches"] > best["dispatches"]:
                best = c[name]
    return best
f, w = counter("pmc_FETCH_SIZE", "FETCH_SIZE"), counter("pmc_WRITE_SIZE", "WRITE_SIZE")
ent = {"traffic_bytes": int(2 * f["mean"] * 1024 + w["mean"] * 1024), "fetch_size_kib": int(f["mean"]),
       "write_size_kib": int(w["mean"]), "dispatches": f["dispatches"],
       "kernel_source_id": line["roofline"]["kernel_source_id"],   # bench.py reports the figure only for this kernel source
       "algorithmic_bytes": json.loads(open(out + "/trace_bench.json").read().strip().splitlines()[-1])["roofline"]["algorithmic_bytes_per_launch"]}
ent["ratio"] = round(ent["traffic_bytes"] / ent["algorithmic_bytes"], 4) if ent["algorithmic_bytes"] else None
print(json.dumps({line["config"]["workload_key"]: ent}, indent=1))
PY
