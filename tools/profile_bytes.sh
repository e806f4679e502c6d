#!/bin/bash
# One bytes experiment on the bench workload: kernel duration + FETCH_SIZE, WRITE_SIZE, TCC_HIT / TCC_MISS of the timed
# search kernel, each counter group in its own rocprofv3 --pmc pass (kernel trace only). Environment = the experiment.
# Usage: [ENV=...] tools/profile_bytes.sh <out-dir> <label> [bench args...]; appends one JSON line to <out-dir>/results.jsonl
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/$1/$2"; LABEL=$2; shift 2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-entry --no-cpu --no-recall --no-latency --no-e2e --no-accounting --target-recall= --steps 6 --warmup 2 $*"
# (PMC_SETS="A B;C" overrides the counter groups, one rocprofv3 pass each)
IFS=';' read -ra SETS <<< "${PMC_SETS:-FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum}"
for C in "${SETS[@]}"; do
  tag=$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$tag" -o pmc -- $BENCH > "$OUT/pmc_$tag.json" 2> "$OUT/pmc_$tag.err"
done
python "$REPO/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json"
find "$OUT" -name "*.db" -delete; find "$OUT" -type f -size +400k -delete
python - "$OUT" "$LABEL" >> "$OUT/../results.jsonl" <<'PY'
import json, sys
out, label = sys.argv[1], sys.argv[2]
s = json.load(open(out + "/summary.json"))
def best(tag, name):
    b = None
    for k, c in s["counters"].get(tag, {}).items():
        if k.startswith("seismic_search_kernel") and name in c and ", false," in k:
            if b is None or c[name]["dispatches"] > b["dispatches"]:
                b = c[name]
    return b
res = {"label": label}
import glob, os
for d_ in sorted(glob.glob(out + "/pmc_*")):
    if not os.path.isdir(d_):
        continue
    tag = os.path.basename(d_)
    for k, c in s["counters"].get(tag, {}).items():
        if k.startswith("seismic_search_kernel") and ", false," in k:
            for n, v in c.items():
                if res.get(n + "_dispatches", 0) < v["dispatches"]:
                    res[n] = v["mean"]
                    res[n + "_dispatches"] = v["dispatches"]
    d = s["durations_us"].get(tag, {})
    ks = [v for k, v in d.items() if k.startswith("seismic_search_kernel") and ", false," in k]
    if ks:
        res.setdefault("kernel_us", {})[tag] = max(ks, key=lambda v: v["dispatches"])["mean"]
res = {k: v for k, v in res.items() if not k.endswith("_dispatches")}
try:
    line = json.loads(open(sorted(glob.glob(out + "/pmc_*.json"))[0]).read().strip().splitlines()[-1])
    res["bench_kernel_ms"] = line["roofline"]["kernel_ms"]
    res["hbm_bytes"] = line["config"]["index"]["hbm_bytes"]
except Exception as e:
    res["bench_error"] = str(e)
if res.get("FETCH_SIZE") is not None and res.get("WRITE_SIZE") is not None:
    res["traffic_bytes"] = int(2 * res["FETCH_SIZE"] * 1024 + res["WRITE_SIZE"] * 1024)   # (gfx950 correction of the guide: FETCH_SIZE counts 2 KiB units)
print(json.dumps(res))
PY
