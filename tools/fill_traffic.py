#!/usr/bin/env python3
"""A bench line printed in the same gpurun call as the counter passes of its workload has `roofline.traffic: null` - the
entry of profiles/pmc_traffic.json did not exist yet. This fills the figure in afterwards, only when the entry was measured
at the line's own kernel source (roofline.kernel_source_id), and says so in `traffic_note`.
  python tools/fill_traffic.py profiles/r06_bench_*.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["workloads"]
for path in sys.argv[1:]:
    lines = open(path).read().strip().splitlines()
    j = json.loads(lines[-1])
    r = j.get("roofline") or {}
    ent = pm.get((j.get("config") or {}).get("workload_key"))
    if r.get("traffic") is None and ent and ent.get("kernel_source_id") == r.get("kernel_source_id"):
        r["traffic"] = float(ent["traffic_bytes"])
        r["traffic_note"] = ("filled in after the run by tools/fill_traffic.py: the counter passes of this workload (%s) ran in the same "
                             "gpurun call as this line, at the same kernel source %s" % (ent.get("recorded"), ent["kernel_source_id"]))
        lines[-1] = json.dumps(j)
        open(path, "w").write("\n".join(lines) + "\n")
        print(path, "traffic", r["traffic"], "= %.3f x algorithmic" % (r["traffic"] / r["algorithmic_bytes_per_launch"]))
    else:
        print(path, "unchanged (traffic %s)" % r.get("traffic"))
    changed = False
    for pt in j.get("operating_points", []) or []:   # (an operating point carries its workload key and traffic itself)
        e2 = pm.get(pt.get("workload_key"))
        if pt.get("traffic") is None and e2 and e2.get("kernel_source_id") == r.get("kernel_source_id"):
            pt["traffic"] = float(e2["traffic_bytes"])
            pt["traffic_note"] = "filled in after the run by tools/fill_traffic.py (%s, kernel source %s)" % (e2.get("recorded"), e2["kernel_source_id"])
            changed = True
            print("   operating point %s: traffic %.3f x algorithmic" % (pt.get("target_recall"), pt["traffic"] / pt["algorithmic_bytes_per_launch"]))
    if changed:
        lines[-1] = json.dumps(j)
        open(path, "w").write("\n".join(lines) + "\n")
