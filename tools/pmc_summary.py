#!/usr/bin/env python3
"""Summarise the rocprofv3 CSV outputs under a directory written by tools/profile_bench.sh:
per kernel the dispatch count and mean duration (kernel trace), and per (kernel, counter) the mean
counter value per dispatch (PMC passes). JSON on stdout."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = {"kernel_stats": [], "counters": {}, "durations_us": {}}


def short(name):
    name = name.replace("void sgpu::", "").replace("sgpu::", "")
    return name.split("(")[0]


for f in sorted(glob.glob(os.path.join(root, "**", "*.csv"), recursive=True)):
    rel = os.path.relpath(f, root)
    with open(f, newline="") as fh:
        rows = list(csv.DictReader(fh))
    if not rows:
        continue
    cols = rows[0].keys()
    if f.endswith("kernel_stats.csv"):
        for r in rows:
            out["kernel_stats"].append({"file": rel, "kernel": short(r.get("Name", "")), "calls": int(r.get("Calls", 0)),
                                        "avg_us": float(r.get("AverageNs", 0)) / 1e3,
                                        "total_us": float(r.get("TotalDurationNs", 0)) / 1e3,
                                        "min_us": float(r.get("MinNs", 0)) / 1e3, "max_us": float(r.get("MaxNs", 0)) / 1e3,
                                        "percent": float(r.get("Percentage", 0))})
    elif "Counter_Name" in cols:
        acc = defaultdict(list)
        for r in rows:
            acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            out["counters"].setdefault(rel.split(os.sep)[0], {}).setdefault(k, {})[c] = {
                "dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
    elif "Start_Timestamp" in cols and "Kernel_Name" in cols:
        acc = defaultdict(list)
        for r in rows:
            acc[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        out["durations_us"][rel.split(os.sep)[0]] = {k: {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v),
                                                         "max": max(v)} for k, v in acc.items()}
json.dump(out, sys.stdout, indent=1)
