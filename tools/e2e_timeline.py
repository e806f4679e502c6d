"""Device timeline of sgpu_batch_search calls from a rocprofv3 --kernel-trace csv: start / end of every kernel relative to the
first kernel of its call (calls are separated by gaps with no kernel resident)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
    if "plan_cost" in n: return "plan_cost"
    if "plan_sort" in n: return "plan_sort"
    if "search_kernel" in n: return "search" + ("(streamed)" if n.rstrip(">) ").endswith("true") else "")
    return n[:40]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-last:]
t0 = rows[0][0]
prev_end = None
for s, e, n in rows:
    if prev_end is not None and s - prev_end > 50_000:
        t0 = s
        print("--")
    print("%9.1f .. %9.1f us  (%8.1f)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, short(n)))
    prev_end = max(prev_end or 0, e)
