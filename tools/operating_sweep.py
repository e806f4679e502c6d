#!/usr/bin/env python3
"""Operating points the way the reference makes them: recall targets are reached through INDEX parameters
(n_postings, max_fraction, ...) with a small query_cut, not by widening query_cut on one index
(reference experiments/best_configs/msmarco-v1/splade-v3/mem_budget_2.0/recall_90 ... recall_99.toml).

For every index configuration of --configs (n_postings:max_fraction[:centroid_fraction[:summary_energy]]) the
collection is indexed (device-assisted build), uploaded, and a (query_cut, heap_factor, first_sorted) grid is
run on ONE resident batch of --queries queries: kernel time of the whole launch (HIP events, best of --reps)
and recall@k of the batch's first --sample queries against the exact top-k. One JSON line per grid point is
appended to --out as it is measured; the last line holds, per recall target, the cheapest point.

  python tools/operating_sweep.py --out gpurun_out/r04_sweep.jsonl
  python tools/operating_sweep.py --summarise gpurun_out/r04_sweep.jsonl > profiles/operating_points.json
"""
import argparse
import json
import os
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarise(rows, targets):
    """Per target: the cheapest measured point (kernel time of the whole launch) that reaches it."""
    pts = [r for r in rows if "recall" in r]
    out = []
    for t in targets:
        ok = [p for p in pts if p["recall"] >= t]
        if not ok:
            best = max(pts, key=lambda p: p["recall"]) if pts else None
            out.append({"target_recall": t, "reached": False, "best": best})
            continue
        ok.sort(key=lambda p: p["kernel_ms"])
        out.append({"target_recall": t, "reached": True, "best": ok[0], "runners_up": ok[1:4]})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=8_800_000)
    ap.add_argument("--dim", type=int, default=30_000)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--sample", type=int, default=1000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--configs", default="2000:6,2000:3,3000:4,3000:6,4000:3,4000:6,6000:4")
    ap.add_argument("--query-cuts", default="2,3,4,5,6,8,10")
    ap.add_argument("--heap-factors", default="0.7,0.8,0.9,1.0")
    ap.add_argument("--first-sorted", default="0,1")
    ap.add_argument("--targets", default="0.90,0.95,0.99")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--query-batch", type=int, default=3, help="which 10 000-query set of the generator's stream (bench.py's first timed batch is 3)")
    ap.add_argument("--out", default="gpurun_out/operating_sweep.jsonl")
    ap.add_argument("--summarise", default="", help="only summarise an existing .jsonl")
    a = ap.parse_args()
    targets = [float(x) for x in a.targets.split(",") if x]
    if a.summarise:
        rows = [json.loads(l) for l in open(a.summarise) if l.strip().startswith("{")]
        gen = next((r for r in rows if "generated_s" in r), {})
        print(json.dumps({"docs": gen.get("docs"), "dim": gen.get("dim"), "targets": summarise(rows, targets),
                          "points": len([r for r in rows if "recall" in r]), "indexes": [r for r in rows if "index_built" in r]}, indent=1))
        return
    from seismic_amd import _native
    from seismic_amd._abi import BuildConfig
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    f = open(a.out, "a")

    def emit(obj):
        f.write(json.dumps(obj) + "\n")
        f.flush()
        print(json.dumps(obj), flush=True)

    t0 = time.time()
    docs = _native.synth(a.docs, a.dim, 42, 0)
    # the same stream of queries bench.py draws (seed 43): set `query_batch` of it
    allq = _native.synth(a.queries * (a.query_batch + 1), a.dim, 43, 1, docs)
    lo = a.query_batch * a.queries
    qo = allq[0]
    o0, o1 = int(qo[lo]), int(qo[lo + a.queries])
    q = ((qo[lo:lo + a.queries + 1] - qo[lo]).astype(np.uint64), allq[1][o0:o1], allq[2][o0:o1])
    ns = min(a.sample, a.queries)
    s_off = q[0][:ns + 1].copy()
    sq = (s_off, q[1][:int(s_off[ns])], q[2][:int(s_off[ns])])
    emit({"generated_s": time.time() - t0, "docs": a.docs, "dim": a.dim, "queries": a.queries, "sample": ns})
    exact = None
    rows = []
    for cfg_s in a.configs.split(","):
        parts = cfg_s.split(":")
        npost, mf = int(parts[0]), float(parts[1])
        cf = float(parts[2]) if len(parts) > 2 else 0.2
        se = float(parts[3]) if len(parts) > 3 else 0.5
        cfg = BuildConfig.defaults(n_postings=npost, centroid_fraction=cf, summary_energy=se, max_fraction=mf,
                                   min_cluster_size=2, doc_cut=15, use_device=1)
        t0 = time.time()
        ix = _native.NativeIndex.build(2, a.dim, *docs, cfg)
        t_build = time.time() - t0
        t0 = time.time()
        ix.upload(0)
        t_up = time.time() - t0
        d = ix.desc
        idx = {"n_postings": npost, "max_fraction": mf, "centroid_fraction": cf, "summary_energy": se}
        emit({"index_built": idx, "build_s": t_build, "upload_s": t_up, "hbm_bytes": ix.device_bytes(),
              "n_blocks": int(d.n_blocks), "postings_kept": int(d.n_postings), "summary_entries": int(d.n_entries)})
        if exact is None:
            t0 = time.time()
            _, ei, en = ix.exact_search(*sq, a.k)
            exact = [set(ei[i, :en[i]].tolist()) for i in range(ns)]
            emit({"exact_s": time.time() - t0})
        b = _native.DeviceBatch(ix, *q, a.k)
        for cut in [int(x) for x in a.query_cuts.split(",")]:
            for hf in [float(x) for x in a.heap_factors.split(",")]:
                for fs in [bool(int(x)) for x in a.first_sorted.split(",")]:
                    st = b.run(a.k, cut, hf, fs)
                    ms = min(b.run(a.k, cut, hf, fs).kernel_ms for _ in range(a.reps))
                    _, pid, pn = b.fetch(a.k)
                    rec = sum(len(set(pid[i, :pn[i]].tolist()) & exact[i]) for i in range(ns)) / float(ns * a.k)
                    row = {"index": idx, "query_cut": cut, "heap_factor": hf, "first_sorted": fs, "recall": rec,
                           "kernel_ms": float(ms), "qps_device_resident": a.queries / (ms * 1e-3),
                           "grid": int(st.grid), "lds_bytes": int(st.lds_bytes)}
                    rows.append(row)
                    emit(row)
        b.close()
        ix.close()
        del b, ix
    emit({"summary": summarise(rows, targets)})


if __name__ == "__main__":
    main()
