#!/usr/bin/env python3
"""Batch-1 and small-batch latency of the search path, with the host-side breakdown of sgpu_search.

  python tools/latency_probe.py [n_docs]        (env SGPU_WAIT=block|spin, SGPU_COOP=0|1 ... are honoured)

Prints, for the MS MARCO-shaped synthetic collection of n_docs documents:
  * wall time and kernel time (HIP events) per synchronous device-resident pass of 1 ... 1000 queries
  * the reference's sequential loop (sgpu_search_sequential = one sgpu_search per query) natively, with
    the mean microseconds per query spent in every host-side phase
  * the same loop through the Python binding (what a PyO3-style caller pays on top)
"""
import json, os, sys, time
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seismic_amd import _native
from seismic_amd._abi import BuildConfig

PHASES = ["validate+plan", "staging", "enqueue_h2d", "configure+launch", "enqueue_d2h", "wait", "copy_out", "-"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    docs = _native.synth(n, 30000, 42, 0)
    path = os.path.join(os.environ.get("SGPU_INDEX_CACHE", "/tmp"), "lat_%d.idx" % n)
    if os.path.exists(path):
        ix = _native.NativeIndex.load(path)
    else:
        ix = _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(
            n_postings=2000, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0, use_device=1))
        ix.save(path)
    ix.upload(0)
    NQ = 5000
    q_off, qc, qv = _native.synth(NQ, 30000, 43, 1, docs)
    env = {k: v for k, v in os.environ.items() if k.startswith("SGPU_")}
    out = {"n_docs": n, "env": env, "passes": {}}
    sizes = [int(x) for x in os.environ.get("LATENCY_PROBE_SIZES", "1,8,64,256,1000,1250").split(",")]
    for nq in sizes:
        reps = 200 if nq == 1 else (20 if nq < 1000 else 4)
        batches = []
        for r in range(min(reps, NQ // nq)):
            lo, hi = r * nq, (r + 1) * nq
            batches.append(_native.DeviceBatch(ix, q_off[lo:hi + 1] - q_off[lo], qc[q_off[lo]:q_off[hi]], qv[q_off[lo]:q_off[hi]], 10))
        for b in batches[:3]:
            b.run(10, 4, 1.0, False)
        t = time.perf_counter()
        km = 0.0
        for b in batches:
            km += b.run(10, 4, 1.0, False).kernel_ms
        dt = (time.perf_counter() - t) / len(batches)
        out["passes"][nq] = {"wall_us": dt * 1e6, "kernel_us": km * 1e3 / len(batches), "us_per_query": dt * 1e6 / nq}
        print("nq=%4d: %.1f us wall per pass, %.1f us kernel, %.2f us/query" % (nq, dt * 1e6, km * 1e3 / len(batches), dt * 1e6 / nq))
        del batches
    if os.environ.get("LATENCY_PROBE_SIZES"):
        return
    nl = 200
    lo = q_off[:nl + 1]
    ix.search_sequential(lo[:11], qc, qv, 10, 4, 1.0, False)          # warm-up (grows the lane's arena)
    for rep in range(2):
        _, _, _, mean, ph = ix.search_sequential(lo, qc, qv, 10, 4, 1.0, False)
    out["sequential_native_us"] = mean
    out["breakdown_us"] = {PHASES[i]: float(ph[i]) for i in range(7)}
    print("sgpu_search_sequential: %.1f us per query;  %s;  unaccounted %.1f" % (
        mean, "  ".join("%s %.1f" % (PHASES[i], ph[i]) for i in range(7)), mean - ph[:7].sum()))
    t = time.perf_counter()
    for i in range(nl):
        ix.search(qc[q_off[i]:q_off[i + 1]], qv[q_off[i]:q_off[i + 1]], 10, 4, 1.0, False)
    out["python_binding_us"] = (time.perf_counter() - t) / nl * 1e6
    print("sgpu_search through the Python binding: %.1f us per query" % out["python_binding_us"])
    # after a pause the host may have dropped into a deeper idle state: the first calls show the wake-up cost
    time.sleep(0.5)
    _, _, _, cold, _ = ix.search_sequential(lo[:6], qc, qv, 10, 4, 1.0, False)
    out["after_500ms_idle_us"] = cold
    print("first 5 calls after 0.5 s of idling: %.1f us per query" % cold)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
