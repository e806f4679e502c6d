#!/usr/bin/env python3
"""Batch-1 and small-batch latency of the search kernel (wall time per synchronous pass)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
docs = _native.synth(n, 30000, 42, 0)
path = "/tmp/lat_%d.idx" % n
if os.path.exists(path):
    ix = _native.NativeIndex.load(path)
else:
    ix = _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(n_postings=max(1, 2000 * n // 1000000),
                                                                          centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0))
    ix.save(path)
ix.upload(0)
q_off, qc, qv = _native.synth(1000, 30000, 43, 1, docs)
for nq in (1, 8, 64, 256, 1000):
    reps = 200 if nq == 1 else 20
    batches = []
    for r in range(min(reps, 1000 // nq)):
        lo, hi = r * nq, (r + 1) * nq
        batches.append(_native.DeviceBatch(ix, q_off[lo:hi + 1] - q_off[lo], qc[q_off[lo]:q_off[hi]], qv[q_off[lo]:q_off[hi]], 10))
    for b in batches[:3]:
        b.run(10, 4, 1.0, False)
    t = time.perf_counter()
    km = 0.0
    for b in batches:
        km += b.run(10, 4, 1.0, False).kernel_ms
    dt = (time.perf_counter() - t) / len(batches)
    print("nq=%4d: %.1f us wall per pass, %.1f us kernel, %.2f us/query  (env %s)" % (
        nq, dt * 1e6, km * 1e3 / len(batches), dt * 1e6 / nq, {k: v for k, v in os.environ.items() if k.startswith("SGPU_")}))
# the host-buffer entry point (sgpu_search): H2D of the query, kernel pass, D2H of the results
t = time.perf_counter()
for i in range(200):
    ix.search(qc[q_off[i]:q_off[i + 1]], qv[q_off[i]:q_off[i + 1]], 10, 4, 1.0, False)
print("sgpu_search (host buffers in/out): %.1f us per query" % ((time.perf_counter() - t) / 200 * 1e6))
