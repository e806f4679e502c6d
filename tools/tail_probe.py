#!/usr/bin/env python3
"""Mid-size calls of sgpu_batch_search (host buffers in and out) with and without the cooperative tail launch
(SGPU_TAIL_COOP = queries in the tail; abi.cpp search_shard): wall time per call, rows compared with the unsplit call."""
import os, sys, time
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8800000
docs = _native.synth(n, 30000, 42, 0)
path = os.path.join(os.environ.get("SGPU_INDEX_CACHE", "/tmp"), "lat_%d.idx" % n)
if os.path.exists(path):
    ix = _native.NativeIndex.load(path)
else:
    ix = _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(n_postings=2000, centroid_fraction=0.2, summary_energy=0.5,
                                                                          max_fraction=6.0, use_device=1))
    ix.save(path)
ix.upload(0)
NQ = 20000
q_off, qc, qv = _native.synth(NQ, 30000, 43, 1, docs)
for nq in (600, 1000, 1250, 2500):
    sets = []
    for r in range(min(8, NQ // nq)):
        lo, hi = r * nq, (r + 1) * nq
        sets.append(((q_off[lo:hi + 1] - q_off[lo]).astype(np.uint64), qc[q_off[lo]:q_off[hi]], qv[q_off[lo]:q_off[hi]]))
    outs = [(np.zeros((nq, 10), np.float32), np.zeros((nq, 10), np.uint64), np.zeros(nq, np.uint32)) for _ in sets]
    ref = None
    for tail in os.environ.get("TAILS", "0,64,128,192,256").split(","):
        os.environ["SGPU_TAIL_COOP"] = tail
        for i, s in enumerate(sets):
            ix.batch_search(*s, 10, 4, 1.0, False, out=outs[i])
        got = [tuple(a.copy() for a in o) for o in outs]
        if ref is None:
            ref = got
        same = all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for g, r_ in zip(got, ref) for a, b in zip(g, r_))
        ts = []
        for rep in range(6):
            for i, s in enumerate(sets):
                t0 = time.perf_counter()
                ix.batch_search(*s, 10, 4, 1.0, False, out=outs[i])
                ts.append((time.perf_counter() - t0) * 1e6)
        print("nq %5d  tail %3s: %7.1f us per call (mean), %7.1f (median), %.3f us/query  rows identical to the unsplit call: %s"
              % (nq, tail, np.mean(ts), np.median(ts), np.median(ts) / nq, same), flush=True)
