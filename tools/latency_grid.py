#!/usr/bin/env python3
"""Latency of small launches against the number of resident workgroups of a cooperative launch (SGPU_COOP_GRID: unset =
the library's rule, 0 = every slot of the chip). nq = 1: the reference's sequential loop, natively; nq > 1: `sgpu_batch_search` calls of nq queries
through the binding (wall time per call, best of three passes).
  python tools/latency_grid.py [n_docs] [n_postings] [max_fraction] [query_cut]"""
import os, sys, time
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seismic_amd import _native
from seismic_amd._abi import BuildConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_800_000
npost = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
mf = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
qcut = int(sys.argv[4]) if len(sys.argv) > 4 else 4
docs = _native.synth(n, 30000, 42, 0)
ix = _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(n_postings=npost, centroid_fraction=0.2, summary_energy=0.5,
                                                                      max_fraction=mf, use_device=1))
ix.upload(0)
q = _native.synth(10000 * 4, 30000, 43, 1, docs)
lo = 30000


def subset(a, b):
    off = (q[0][a:b + 1] - q[0][a]).astype(np.uint64)
    return off, q[1][int(q[0][a]):int(q[0][b])], q[2][int(q[0][a]):int(q[0][b])]


def run(nq, grid):
    def knob(g):   # None: the library's rule; 0: every slot of the chip; n: at most n workgroups
        if g is None:
            os.environ.pop("SGPU_COOP_GRID", None)
        else:
            os.environ["SGPU_COOP_GRID"] = str(g)
    knob(grid)
    if nq == 1:
        off, qc, qv = subset(lo, lo + 400)
        want = ix.batch_search(off, qc, qv, 10, qcut, 1.0, False)
        ix.search_sequential(off[:21], qc, qv, 10, qcut, 1.0, False)
        best = None
        for _ in range(3):
            sc, ids, cnt, us, ph = ix.search_sequential(off, qc, qv, 10, qcut, 1.0, False)
            best = us if best is None else min(best, us)
        same = bool(np.array_equal(ids, want[1]) and np.array_equal(sc.view(np.uint32), want[0].view(np.uint32)))
    else:
        sets = [subset(lo + i * nq, lo + (i + 1) * nq) for i in range(40)]
        knob(0)
        want = [ix.batch_search(*s, 10, qcut, 1.0, False) for s in sets[:3]]
        knob(grid)
        got = [ix.batch_search(*s, 10, qcut, 1.0, False) for s in sets[:3]]
        same = all(np.array_equal(g[1], w[1]) and np.array_equal(g[0].view(np.uint32), w[0].view(np.uint32)) for g, w in zip(got, want))
        best = None
        for _ in range(3):
            t = time.perf_counter()
            for s in sets:
                ix.batch_search(*s, 10, qcut, 1.0, False)
            us = (time.perf_counter() - t) * 1e6 / len(sets)
            best = us if best is None else min(best, us)
    print("nq %4d  grid %5s  %8.1f us per call  rows identical %s" % (nq, "rule" if grid is None else (grid or "all"), best, same), flush=True)


for nq, grids in ((1, (None, 48, 64, 80, 96, 128, 0)), (2, (None, 64, 96, 128, 0)), (4, (None, 128, 176, 0)), (8, (None, 64, 96, 128, 192, 0)),
                  (32, (None, 96, 128, 192, 0)), (64, (None, 128, 192, 0)), (128, (None, 192, 0))):
    for g in grids:
        run(nq, g)
