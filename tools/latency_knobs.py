#!/usr/bin/env python3
"""Single-query latency (the reference's sequential loop, natively) against the cooperative variant's knobs, one
index build, knobs changed in-process (the library re-reads SGPU_* when the environment changes).
  python tools/latency_knobs.py [n_docs] [n_queries] [n_postings] [max_fraction] [query_cut] [short]
(short: only the handful of combinations around the defaults)"""
import itertools, os, sys
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seismic_amd import _native
from seismic_amd._abi import BuildConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_800_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 400
npost = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
mf = float(sys.argv[4]) if len(sys.argv) > 4 else 6.0
qcut = int(sys.argv[5]) if len(sys.argv) > 5 else 4
short = len(sys.argv) > 6
docs = _native.synth(n, 30000, 42, 0)
ix = _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(n_postings=npost, centroid_fraction=0.2, summary_energy=0.5,
                                                                      max_fraction=mf, use_device=1))
ix.upload(0)
q = _native.synth(10000 * 4, 30000, 43, 1, docs)
lo = 30000
off = (q[0][lo:lo + nq + 1] - q[0][lo]).astype(np.uint64)
qc, qv = q[1][int(q[0][lo]):int(q[0][lo + nq])], q[2][int(q[0][lo]):int(q[0][lo + nq])]
want = ix.batch_search(off, qc, qv, 10, qcut, 1.0, False)


def run(env):
    for k in list(os.environ):
        if k.startswith("SGPU_COOP_"):
            del os.environ[k]
    os.environ.update(env)
    ix.search_sequential(off[:21], qc, qv, 10, qcut, 1.0, False)
    best = None
    for _ in range(3):
        sc, ids, cnt, us, ph = ix.search_sequential(off, qc, qv, 10, qcut, 1.0, False)
        best = us if best is None else min(best, us)
    same = bool(np.array_equal(ids, want[1]) and np.array_equal(sc.view(np.uint32), want[0].view(np.uint32)))
    print("%-90s %7.1f us  rows identical %s" % (" ".join("%s=%s" % (k[10:], v) for k, v in sorted(env.items())) or "(defaults)", best, same), flush=True)


if short:
    for env in ({}, {"SGPU_COOP_FIRST_REACH": "128"}, {"SGPU_COOP_CHUNK_MIN": "8"}, {"SGPU_COOP_FIRST_REACH": "128", "SGPU_COOP_CHUNK_MIN": "8"},
                {"SGPU_COOP_FIRST_REACH": "192"}, {"SGPU_COOP_FIRST_REACH": "128", "SGPU_COOP_ITEMS_INIT": "256"}, {}):
        run(env)
    sys.exit(0)
run({})
for ii, fr in itertools.product((64, 128, 256, 512), (128, 256, 512, 1024, 100000)):
    run({"SGPU_COOP_ITEMS_INIT": str(ii), "SGPU_COOP_FIRST_REACH": str(fr)})
for mi in (0, 32, 128, 256):
    run({"SGPU_COOP_MIN_ITEMS": str(mi)})
for ch in (2, 8, 16, 32):
    run({"SGPU_COOP_CHUNK_MIN": str(ch)})
for mc in (512, 1024):
    run({"SGPU_COOP_MAX_CAND": str(mc), "SGPU_COOP_FIRST_REACH": "100000", "SGPU_COOP_ITEMS_INIT": "512"})
run({})
