#!/usr/bin/env python3
"""A/B of the tuned CPU oracle on the GPU box's host: ORC_LOOKAHEAD=0/1 python tools/cpu_ab.py [n_docs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
import orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8800000
docs = _native.synth(n, 30000, 42, 0)
path = "/tmp/lat_%d.idx" % n
ix = _native.NativeIndex.load(path) if os.path.exists(path) else _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(
    n_postings=2000, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0, use_device=1))
if not os.path.exists(path):
    ix.save(path)
q = _native.synth(1000, 30000, 43, 1, docs)
d = ix.desc
for nt in (1, 16, 64):
    orc.batch_search(d, *q, 10, 4, 1.0, False, num_threads=nt, tuned=True)
    best = min(orc.batch_search(d, *q, 10, 4, 1.0, False, num_threads=nt, tuned=True)[4] for _ in range(3))
    print("ORC_LOOKAHEAD=%s threads %3d: %.1f us/query, %.0f queries/s" % (os.environ.get("ORC_LOOKAHEAD", "0"), nt, best / 1000 * 1e6, 1000 / best), flush=True)
