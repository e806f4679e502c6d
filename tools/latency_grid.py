#!/usr/bin/env python3
"""Single-query latency (the reference's sequential loop, natively) against the number of resident workgroups of a
cooperative launch (SGPU_COOP_GRID; default: every slot of the chip).   python tools/latency_grid.py [n_docs] [n_queries]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seismic_amd import _native
from seismic_amd._abi import BuildConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_800_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 400
docs = _native.synth(n, 30000, 42, 0)
ix = _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(n_postings=2000, centroid_fraction=0.2, summary_energy=0.5,
                                                                      max_fraction=6.0, use_device=1))
ix.upload(0)
q = _native.synth(10000 * 4, 30000, 43, 1, docs)
lo = 30000
off = (q[0][lo:lo + nq + 1] - q[0][lo]).astype(np.uint64)
qc, qv = q[1][int(q[0][lo]):int(q[0][lo + nq])], q[2][int(q[0][lo]):int(q[0][lo + nq])]
want = ix.batch_search(off, qc, qv, 10, 4, 1.0, False)


def run(env):
    for k in ("SGPU_COOP_GRID", "SGPU_COOP_CHUNK", "SGPU_COOP_CHUNK_MIN"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ix.search_sequential(off[:21], qc, qv, 10, 4, 1.0, False)
    best = None
    for _ in range(3):
        sc, ids, cnt, us, ph = ix.search_sequential(off, qc, qv, 10, 4, 1.0, False)
        best = us if best is None else min(best, us)
    same = bool(np.array_equal(ids, want[1]) and np.array_equal(sc.view(np.uint32), want[0].view(np.uint32)))
    print("%-60s %7.1f us  rows identical %s" % (" ".join("%s=%s" % (k[5:], v) for k, v in sorted(env.items())) or "(defaults)", best, same), flush=True)


run({})
for g in (32, 64, 96, 128, 160, 192, 224, 256):
    run({"SGPU_COOP_GRID": str(g)})
for g, ch in ((128, 8), (128, 16), (64, 16), (64, 32), (192, 8)):
    run({"SGPU_COOP_GRID": str(g), "SGPU_COOP_CHUNK_MIN": str(ch)})
run({})
