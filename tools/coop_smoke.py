#!/usr/bin/env python3
"""Small stand-alone check of the cooperative kernel variant against the oracle (no pytest capture)."""
import os, sys
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
from util import random_dataset, random_queries

mode = sys.argv[1] if len(sys.argv) > 1 else "force"
os.environ["SGPU_COOP"] = mode
os.environ.setdefault("SGPU_COOP_CHECK", "1")
dim = 700
off, comps, vals = random_dataset(5, 20000, dim, nnz_lo=8, nnz_hi=120)
ix = _native.NativeIndex.build(2, dim, off, comps, vals, BuildConfig.defaults(n_postings=2000, centroid_fraction=0.2,
                                                                             summary_energy=0.5, max_fraction=6.0))
ix.upload(0)
q = random_queries(6, int(sys.argv[2]) if len(sys.argv) > 2 else 40, dim, 5, 50)
print("searching, mode", mode, flush=True)
gs, gi, gn = ix.batch_search(*q, 10, 4, 1.0, False)
print("searched", flush=True)
os_, oi, on, st, _, _ = orc.batch_search(ix.desc, *q, 10, 4, 1.0, False)
ok = np.array_equal(gn, on) and np.array_equal(gi, oi) and np.array_equal(gs.view(np.uint32), os_.view(np.uint32))
print("identical to oracle:", ok, "docs scored/query", st["docs_scored"] / len(gn))
sys.exit(0 if ok else 1)
