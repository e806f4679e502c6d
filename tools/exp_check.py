"""Rows of 2000 queries on the cached 8.8M-document index, saved to argv[1] (.npz): to compare an experiment build with the product."""
import os, sys
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seismic_amd import _native
n = 8800000
docs = _native.synth(n, 30000, 42, 0)
ix = _native.NativeIndex.load("/tmp/lat_%d.idx" % n)
if len(sys.argv) > 2:
    ix = ix.convert(int(sys.argv[2]))
ix.upload(0)
q = _native.synth(2000, 30000, 43, 1, docs)
sc, ids, nn = ix.batch_search(*q, 10, 4, 1.0, False)
np.savez(sys.argv[1], sc=sc.view(np.uint32), ids=ids, nn=nn)
