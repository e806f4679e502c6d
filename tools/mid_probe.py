"""Kernel time of mid-size launches (256 ... 2500 queries, or the sizes given) on the 8.8M-document shape; SGPU_COOP* knobs are honoured."""
import os, sys, time
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seismic_amd import _native
n=8800000
docs = _native.synth(n, 30000, 42, 0)
path = "/tmp/lat_%d.idx" % n
if os.path.exists(path):
    ix = _native.NativeIndex.load(path)
else:
    from seismic_amd._abi import BuildConfig
    ix = _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(n_postings=2000, centroid_fraction=0.2, summary_energy=0.5,
                                                                          max_fraction=6.0, use_device=1))
    ix.save(path)
ix.upload(0)
NQ=5000
q_off, qc, qv = _native.synth(NQ, 30000, 43, 1, docs)
SIZES = [int(x) for x in sys.argv[1:]] or [256, 1000, 1250, 2500]
for nq in SIZES:
    bs=[]
    for r in range(NQ//nq):
        lo,hi=r*nq,(r+1)*nq
        bs.append(_native.DeviceBatch(ix, q_off[lo:hi+1]-q_off[lo], qc[q_off[lo]:q_off[hi]], qv[q_off[lo]:q_off[hi]], 10))
    for b in bs[:2]: b.run(10,4,1.0,False)
    km=[]
    for rep in range(3):
        for b in bs: km.append(b.run(10,4,1.0,False).kernel_ms*1e3)
    print("nq=%4d kernel us mean %.1f min %.1f" % (nq, np.mean(km), np.min(km)), flush=True)
    del bs
