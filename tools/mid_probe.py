import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from seismic_amd import _native
n=8800000
docs = _native.synth(n, 30000, 42, 0)
ix = _native.NativeIndex.load("/tmp/lat_%d.idx" % n)
ix.upload(0)
NQ=5000
q_off, qc, qv = _native.synth(NQ, 30000, 43, 1, docs)
for nq in (256, 1000, 1250, 2500):
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
