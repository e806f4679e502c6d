"""Throughput of sgpu_batch_search (host buffers in and out) from one and two request threads; SGPU_CHUNK_* knobs are honoured."""
import os, sys, time, threading
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import numpy as np
sys.path.insert(0,'/root/repo')
from seismic_amd import _native
n=8800000
docs=_native.synth(n,30000,42,0)
path="/tmp/lat_%d.idx"%n
if os.path.exists(path):
    ix=_native.NativeIndex.load(path)
else:
    from seismic_amd._abi import BuildConfig
    ix=_native.NativeIndex.build(2,30000,*docs,BuildConfig.defaults(n_postings=2000,centroid_fraction=0.2,summary_energy=0.5,max_fraction=6.0,use_device=1)); ix.save(path)
ix.upload(0)
NB=int(os.environ.get('E2E_BATCHES','6'))   # (24: the batches are cold in the host's caches, as in the bench line's leg)
q_off,qc,qv=_native.synth(10000*NB,30000,43,1,docs)
hb=[]
for r in range(NB):
    lo,hi=r*10000,(r+1)*10000
    hb.append(((q_off[lo:hi+1]-q_off[lo]).astype(np.uint64), qc[q_off[lo]:q_off[hi]], qv[q_off[lo]:q_off[hi]]))
outs=[(np.zeros((10000,10),np.float32),np.zeros((10000,10),np.uint64),np.zeros(10000,np.uint32)) for _ in range(NB)]
def call(i): ix.batch_search(*hb[i%NB],10,4,1.0,False,out=outs[i%NB])
for nt in (1,2):
    for i in range(4): call(i)
    K=24
    def w(t):
        for i in range(t,K,nt): call(i)
    th=[threading.Thread(target=w,args=(t,)) for t in range(nt)]
    t0=time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt=time.perf_counter()-t0
    print("env",{k:v for k,v in os.environ.items() if k.startswith("SGPU_")},"threads",nt,"qps %.0f"%(K*10000/dt),flush=True)
