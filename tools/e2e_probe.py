import os, sys, time, threading
import numpy as np
sys.path.insert(0,'/root/repo')
from seismic_amd import _native
n=8800000
docs=_native.synth(n,30000,42,0)
ix=_native.NativeIndex.load("/tmp/lat_%d.idx"%n); ix.upload(0)
q_off,qc,qv=_native.synth(60000,30000,43,1,docs)
hb=[]
for r in range(6):
    lo,hi=r*10000,(r+1)*10000
    hb.append(((q_off[lo:hi+1]-q_off[lo]).astype(np.uint64), qc[q_off[lo]:q_off[hi]], qv[q_off[lo]:q_off[hi]]))
outs=[(np.zeros((10000,10),np.float32),np.zeros((10000,10),np.uint64),np.zeros(10000,np.uint32)) for _ in range(6)]
def call(i): ix.batch_search(*hb[i%6],10,4,1.0,False,out=outs[i%6])
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
