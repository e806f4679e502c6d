#!/usr/bin/env python3
"""Reads tools/proxy_probe.py's dump: how well does anything the host knows before the launch predict a query's time in the
kernel, and what would a better launch order buy a 1250-query launch (list scheduling on 512 slots, simulated with the
measured per-query clocks)? Result of r05: profiles/r05_midsize_coop_study.md section 4.
usage: proxy_fit.py DUMP.npz"""
import sys, heapq
import numpy as np
z = np.load(sys.argv[1])
feat, big, mid = z["feat"], z["big"].astype(np.float64), z["mid"].astype(np.float64)
CUT = 4
def tot(st):
    cyc = st[:, 8:20] * 16
    return cyc[:, [0,1,2,3,4,5,6,7,8,9,11]].sum(1)
tb, tm = tot(big), tot(mid)
print("ms_big", z["ms_big"], "ms_mid", z["ms_mid"])
print("cycles big: mean %.0f max %.0f | mid: mean %.0f max %.0f (us @2.1: %.0f / %.0f)" % (tb.mean(), tb.max(), tm.mean(), tm.max(), tm.mean()/2100, tm.max()/2100))
np_ = feat[:, 0:CUT]; nb = feat[:, CUT:2*CUT]; w = feat[:, 2*CUT:3*CUT]; nnz = feat[:, 3*CUT]; sw = feat[:, 3*CUT+1]
proxy0 = np_.sum(1)
def sim(order, dur, slots=512):
    h = [0.0]*slots; heapq.heapify(h)
    end = 0
    for q in order:
        t = heapq.heappop(h) + dur[q]
        end = max(end, t); heapq.heappush(h, t)
    return end
def report(name, score, t, lo, hi):
    # per 1250-launch makespan in us
    ms = []
    for r in range(8):
        idx = np.arange(r*1250, (r+1)*1250)
        o = idx[np.argsort(-score[idx], kind="stable")]
        ms.append(sim(o, t) / 2100)
    print("%-28s corr big %.3f mid %.3f | simulated 1250 makespan us: mean %.0f  %s" % (name, np.corrcoef(score, tb)[0,1], np.corrcoef(score, tm)[0,1], np.mean(ms), [int(x) for x in ms]))
report("sum np (product)", proxy0, tm, 0, 0)
report("oracle (true mid cycles)", tm, tm, 0, 0)
report("true big cycles", tb, tm, 0, 0)
report("input order", -np.arange(10000.0), tm, 0, 0)
for r in range(8):
    idx = np.arange(r*1250, (r+1)*1250)
print("lower bounds per launch us: mean-load %s  max-query %s" % ([int(tm[r*1250:(r+1)*1250].sum()/512/2100) for r in range(8)], [int(tm[r*1250:(r+1)*1250].max()/2100) for r in range(8)]))
# candidate features
X = {
 "sum nb": nb.sum(1),
 "nnz": nnz,
 "sum w*np": (w*np_).sum(1),
 "sum w*np / w1": (w*np_).sum(1)/np.maximum(w[:,0],1e-9),
 "np1": np_[:,0], "np2": np_[:,1], "np3": np_[:,2], "np4": np_[:,3],
 "sw": sw, "w1": w[:,0], "w4/w1": w[:,3]/np.maximum(w[:,0],1e-9),
 "sw/w1": sw/np.maximum(w[:,0],1e-9),
}
for k,v in X.items():
    print("  corr(%-14s, mid cycles) = %.3f   big %.3f" % (k, np.corrcoef(v, tm)[0,1], np.corrcoef(v, tb)[0,1]))
# linear fit on train half, test on the other
def fit(cols, name, target=tm):
    A = np.column_stack(cols + [np.ones(len(target))])
    tr = np.arange(len(target)) % 2 == 0
    coef, *_ = np.linalg.lstsq(A[tr], target[tr], rcond=None)
    pred = A @ coef
    print("fit %-40s test corr %.3f coef %s" % (name, np.corrcoef(pred[~tr], target[~tr])[0,1], np.array2string(coef, precision=3)))
    return pred
p1 = fit([np_[:,i] for i in range(4)], "np_i"); report("fit np_i", p1, tm, 0, 0)
p2 = fit([np_[:,i] for i in range(4)] + [nb[:,i] for i in range(4)], "np_i nb_i"); report("fit np_i nb_i", p2, tm, 0, 0)
p3 = fit([np_[:,i] for i in range(4)] + [nb.sum(1), nnz, sw, w[:,0]], "np_i sumnb nnz sw w1"); report("fit +nnz sw w1", p3, tm, 0, 0)
p4 = fit([np_[:,i] for i in range(4)] + [nb.sum(1), nnz, sw/np.maximum(w[:,0],1e-9), w[:,3]/np.maximum(w[:,0],1e-9), np_[:,0]*w[:,1]/np.maximum(w[:,0],1e-9)], "richer"); report("fit richer", p4, tm, 0, 0)
# work counters as an upper bound on what any a-priori proxy can reach
spec = mid[:,7]; ent = mid[:,2]; blocks = mid[:,0]
print("corr(spec docs, cycles) %.3f  corr(entries, cycles) %.3f corr(blocks, cycles) %.3f" % (np.corrcoef(spec, tm)[0,1], np.corrcoef(ent, tm)[0,1], np.corrcoef(blocks, tm)[0,1]))
for k,v in X.items():
    print("  corr(%-14s, spec docs) = %.3f" % (k, np.corrcoef(v, spec)[0,1]))
from sklearn.ensemble import GradientBoostingRegressor
F = np.column_stack([np_, nb, w, nnz, sw])
tr = np.arange(10000) % 2 == 0
g = GradientBoostingRegressor(n_estimators=300, max_depth=4, learning_rate=0.05, subsample=0.8, random_state=0).fit(F[tr], tm[tr])
pg = g.predict(F)
print("GBM test corr %.3f" % np.corrcoef(pg[~tr], tm[~tr])[0,1])
pg2 = pg.copy(); pg2[tr] = -1e18
# simulate on launches using only test queries' predictions is awkward; report with all predictions (train half optimistic)
report("GBM (half in-sample)", pg, tm, 0, 0)
print("feature importances", np.round(g.feature_importances_, 3))
