#!/usr/bin/env python3
"""How well does the launch order's a-priori cost predict a query's time? Per-query phase clocks of a 1250-query launch
(profiling build) against host-side features; simulated longest-first packing on the launch's workgroup slots under each
ordering. Usage: SGPU_COOP=0 python tools/lpt_study.py [n_docs] [n_queries]"""
import os, sys
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SGPU_LIB", os.path.join(ROOT, "seismic_amd", "libseismic_hip_prof.so"))
os.environ.setdefault("SGPU_COOP", "0")
import orc
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8800000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1250
docs = _native.synth(n, 30000, 42, 0)
path = "/tmp/lat_%d.idx" % n
ix = _native.NativeIndex.load(path) if os.path.exists(path) else _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(
    n_postings=2000, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0, use_device=1))
ix.upload(0)
q_off, qc, qv = _native.synth(nq, 30000, 43, 1, docs)
b = _native.DeviceBatch(ix, q_off, qc, qv, 10)
for _ in range(2):
    b.run(10, 4, 1.0, False)
ms = min(b.run(10, 4, 1.0, False).kernel_ms for _ in range(3))
st = b.fetch_stats().astype(np.float64)
t = (st[:, 8:20].sum(axis=1) - st[:, 18]) * 16.0     # shader cycles per query ([18] is a round counter)
slots = len(set(st[:, 20].astype(int)))
a = orc.desc_arrays(ix.desc)
lb, bp = a["list_block_start"].astype(np.int64), a["block_post_start"].astype(np.int64)
lrs, rp = a["list_row_start"].astype(np.int64), a["row_ptr"].astype(np.int64)
feat = {"postings": [], "blocks": [], "nnz": [], "entries_of_lists": [], "postings x w/w1": [], "postings x (w/w1)^2": [],
        "postings of list 1": [], "postings / sum(w top4)": [], "postings x w/sum(w)": [], "sum(w) all": [], "w1": []}
for i in range(nq):
    c, v = qc[q_off[i]:q_off[i + 1]], qv[q_off[i]:q_off[i + 1]]
    top = c[np.argsort(-v, kind="stable")[:4]]
    feat["postings"].append(sum(bp[lb[x + 1]] - bp[lb[x]] for x in top))
    feat["blocks"].append(sum(lb[x + 1] - lb[x] for x in top))
    feat["nnz"].append(len(c))
    feat["entries_of_lists"].append(sum(rp[lrs[x + 1]] - rp[lrs[x]] for x in top))
    wt = np.sort(v)[::-1][:4].astype(np.float64)
    npl = np.array([bp[lb[x + 1]] - bp[lb[x]] for x in top], np.float64)
    feat["postings x w/w1"].append(float((npl * wt / wt[0]).sum()))
    feat["postings x (w/w1)^2"].append(float((npl * (wt / wt[0]) ** 2).sum()))
    feat["postings of list 1"].append(float(npl[0]))
    feat["postings / sum(w top4)"].append(float(npl.sum() / wt.sum()))
    feat["postings x w/sum(w)"].append(float((npl * wt).sum() / float(v.sum())))
    feat["sum(w) all"].append(float(v.sum()))
    feat["w1"].append(float(wt[0]))
F = {k: np.array(v, np.float64) for k, v in feat.items()}
F["docs_scored (a posteriori)"] = st[:, 7]
F["postings + 40 nnz"] = F["postings"] + 40 * F["nnz"]


def makespan(order):
    load = np.zeros(slots)
    for i in order:
        j = load.argmin()
        load[j] += t[i]
    return load.max()


print("launch: %d queries, %d slots, kernel %.1f us; sum of query cycles / slots = %.0f cycles, longest query %.0f" % (
    nq, slots, ms * 1e3, t.sum() / slots, t.max()))
ideal = makespan(np.argsort(-t))
print("%-28s corr %.3f  packed makespan / (sum/slots) = %.3f" % ("true time (ideal LPT)", 1.0, ideal / (t.sum() / slots)))
for k, f in F.items():
    print("%-28s corr %.3f  packed makespan / (sum/slots) = %.3f" % (k, np.corrcoef(f, t)[0, 1], makespan(np.argsort(-f, kind="stable")) / (t.sum() / slots)))
print("%-28s            packed makespan / (sum/slots) = %.3f" % ("input order", makespan(np.arange(nq)) / (t.sum() / slots)))
X = np.stack([F["postings"], F["blocks"], F["nnz"], F["entries_of_lists"], np.ones(nq)], axis=1)
w, *_ = np.linalg.lstsq(X, t, rcond=None)
fit = X @ w
print("least squares on (postings, blocks, nnz, entries, 1): weights %s corr %.3f makespan ratio %.3f" % (
    np.round(w, 2), np.corrcoef(fit, t)[0, 1], makespan(np.argsort(-fit)) / (t.sum() / slots)))
