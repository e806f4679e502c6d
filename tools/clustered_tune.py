#!/usr/bin/env python3
"""Recall and work per query of the clustered synthetic collection at the reference's recall_95 parameters, for a few
settings of its law (SGPU_SYNTH_CLUSTER = group,take,doc_sigma,query_sigma; a test hook).   python tools/clustered_tune.py [n_docs] setting ..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SGPU_TEST_HOOKS"] = "1"
from seismic_amd import _native
from seismic_amd._abi import BuildConfig

n = int(sys.argv[1])
for setting in sys.argv[2:]:
    os.environ["SGPU_SYNTH_CLUSTER"] = setting
    t = time.time()
    docs = _native.synth(n, 30000, 42, 0, collection=1)
    ix = _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(n_postings=2000, centroid_fraction=0.2, summary_energy=0.5,
                                                                          max_fraction=6.0, min_cluster_size=2, doc_cut=15, use_device=1))
    ix.upload(0)
    q = _native.synth(2000, 30000, 43, 1, docs, collection=1)
    b = _native.DeviceBatch(ix, *q, 10)
    b.run(10, 4, 1.0, False)
    ms = min(b.run(10, 4, 1.0, False).kernel_ms for _ in range(3))
    sc, ids, nn = b.fetch(10)
    b.run_counted(10, 4, 1.0, False)
    ab, st = b.algorithmic_bytes(10, 2, 2, None)
    ns = 300
    qo = q[0][:ns + 1]
    es, ei, en = ix.exact_search(qo, q[1][:int(qo[ns])], q[2][:int(qo[ns])], 10)
    rec = sum(len(set(ids[i, :nn[i]].tolist()) & set(ei[i, :en[i]].tolist())) for i in range(ns)) / (ns * 10.0)
    print("%s: recall@10 %.4f  docs/query %.0f  bytes/query %.0f  kernel ms per 2000 queries %.3f  (%.0f s)" % (
        setting, rec, st[:, 5].mean(), ab / 2000.0, ms, time.time() - t), flush=True)
    b.close()
    ix.close()
