#!/usr/bin/env python3
"""The tuned CPU oracle on the GPU box's host: thread sweep with the topology-aware pinning, before and after spreading the
index's pages over the NUMA nodes.   python tools/cpu_ab.py [n_docs]"""
import os, sys, time
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
import orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8800000
docs = _native.synth(n, 30000, 42, 0)
path = os.path.join(os.environ.get("SGPU_INDEX_CACHE", "/tmp"), "lat_%d.idx" % n)
ix = _native.NativeIndex.load(path) if os.path.exists(path) else _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(
    n_postings=2000, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0, use_device=int(os.environ.get("AB_DEVICE", "1"))))
if not os.path.exists(path):
    ix.save(path)
NQ = int(os.environ.get("AB_QUERIES", "1000"))
THREADS = [int(t) for t in os.environ.get("AB_THREADS", "1,16,32,64,128,256").split(",")]
q = _native.synth(NQ, 30000, 43, 1, docs)
d = ix.desc
ncpu = len(os.sched_getaffinity(0))
nodes = sum(1 for e in os.listdir("/sys/devices/system/node") if e.startswith("node") and e[4:].isdigit()) if os.path.isdir("/sys/devices/system/node") else 1
nplan = min(64, ncpu)
cores, plan = orc.pin_plan(nplan)
print("host: %d CPUs allowed, %d physical cores, %d NUMA nodes; %d pinned threads get CPUs %s" % (ncpu, cores, nodes, nplan, plan.tolist()), flush=True)


def cgroup():
    """cpu.max and the throttle counters of this process's cgroup (v2), when readable"""
    out = []
    for f in ("cpu.max", "cpu.stat", "cpu/cpu.cfs_quota_us", "cpu/cpu.cfs_period_us", "cpu/cpu.stat"):
        try:
            t = open("/sys/fs/cgroup/" + f).read().split("\n")
            out.append(f + "=" + ";".join(l for l in t if l and ("stat" not in f or "throttled" in l)))
        except OSError:
            pass
    return " ".join(out)


def sweep(tag):
    for nt in THREADS:
        if nt > ncpu:
            continue
        orc.batch_search(d, *q, 10, 4, 1.0, False, num_threads=nt, tuned=True)
        best = min(orc.batch_search(d, *q, 10, 4, 1.0, False, num_threads=nt, tuned=True)[4] for _ in range(3 if nt == 1 else 8))
        print("%s threads %3d: %.1f us/query, %.0f queries/s   %s" % (tag, nt, best / NQ * 1e6, NQ / best, cgroup()), flush=True)


sweep("as loaded  ")
used = orc.interleave_index(d)
print("interleave_index: %d nodes used" % used, flush=True)
if used:
    sweep("interleaved")
