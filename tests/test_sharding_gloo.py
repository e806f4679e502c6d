"""N>1 path on CPU: 2 ranks over gloo shard a query batch (index replicated), search their
shards and gather; the result must equal the single-process answer, in input order.
The per-shard search executor is the CPU oracle here (tests may use it; the product's
executor is the GPU path, which bench.py drives the same way)."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    import orc
    from seismic_amd import _native
    from seismic_amd._abi import BuildConfig
    from seismic_amd.sharding import batch_search_sharded
    from util import random_dataset, random_queries

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dim = 256
    off, comps, vals = random_dataset(41, 1500, dim, nnz_lo=5, nnz_hi=60)
    ix = _native.NativeIndex.build(2, dim, off, comps, vals, BuildConfig.defaults(n_postings=40, num_threads=2))
    q = random_queries(42, 37, dim, 3, 40)   # 37: uneven shards

    def search_fn(q_off, c, v):
        s, i, n, _, _, _ = orc.batch_search(ix.desc, q_off, c, v, 10, 5, 0.8, True, num_threads=1)
        return s, i, n

    sc, ids, n = batch_search_sharded(search_fn, *q, 10)
    if rank == 0:
        s1, i1, n1 = search_fn(*q)
        ok = np.array_equal(n, n1) and np.array_equal(ids, i1) and np.array_equal(sc.view(np.uint32), s1.view(np.uint32))
        open(out_path, "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.txt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_shard_bounds_cover_everything():
    from seismic_amd.sharding import shard_bounds, shard_csr
    for n in (0, 1, 7, 1000):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
    q_off = np.array([0, 2, 2, 5, 9], np.uint64)
    o, c, v = shard_csr(q_off, np.arange(9), np.arange(9.0), 2, 1)
    assert o.tolist() == [0, 3, 7] and c.tolist() == [2, 3, 4, 5, 6, 7, 8]
