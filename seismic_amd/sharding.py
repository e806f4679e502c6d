"""Multi-GPU = replicas: the index is replicated in every GPU's HBM, a query batch is cut into
contiguous shards, one per rank (one process per GPU), and the per-rank result slabs are
concatenated in input order. Queries are independent (reference src/pylib/mod.rs:629-652 runs
them as independent rayon tasks on a shared &self index), so there is NO collective on the
data path; the only communication is the final gather of (score, id, count) rows.
"""
import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous, balanced [lo, hi) of `n` items for `rank` of `world`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_csr(q_off, comps, vals, world, rank):
    """The CSR sub-batch of `rank`."""
    q_off = np.asarray(q_off, np.uint64)
    lo, hi = shard_bounds(len(q_off) - 1, world, rank)
    s, e = int(q_off[lo]), int(q_off[hi])
    return (q_off[lo:hi + 1] - q_off[lo]).astype(np.uint64), np.asarray(comps)[s:e], np.asarray(vals)[s:e]


def gather_rows(sc, ids, n, nq, group=None):
    """All ranks contribute the result rows of their contiguous shard of an `nq`-query batch and
    receive the full (scores[nq,k], ids[nq,k], n[nq]) in input order. This gather of result rows
    is the only communication of the multi-GPU path (nccl == RCCL on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [shard_bounds(nq, world, r)[1] - shard_bounds(nq, world, r)[0] for r in range(world)]
    mx = max(sizes) if sizes else 0
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")

    def gather(a, dtype):
        pad = np.zeros((mx,) + a.shape[1:], a.dtype)
        pad[: len(a)] = a
        t = torch.from_numpy(pad.view(dtype)).to(dev)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        parts = [o.cpu().numpy().view(a.dtype)[: sizes[r]] for r, o in enumerate(outs)]
        return np.concatenate(parts) if parts else pad[:0]

    return (gather(np.ascontiguousarray(sc, np.float32), np.float32),
            gather(np.ascontiguousarray(ids, np.uint64).view(np.int64), np.int64).view(np.uint64),
            gather(np.ascontiguousarray(n, np.uint32).view(np.int32), np.int32).view(np.uint32))


def batch_search_sharded(search_fn, q_off, comps, vals, k, group=None):
    """Every rank searches its shard with `search_fn(q_off, comps, vals) -> (scores[nq,k], ids[nq,k], n[nq])`
    and all ranks receive the full result in input order. Uses torch.distributed when it is
    initialised (backend nccl == RCCL on GPUs, gloo on CPU); otherwise a single shard."""
    try:
        import torch.distributed as dist
        active = dist.is_available() and dist.is_initialized()
    except ImportError:
        active = False
    if not active:
        return search_fn(np.asarray(q_off, np.uint64), comps, vals)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sc, ids, n = search_fn(*shard_csr(q_off, comps, vals, world, rank))
    return gather_rows(sc, ids, n, len(q_off) - 1, group)
