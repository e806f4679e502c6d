#!/usr/bin/env python3
"""bench.py — Seismic search hot path on MI355X: queries/sec at fixed recall@10.

A "step" is ONE pass of the search kernel over one batch of queries that is
already resident in HBM (index resident too). Workload (BASELINE.json configs[1]):
synthetic SPLADE-shape, 1M docs x 30K vocab x ~120 nnz/doc, 1K queries, k=10,
index params of best_configs/msmarco-v1/splade-v3/mem_budget_2.0/recall_95.toml
(n_postings=2000, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6,
min_cluster_size=2, doc_cut=15), query params query_cut=4, heap_factor=1.0,
first_sorted=false.

  python bench.py --gpus N --steps K --warmup W
N>1 is launched by the driver through torch.distributed.run (one rank per GPU):
the index is replicated, every rank searches its own batch of --queries queries
(weak scaling, no collective on the data path); value = total queries / max-over-ranks time.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=30_000)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--query-cut", type=int, default=4)
    ap.add_argument("--heap-factor", type=float, default=1.0)
    ap.add_argument("--first-sorted", type=int, default=0)
    ap.add_argument("--n-postings", type=int, default=2000)
    ap.add_argument("--centroid-fraction", type=float, default=0.2)
    ap.add_argument("--summary-energy", type=float, default=0.5)
    ap.add_argument("--max-fraction", type=float, default=6.0)
    ap.add_argument("--min-cluster-size", type=int, default=2)
    ap.add_argument("--comp-width", type=int, default=2, choices=[2, 4])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-recall", action="store_true", help="skip recall@k vs exact")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency measurement")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline time budget per mode")
    ap.add_argument("--index-cache", default=os.environ.get("SGPU_INDEX_CACHE", ""))
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch measured by a separate rocprofv3 --pmc pass (roofline.traffic)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from seismic_amd import _native
    from seismic_amd._abi import BuildConfig

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
    if not torch.cuda.is_available() or _native.device_count() < 1:
        raise SystemExit("no GPU visible: the search path has no CPU fallback")
    # SGPU_BENCH_BACKEND=gloo + SGPU_BENCH_ONE_DEVICE=1 lets the N>1 control flow be exercised on a
    # 1-GPU box (all ranks share device 0); the real multi-GPU run uses nccl (RCCL), one GPU per rank.
    backend = os.environ.get("SGPU_BENCH_BACKEND", "nccl")
    if os.environ.get("SGPU_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- data + index (rank 0 builds, the others load the file) ----------------
    cfg = BuildConfig.defaults(n_postings=args.n_postings, centroid_fraction=args.centroid_fraction,
                               summary_energy=args.summary_energy, max_fraction=args.max_fraction,
                               min_cluster_size=args.min_cluster_size, doc_cut=15)
    tag = "sgpu_%d_%d_cw%d_np%d_cf%g_se%g_mf%g_mc%d.idx" % (
        args.docs, args.dim, args.comp_width, args.n_postings, args.centroid_fraction, args.summary_energy,
        args.max_fraction, args.min_cluster_size)
    cache_dir = args.index_cache or tempfile.gettempdir()
    path = os.path.join(cache_dir, tag)
    t0 = time.time()
    docs = _native.synth(args.docs, args.dim, 42, 0)
    t_gen = time.time() - t0
    t_build = 0.0
    if rank == 0:
        index = None
        if os.path.exists(path):
            try:
                index = _native.NativeIndex.load(path)
            except _native.SeismicHipError:   # a stale / truncated file from an interrupted run
                index = None
        if index is None:
            t0 = time.time()
            index = _native.NativeIndex.build(args.comp_width, args.dim, *docs, cfg)
            t_build = time.time() - t0
            if world > 1 or args.index_cache:
                tmp = "%s.tmp.%d" % (path, os.getpid())
                index.save(tmp)
                os.replace(tmp, path)     # the other ranks only ever see a complete file
        log("[bench] docs generated in %.1fs, index built in %.1fs" % (t_gen, t_build))
    barrier()
    if rank != 0:
        index = _native.NativeIndex.load(path)
    t0 = time.time()
    index.upload(local_rank)
    t_up = time.time() - t0
    d = index.desc
    queries = _native.synth(args.queries, args.dim, 43 + 1000 * rank, 1, docs)
    q_off, q_comp, q_val = queries
    batch = _native.DeviceBatch(index, q_off, q_comp, q_val, args.k)

    def step(sync=False):
        return batch.run(args.k, args.query_cut, args.heap_factor, bool(args.first_sorted), sync=sync)

    # ---------------- timed region ----------------
    for _ in range(args.warmup):
        step()
    batch.sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    batch.sync_stats = batch.sync()   # waits for the K launches; mean kernel duration from HIP events
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(batch.sync_stats.kernel_ms)

    # ---------------- accounting (outside the timed region) ----------------
    gsc, gid, gn = batch.fetch(args.k)
    # one extra, untimed pass with the visited set materialised (sgpu_batch_run_counted): identical
    # results, and work counters that exclude re-encountered documents exactly as the reference does
    batch.run_counted(args.k, args.query_cut, args.heap_factor, bool(args.first_sorted))
    csc, cid, cn = batch.fetch(args.k)
    counted_identical = bool(np.array_equal(cn, gn) and np.array_equal(cid, gid)
                             and np.array_equal(csc.view(np.uint32), gsc.view(np.uint32)))
    algo_bytes, counters = batch.algorithmic_bytes(args.k, args.comp_width)
    total_q = args.queries * world
    ms_per_step = elapsed * 1e3 / args.steps
    qps = total_q * args.steps / elapsed
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    # roofline.traffic: HBM bytes per launch from a SEPARATE rocprofv3 --pmc pass (counters cannot be
    # collected from inside this process); taken from --traffic-bytes or profiles/pmc_traffic.json
    # when that measurement was made on this very workload, else null.
    traffic_bytes = args.traffic_bytes
    if traffic_bytes is None:
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = "docs=%d dim=%d queries=%d k=%d query_cut=%d heap_factor=%s first_sorted=%d" % (
                args.docs, args.dim, args.queries, args.k, args.query_cut, args.heap_factor, args.first_sorted)
            if pm.get("workload") == key:
                traffic_bytes = float(pm["traffic_bytes"])
        except (OSError, ValueError, KeyError):
            pass
    out = {
        "metric": "queries/sec + mean latency (\u00b5s) at fixed recall@10 vs exact, SPLADE-v3 MSMARCO",
        "value": qps,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "mean_latency_us_per_query_in_batch": ms_per_step * 1e3 / args.queries,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "Synthetic SPLADE-shape: %d docs, %d vocab, ~120 nnz/doc, %d queries/GPU, k=%d, 1xMI355X per rank"
                        % (args.docs, args.dim, args.queries, args.k),
            "index": {"n_postings": args.n_postings, "centroid_fraction": args.centroid_fraction,
                      "summary_energy": args.summary_energy, "max_fraction": args.max_fraction,
                      "min_cluster_size": args.min_cluster_size, "doc_cut": 15,
                      "hbm_bytes": index.device_bytes(), "n_blocks": int(d.n_blocks),
                      "n_postings_kept": int(d.n_postings), "summary_entries": int(d.n_entries)},
            "query": {"k": args.k, "query_cut": args.query_cut, "heap_factor": args.heap_factor,
                      "first_sorted": bool(args.first_sorted)},
            "storage": "f16 document values, u%d components, u8-quantised block summaries" % (8 * args.comp_width),
            "parallelism": "index replicated, %d query batch(es) of %d, no collective" % (world, args.queries),
            "launch": {"grid": int(batch.sync_stats.grid), "block": int(batch.sync_stats.block),
                       "lds_bytes": int(batch.sync_stats.lds_bytes)},
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic_bytes,
            "kernel": "seismic_search_kernel",
            "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": algo_bytes,
            "bytes_per_query": algo_bytes / max(args.queries, 1),
            "docs_scored_per_query": float(counters[:, 5].mean()) if len(counters) else 0.0,
            "docs_scored_speculatively_per_query": float(counters[:, 7].mean()) if len(counters) else 0.0,
            "summary_entries_per_query": float(counters[:, 2].mean()) if len(counters) else 0.0,
            "counted_pass_identical": counted_identical,
        },
        "timing_s": {"generate": t_gen, "build": t_build, "upload": t_up},
    }
    if rank == 0 and not args.no_latency:
        # mean latency of batch-1 searches (the reference's AQT: one query at a time,
        # src/bin/perf_inverted_index.rs:184-216): resident single-query batches, one synchronous
        # kernel pass each; wall time around launch + completion.
        nlat = min(200, args.queries)
        singles = [_native.DeviceBatch(index, np.array([0, q_off[i + 1] - q_off[i]], np.uint64),
                                       q_comp[q_off[i]:q_off[i + 1]], q_val[q_off[i]:q_off[i + 1]], args.k)
                   for i in range(nlat)]
        for sb in singles[:10]:
            sb.run(args.k, args.query_cut, args.heap_factor, bool(args.first_sorted), sync=True)
        t0 = time.perf_counter()
        kms = 0.0
        for sb in singles:
            kms += sb.run(args.k, args.query_cut, args.heap_factor, bool(args.first_sorted), sync=True).kernel_ms
        out["mean_latency_us_single_query"] = (time.perf_counter() - t0) * 1e6 / nlat
        out["mean_kernel_us_single_query"] = kms * 1e3 / nlat
        del singles
    if rank == 0 and not args.no_recall:
        t0 = time.time()
        es, ei, en = index.exact_search(q_off, q_comp, q_val, args.k)
        hits = 0
        for i in range(args.queries):
            hits += len(set(gid[i, :gn[i]].tolist()) & set(ei[i, :en[i]].tolist()))
        out["recall_at_k"] = hits / float(args.queries * args.k)
        out["timing_s"]["exact_ground_truth"] = time.time() - t0
    if rank == 0 and world == 1 and not args.no_cpu:
        # ---- cpu_baseline: the CPU oracle (a port; the Rust reference cannot be built here),
        # timed on this box's host cores on the same index + the same query batch.
        import orc
        ncores = os.cpu_count() or 1
        # single thread: the sequential loop of perf_inverted_index (bounded sample)
        osc, oid, on, ost, secs1, _ = orc.batch_search(d, q_off, q_comp, q_val, args.k, args.query_cut,
                                                       args.heap_factor, bool(args.first_sorted), num_threads=1)
        identical = bool(np.array_equal(on, gn) and np.array_equal(oid, gid)
                         and np.array_equal(osc.view(np.uint32), gsc.view(np.uint32)))
        runs1 = 1
        t_total = secs1
        while t_total < args.cpu_seconds / 2 and runs1 < 64:
            t_total += orc.batch_search(d, q_off, q_comp, q_val, args.k, args.query_cut, args.heap_factor,
                                        bool(args.first_sorted), num_threads=1)[4]
            runs1 += 1
        qps1 = runs1 * args.queries / t_total
        # many cores: one query per task (rayon global pool in batch_search). The best thread count
        # is searched (random 480-byte gathers stop scaling long before 256 hardware threads).
        sweep = {}
        cands = sorted({c for c in (0, ncores // 2, ncores // 4, 64, 32, 16) if c == 0 or 2 <= c <= ncores})
        for nt in cands:
            best = 0.0
            for rep in range(3):
                r = orc.batch_search(d, q_off, q_comp, q_val, args.k, args.query_cut, args.heap_factor,
                                     bool(args.first_sorted), num_threads=nt)
                if rep:
                    best = max(best, args.queries / r[4])
            sweep[int(r[5])] = best
        used = max(sweep, key=sweep.get)
        runs_n, t_n = 0, 0.0
        while (t_n < args.cpu_seconds / 2 and runs_n < 512) or runs_n < 2:
            r = orc.batch_search(d, q_off, q_comp, q_val, args.k, args.query_cut, args.heap_factor,
                                 bool(args.first_sorted), num_threads=used)
            if runs_n > 0:   # first run warms the per-thread scratch
                t_n += r[4]
            runs_n += 1
        qpsn = (runs_n - 1) * args.queries / t_n
        out["cpu_baseline"] = {
            "value": qpsn, "unit": "queries/s", "cores": int(used), "kind": "port",
            "sample": "the same %d-query batch, %d passes on %d threads (OpenMP, one query per task); "
                      "single thread: %d passes" % (args.queries, runs_n - 1, used, runs1),
            "single_thread_qps": qps1, "single_thread_us_per_query": 1e6 / qps1,
            "host_cores": ncores, "thread_sweep_qps": {str(k_): v_ for k_, v_ in sorted(sweep.items())},
            "gpu_results_identical_to_cpu": identical,
            "algorithmic_bytes_cpu": int(ost["algo_bytes"]),
        }
        out["gpu_over_cpu_allcore"] = qps / qpsn if qpsn > 0 else None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
