"""bench.py's N>1 control flow on a 1-GPU box: 2 ranks share device 0 (SGPU_BENCH_ONE_DEVICE=1), gloo
instead of RCCL for the barrier / max-over-ranks / result gather. Strong scaling = BASELINE config 4's
shape (ONE batch cut into contiguous shards, index replicated): the gathered rows must be the 1-GPU
answer. Run with `-m gpu`."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(scaling, tmp_path, **extra_env):
    env = dict(os.environ, SGPU_BENCH_ONE_DEVICE="1", SGPU_BENCH_BACKEND="gloo", SGPU_INDEX_CACHE=str(tmp_path),
               MASTER_ADDR="127.0.0.1", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
           "3", "--warmup", "1", "--docs", "60000", "--queries", "1501", "--n-postings", "300", "--scaling", scaling,
           "--no-cpu", "--no-recall", "--no-latency", "--no-e2e"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_two_ranks_strong_scaling_is_the_single_gpu_answer(tmp_path):
    out = _run("strong", tmp_path)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 3
    assert out["sharded_identical_to_single_gpu"] is True
    assert out["config"]["launch"]["queries_per_launch"] in (750, 751)
    assert out["roofline"]["counted_pass_identical"] is True
    assert out["value"] > 0 and out["roofline"]["frac"] > 0
    # the same ranks as independent replicas (a whole batch per rank and step), reported next to `value`
    w = out["replicas_weak_scaling"]
    assert w["queries_per_gpu_per_step"] == 1501 and w["value"] > 0 and w["ms_per_step"] > 0
    assert "single_gpu_legs" in out and "end_to_end" not in out
    # what `value` measured is said in the line, and the literal reading of BASELINE configs[3] stands next to it:
    # one batch at a time (a barrier around every step, one call per rank)
    assert "pipelined" in out["value_is"]
    assert out["value_one_batch_at_a_time"] > 0 and out["one_batch_at_a_time"]["ms_per_step"] > 0
    assert len(out["kernel_ms_per_rank"]) == out["n_gpus"] and all(x > 0 for x in out["kernel_ms_per_rank"])   # (r06)
    assert int(out["host_threads_per_rank_for_host_phases"]) >= 1


def test_two_ranks_weak_scaling(tmp_path):
    # (and without the file hand-off: when rank 0 cannot write the index, every rank builds its own)
    out = _run("weak", tmp_path, SGPU_BENCH_NO_HANDOFF="1")
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".idx")]
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["launch"]["queries_per_launch"] == 1501
    assert out["value"] > 0


def test_bench_on_files_in_the_reference_formats(tmp_path):
    """bench.py --documents/--queries-file/--groundtruth/--results-tsv: Seismic's inner binary format in,
    perf_inverted_index's TSV out, accuracy as scripts/run_experiments.py:287-309 computes it. The ground
    truth here is the exact top-10 (sgpu_exact_search) written in the same TSV layout."""
    import numpy as np
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import seismic_amd
    from seismic_amd import _native
    from seismic_amd._abi import BuildConfig
    from seismic_amd.index import read_results_tsv
    dim = 30_000
    docs = _native.synth(60_000, dim, 42, 0)
    q = _native.synth(300, dim, 43, 1, docs)
    dp, qp, gp, rp = (str(tmp_path / n) for n in ("documents.bin", "queries.bin", "groundtruth.tsv", "results.tsv"))
    seismic_amd.write_inner_format(dp, *docs)
    seismic_amd.write_inner_format(qp, *q)
    ix = _native.NativeIndex.build(2, dim, *docs, BuildConfig.defaults(n_postings=300, centroid_fraction=0.2,
                                                                        summary_energy=0.5, max_fraction=6.0))
    es, ei, en = ix.exact_search(*q, 10)
    _native.write_results_tsv(gp, es, ei, en)
    env = dict(os.environ, SGPU_INDEX_CACHE=str(tmp_path))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--documents", dp, "--queries-file", qp, "--groundtruth", gp,
           "--results-tsv", rp, "--n-postings", "300", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-latency", "--no-e2e",
           "--target-recall", "0.5,0.999999"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["data"] == "file" and out["config"]["launch"]["queries_per_launch"] == 300
    res, gt = read_results_tsv(rp), read_results_tsv(gp)
    assert len(res) == 300 and all(len(v) == 10 for v in res.values())
    hit = sum(len(set(res[i]) & set(gt[i])) for i in range(300)) / 3000.0
    assert out["accuracy_vs_groundtruth"] == pytest.approx(hit) and hit > 0.8
    assert out["recall_at_k"] == pytest.approx(hit)        # the same quantity, computed by bench against exact search
    assert out["entry_point"]["rows_identical_to_device_resident_launch"] is True
    op = out["operating_points"]                            # fixed-recall operating points, each checked against the oracle
    assert op[0]["reached"] and op[0]["recall_at_k"] >= 0.5 and op[0]["identical_to_cpu_oracle_on_sample"] is True
    assert op[0]["value"] > 0 and 0 < op[0]["roofline_frac"] < 1
    assert op[1]["target_recall"] == 0.999999 and (op[1]["reached"] or op[1]["best_recall_on_grid"] < 0.999999)
    ix.upload(0)                                            # the TSV holds exactly what the API returns
    gs, gi, gn = ix.batch_search(*q, 10, 4, 1.0, False)
    assert [int(x) for x in gi[7, :gn[7]]] == res[7]
