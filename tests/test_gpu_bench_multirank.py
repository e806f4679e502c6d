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


def _run(scaling, tmp_path):
    env = dict(os.environ, SGPU_BENCH_ONE_DEVICE="1", SGPU_BENCH_BACKEND="gloo", SGPU_INDEX_CACHE=str(tmp_path),
               MASTER_ADDR="127.0.0.1")
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


def test_two_ranks_weak_scaling(tmp_path):
    out = _run("weak", tmp_path)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["launch"]["queries_per_launch"] == 1501
    assert out["value"] > 0
