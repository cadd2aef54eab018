"""The self-launcher behind `python bench.py --gpus N` (vsearch_amd/launch.py), exercised on CPU: N fresh rank processes
from a plain parent, env:// rendezvous on 127.0.0.1, gloo standing in for RCCL."""
import os
import subprocess
import sys

import numpy as np

from vsearch_amd import launch

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "_rank_worker.py")


def test_spawned_ranks_search_a_sharded_index(tmp_path):
    import oracle
    rc = launch.spawn_ranks([sys.executable, WORKER, "search", str(tmp_path)], 2, timeout_s=300)
    assert rc == 0
    V, n_total, k = 29523, 1501, 100
    ip, ix, d = oracle.synth_csr(21, 0, n_total, V, 768, 0)
    want_ids, want_sc = oracle.csr_search(ip, ix, d, V, oracle.synth_queries(22, 4), k)
    for r in range(2):
        got = np.load(tmp_path / f"r{r}.npz")
        assert int(got["world"]) == 2
        assert (got["ids"] == want_ids).all() and (got["sc"] == want_sc).all()


def test_failed_rank_ends_the_job(tmp_path):
    rc = launch.spawn_ranks([sys.executable, WORKER, "fail", str(tmp_path)], 2, timeout_s=120)
    assert rc == 3


def test_rank_env_and_detection(monkeypatch):
    env = launch.rank_env(3, 8, 29500, base={})
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    assert not launch.under_launcher()
    launch.relaunch_as_ranks(1, "x.py", [])            # one rank: no relaunch
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    assert launch.under_launcher()
    launch.relaunch_as_ranks(2, "x.py", [])            # already a rank: returns


def test_bench_parent_does_not_need_a_gpu():
    """`bench.py --gpus 2` from a plain shell: the parent only launches; without a GPU the ranks fail loudly and the
    parent reports their failure (non-zero), it does not raise the old 'use torch.distributed.run' refusal."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--docs", "4096", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0 and "no HIP device visible" in p.stderr
        assert "torch.distributed.run" not in p.stderr
