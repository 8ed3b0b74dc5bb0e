"""world_size=2 test of the N>1 path on CPU (gloo): permutation-range sharding + exact integer all-reduce,
seed broadcast, co-occurrence shard sums, autocorr feature-block merge."""

from __future__ import annotations

import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo():
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
        os.path.join(ROOT, "tests", "dist_worker.py"),
    ]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "DIST_OK" in res.stdout


def test_socket_rendezvous_without_torch_up_to_the_node_size():
    """The product's own side channel: plain processes, launcher-style environment, no torch anywhere — 2, 3 and 8 ranks (the
    target machine has 8 GPUs): config 5's strong-scaling shard arithmetic, both co-occurrence shard axes (with 8 ranks and 6
    thresholds some ranks own no interval at all), ragged feature-block ownership."""
    for world in (2, 3, 8):
        port = _free_port()
        procs = []
        for rank in range(world):
            env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", SQGR_TEST_GROUP="socket", RANK=str(rank), LOCAL_RANK=str(rank),
                       WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=600) for p in procs]
        for p, (o, e) in zip(procs, outs):
            assert p.returncode == 0, o[-2000:] + e[-3000:]
        assert "DIST_OK" in outs[0][0]
