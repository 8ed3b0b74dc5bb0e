"""The N>1 code path — libsqgr's own RCCL communicator (sqgr_comm_*) — exercised on real hardware with one rank, once
under a torch.distributed nccl group (torch only carries the unique id) and once with no torch at all."""
import os, socket, subprocess, sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_collective_path_on_rccl():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "rccl_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "RCCL_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


def test_collective_path_on_rccl_without_torch():
    env = dict(os.environ, PYTHONPATH=ROOT, SQGR_TEST_GROUP="socket", RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "RCCL_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
