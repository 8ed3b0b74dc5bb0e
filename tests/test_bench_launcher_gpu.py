"""`python bench.py --gpus 2` — exactly as the driver invokes it, no launcher in front — starts two ranks, and the figure it
prints belongs to two ranks: n_gpus == 2, and the all-reduced moments of its timed steps equal those of ONE rank over the same
global permutation range.  The test box has one GPU: `--share-devices` puts both ranks on it (integer all-reduce through the host
side channel, as tests/test_dist2_gpu.py does); on a multi-GPU node the same command without that flag uses RCCL inside libsqgr."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SMALL = ["--rows", "120", "--cols", "150", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-legs", "--no-secondary", "--no-numpy-leg",
         "--emulate-ranks", "0"]


def _bench(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


def test_gpus_2_runs_two_ranks_and_their_moments_are_one_ranks_moments(tmp_path):
    two = _bench("--gpus", "2", "--share-devices", "--perms-per-step", "320", "--detail-out", str(tmp_path / "two.json"), *SMALL)
    one = _bench("--gpus", "1", "--perms-per-step", "640", "--detail-out", str(tmp_path / "one.json"), *SMALL)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["ranks_on_devices"] == [0, 0] and two["config"]["perms_per_step_per_gpu"] == 320
    assert "host" in two["config"]["collective"] and two["config"]["rccl_world"] is None   # two ranks on one device: no RCCL communicator
    assert two["check"]["perm_range"] == one["check"]["perm_range"] == [640, 1920]
    assert two["check"]["moments_sha16"] == one["check"]["moments_sha16"]
    assert two["value"] > 0 and two["scaling"] == "weak"


def test_strong_scaling_shards_config5s_range(tmp_path):
    two = _bench("--gpus", "2", "--share-devices", "--scaling", "strong", "--total-perms", "1000", "--detail-out", str(tmp_path / "two.json"), *SMALL)
    one = _bench("--gpus", "1", "--scaling", "strong", "--total-perms", "1000", "--detail-out", str(tmp_path / "one.json"), *SMALL)
    assert two["n_gpus"] == 2 and two["config"]["perms_per_step_per_gpu"] == 500
    assert two["check"] == one["check"]


def test_gpus_8_at_the_nodes_size_on_one_gpu(tmp_path):
    """The driver's 8-GPU invocation, as far as one GPU can show it: `bench.py --gpus 8` starts EIGHT ranks of itself;
    their all-reduced moments are one rank's over the same global permutation range."""
    flags = ["--no-legs", "--no-secondary", "--no-cpu-baseline", "--no-numpy-leg", "--emulate-ranks", "0", "--rows", "120", "--cols", "150", "--steps", "2", "--warmup", "1"]
    eight = _bench("--gpus", "8", "--share-devices", "--perms-per-step", "1024", "--detail-out", str(tmp_path / "eight.json"), *flags)
    one = _bench("--gpus", "1", "--perms-per-step", "8192", "--detail-out", str(tmp_path / "one.json"), *flags)
    assert eight["n_gpus"] == 8 and eight["config"]["ranks_on_devices"] == [0] * 8 and eight["config"]["perms_per_step_per_gpu"] == 1024
    assert eight["check"]["perm_range"] == one["check"]["perm_range"] == [8192, 3 * 8192]
    assert eight["check"]["moments_sha16"] == one["check"]["moments_sha16"]


def test_strong_scaling_shards_config5_into_8_x_12500(tmp_path):
    """BASELINE config 5 in full — 1e6 spots, 30 clusters, 100 000 permutations per step — cut into the 8 rank shards of a node."""
    flags = ["--no-legs", "--no-secondary", "--no-cpu-baseline", "--no-numpy-leg", "--emulate-ranks", "0", "--steps", "2", "--warmup", "1", "--scaling", "strong",
             "--total-perms", "100000"]
    eight = _bench("--gpus", "8", "--share-devices", "--detail-out", str(tmp_path / "eight.json"), *flags)
    one = _bench("--gpus", "1", "--detail-out", str(tmp_path / "one.json"), *flags)
    assert eight["n_gpus"] == 8 and eight["scaling"] == "strong" and eight["config"]["perms_per_step_per_gpu"] == 12500
    assert eight["config"]["perms_per_step"] == 100000 and "1000000 spots" in eight["config"]["workload"]
    assert eight["check"] == one["check"] and eight["check"]["perm_range"] == [100000, 300000]
