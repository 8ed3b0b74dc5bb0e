"""`python bench.py --gpus N` with no launcher in front of it must start N ranks by itself (the driver runs exactly
that command; round 4 would have measured ONE GPU and printed n_gpus 1).  The launcher — environment per rank, relay of rank 0's
stdout, failure handling — is plain host logic and is tested here with stand-in rank commands; the GPU test
(`tests/test_bench_launcher_gpu.py`) runs the real two-rank bench on the one test GPU."""

import json
import os
import subprocess
import sys
import textwrap

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_launcher(body: str, n: int, tmp_path, timeout_s=None):
    """launch_ranks() in a child interpreter (it writes to ITS stdout); returns (rc, stdout lines)."""
    rank_script = tmp_path / "rank.py"
    rank_script.write_text(textwrap.dedent(body))
    prog = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
            f"sys.exit(bench.launch_ranks([sys.executable, {str(rank_script)!r}], {n}, timeout_s={timeout_s!r}))")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    res = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=120, env=env)
    return res.returncode, res.stdout.splitlines(), res.stderr


def test_every_rank_gets_the_launcher_environment_and_rank0_is_relayed(tmp_path):
    rc, out, err = _run_launcher(f"""
        import json, os
        rec = {{k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}}
        open(os.path.join({str(tmp_path)!r}, "env_" + rec["RANK"] + ".json"), "w").write(json.dumps(rec))
        print("noise of rank", rec["RANK"])
        print(json.dumps({{"n_gpus": int(rec["WORLD_SIZE"]), "rank": int(rec["RANK"])}}))
        """, 4, tmp_path)
    assert rc == 0, err
    assert out == ["noise of rank 0", json.dumps({"n_gpus": 4, "rank": 0})]        # ranks 1..3 are not heard; the LAST line is rank 0's record
    envs = [json.loads((tmp_path / f"env_{r}.json").read_text()) for r in range(4)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert {e["WORLD_SIZE"] for e in envs} == {"4"} and {e["MASTER_ADDR"] for e in envs} == {"127.0.0.1"}
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and int(envs[0]["MASTER_PORT"]) > 0


def test_a_failing_rank_stops_the_others_and_fails_the_launch(tmp_path):
    rc, out, err = _run_launcher("""
        import os, sys, time
        if os.environ["RANK"] == "2":
            sys.exit(7)
        time.sleep(60)
        print("never")
        """, 3, tmp_path)
    assert rc == 7 and out == [] and "rank 2 exited with status 7" in err


def test_ranks_that_hang_are_stopped_at_the_deadline(tmp_path):
    rc, out, err = _run_launcher("""
        import time
        time.sleep(60)
        """, 2, tmp_path, timeout_s=1.0)
    assert rc != 0 and "still running" in err


def test_the_ranks_rendezvous_through_the_socket_group(tmp_path):
    """The environment the launcher exports is what squidpy_amd._dist's rendezvous reads: 3 ranks meet and sum an array (no GPU)."""
    rc, out, err = _run_launcher(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import numpy as np
        from squidpy_amd import _dist
        _dist.init()
        r, w = _dist.world()
        (tot,) = _dist.allreduce_sum_([np.array([r + 1, 10 * (r + 1)], dtype=np.int64)])
        _dist.barrier()
        print(w, tot.tolist())
        _dist.shutdown()
        """, 3, tmp_path)
    assert rc == 0, err
    assert out == ["3 [6, 60]"]


def _bench(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)


def test_more_ranks_than_gpus_is_an_error_not_a_relabelled_figure():
    """No GPU here: `--gpus 2` must refuse (exit 2, a message that names both numbers), not run one rank and print n_gpus 1."""
    from squidpy_amd import _lib

    n_dev = _lib.device_count()
    ask = max(n_dev, 1) + 1
    res = _bench("--gpus", str(ask), "--steps", "1", "--warmup", "0")
    assert res.returncode == 2 and res.stdout.strip() == ""
    assert f"--gpus {ask}" in res.stderr and f"{n_dev} GPU" in res.stderr


def test_a_launcher_that_started_another_world_than_gpus_is_refused():
    res = _bench("--gpus", "8", "--steps", "1", "--warmup", "0", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode == 2 and "WORLD_SIZE=2" in res.stderr and res.stdout.strip() == ""
