"""bench.py's N>1 launch path on CPU: started plainly (no RANK in the environment) with
--gpus N it must spawn its own ranks, rendezvous on 127.0.0.1 and let rank 0 print ONE JSON
line — the form the round driver uses.  `--rehearse-launch` keeps GPU and evaluation out of it
(it is a launch-path test, not a measurement)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


@pytest.mark.parametrize("n", [2, 3])
def test_plain_launch_spawns_its_ranks(n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--rehearse-launch"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {"rehearsal": True, "n_gpus": n, "max_rank_seen": n - 1}


@pytest.mark.parametrize("branch,path", [("own", "own"), ("children_failed", "rccl"), ("connect_failed", "rccl"), ("warmup_timeout", "rccl"),
                                         ("rccl_failed", "none")])
def test_comm_path_selection_branches(branch, path):
    """What the N > 1 timed step contains (bench.choose_comm_path), every branch forced with simulated outcomes under gloo:
    the library's mailboxes, the torch.distributed fallback when the child check / the wiring / a warm-up wait failed (one
    rank's time-out is everybody's), and 'none' only when the fallback failed too."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch", "--rehearse-comm", branch],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["comm"]["path"] == path and out["comm"]["world_size"] == 2
    assert (out["comm"]["why"] == "") == (path == "own")


def test_failed_rank_fails_the_launch():
    """Without a GPU a real run cannot start: every rank dies, and the parent must report it (non-zero
    exit, no JSON line) instead of hanging at a barrier."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
