"""`python bench.py --gpus N` with no launcher around it starts its own ranks (bench.py:
launch_ranks).  Rehearsed on CPU: `--dry-launch` ranks meet over gloo instead of running the
GPU path, so what is tested is the launcher -- N fresh children with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_*, one line on stdout from rank 0, the worst exit code handed on."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(*args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args),
                          capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.parametrize("n", [2, 3])
def test_self_launch_starts_n_ranks_and_prints_one_line(n):
    r = launch("--gpus", str(n), "--dry-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_launch"] is True and d["n_gpus"] == n
    assert d["rank_sum"] == n * (n + 1) / 2   # every rank took part in the all-reduce


def test_self_launch_hands_on_a_failing_rank():
    """A rank that dies ends the job with a non-zero code instead of leaving the others in a
    collective: here every rank fails at argument parsing."""
    r = launch("--gpus", "2", "--dry-launch", "--no-such-flag")
    assert r.returncode != 0
    assert r.stdout.strip() == ""


def test_launcher_is_not_used_under_an_external_launcher():
    """With WORLD_SIZE in the environment (torch.distributed.run) the process is a rank."""
    r = launch("--gpus", "2", "--dry-launch", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0",
                                                         "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577"})
    # a lone rank of a one-rank world: the rehearsal completes with a world of one
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]  # gloo prints a banner too
    assert json.loads(line)["n_gpus"] == 1
