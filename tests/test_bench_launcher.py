"""bench.py's rank launcher: `python bench.py --gpus N` as a plain command must start N
ranks itself (never a silent 1-GPU run), and a WORLD_SIZE that disagrees with --gpus is an
error.  The CPU half checks launcher and rank plumbing (--rendezvous-only: the ranks meet
on gloo and build their rank-local cases); the GPU half runs the real measurement with
two ranks sharing cuda:0 over the host-staged gloo transport."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env=None, timeout=900):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + argv, env=e, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


@pytest.mark.parametrize("workload", ["lusgs", "dplur8"])
def test_plain_command_launches_n_ranks(workload):
    r = _run(["--gpus", "2", "--backend", "gloo", "--size", "16", "--workload", workload,
              "--rendezvous-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["rendezvous_only"] is True
    # chain: one block per rank, one connection between them; dplur8: 8 blocks over 2 ranks
    assert line["blocks_held"] == (2 if workload == "lusgs" else 8)
    assert line["remote_connections"] >= 1


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--rendezvous-only"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    r = _run(["--gpus", "1", "--rendezvous-only"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_failing_rank_fails_the_command():
    # an unknown workload makes every rank exit non-zero; the parent must relay that
    r = _run(["--gpus", "2", "--backend", "gloo", "--workload", "nope", "--rendezvous-only"])
    assert r.returncode != 0


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_plain_command():
    r = _run(["--gpus", "2", "--backend", "gloo", "--size", "32", "--steps", "2",
              "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 2
    assert "gloo" in line["config"]["halo"]
    assert line["value"] > 0 and line["config"]["cells_per_gpu"] == 32 ** 3
    # the 8-block DPLUR case (strong scaling) rides along for N > 1
    d8 = line["extra"]["dplur8"]
    assert d8["n_gpus"] == 2 and d8["scaling"] == "strong" and d8["value"] > 0
