"""CPU: `bench.py --gpus N` really starts N ranks, and the exchange step runs inside the step loop.

The driver launches bench.py under torchrun for N > 1; a plain `python bench.py --gpus 2` must do the same by itself (round 1
parsed the flag and ignored it).  --dry-run keeps everything of the N > 1 path that does not need a GPU -- the launcher, the
rank environment, the per-rank corpus seeds, the bitmap layout of the exchange and its place in the step loop -- and stubs the
verify call with the corpus' constructed verdicts and RCCL with gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "2", "--items", "24",
                          "--replicas", "4"] + extra, capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_flag_spawns_the_ranks():
    r = _run(["--gpus", "2"])
    assert r["n_gpus"] == 2 and r["world_size"] == 2
    assert r["allgathers_in_step_loop"] == 3 + 2          # one exchange per step, warm-up included
    assert r["gather_rows"] == 2 and r["own_row_matches"]
    assert r["gathered_ok"] == r["sum_of_rank_ok"] and 0 < r["gathered_ok"] < 2 * 24


def test_single_rank_needs_no_process_group():
    r = _run([])
    assert r["n_gpus"] == 1 and r["gather_rows"] == 1 and r["own_row_matches"] and r["allgathers_in_step_loop"] == 5


def test_torchrun_environment_is_honoured():
    """started the way the driver starts it for N > 1: under torch.distributed.run, with --gpus N on the command line too"""
    e = dict(os.environ)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1",
                          "--items", "16", "--replicas", "4"], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["world_size"] == 2 and r["allgathers_in_step_loop"] == 3
