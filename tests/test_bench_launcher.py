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
    # ... and nothing else on stdout: whatever libraries print (gloo's connection notes here, RCCL's version banner on the GPU box)
    # goes to stderr -- bench.py points file descriptor 1 there and writes its line to the descriptor stdout had
    assert out.stdout.strip() == lines[0], out.stdout[:400]
    assert len(lines[0]) < 6144           # the bound of every stdout line (bench.LINE_MAX): round 5's 25 KB line came back unparsed
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


# ---- the rank split of BASELINE's actual multi-GPU configs (cfg 3 / 4 / 5), worlds 2 and 8: no GPU, no number -----------------
def _dry(cfg, world, extra=()):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--config", str(cfg), "--gpus", str(world), "--steps", "2",
                          "--warmup", "1"] + list(extra), capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("world", (2, 8))
def test_cfg4_storm_is_split_and_gathered_in_write_order(world):
    """1,000,000 writes over `world` ranks (strong scaling): the shares sum to the storm, every call's all-gather moves chunk / 8
    bytes per rank (8 ranks: 15,625 B each, 125 KB gathered), and the gathered rows, laid out rank by rank and call by call, ARE the
    storm's verdict vector in write order (rebuilt here from the constructed verdicts of the global write indices)."""
    import bench
    r = _dry(4, world)
    assert r["world_size"] == world and r["scaling"] == "strong"
    assert r["writes_per_step"] == 1000000 == r["share_per_rank"] * world and r["share_per_rank"] == r["writes_per_call"] * r["calls_per_step"]
    assert r["calls_per_step"] == 8 // world and r["writes_per_call"] == 125000 and r["tiles"] == 50
    assert r["bitmap_bytes_per_rank_per_call"] == 15625 and r["gathered_bytes_per_call"] == 15625 * world
    if world == 8:
        assert r["gathered_bytes_per_call"] == 125000                       # "cfg 4: 125 KB total"
    assert r["allgathers_in_step_loop"] == (2 + 1) * r["calls_per_step"]     # one exchange per call, warm-up included
    want = bench.constructed_ok(0, 1000000, 4)
    assert r["gathered_ok"] == r["sum_of_rank_ok"] == int(want.sum()) and 0 < r["gathered_ok"] < 1000000
    assert r["ranks_whose_own_rows_match"] == world
    assert r["verdict_sha256"] == bench.verdict_digest(want)


@pytest.mark.parametrize("world", (2, 8))
def test_cfg5_operations_are_sharded_without_an_exchange(world):
    r = _dry(5, world)
    assert r["world_size"] == world and r["scaling"] == "strong" and r["exchange_steps"] == 0
    rs = r["operation_ranges"]
    assert rs[0][0] == 0 and rs[-1][1] == 10000 and all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))      # the shards tile [0, 10000)
    assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1
    assert r["scheme_ops_per_step"] == 30000 and len(set(r["corpus_seeds"])) == world


@pytest.mark.parametrize("world", (2, 8))
def test_cfg3_ranks_with_different_reply_counts_share_one_bitmap_width(world):
    """cfg 3 shards by variable and its ranks hold DIFFERENT numbers of replies: every rank must contribute the same byte count to
    the all-gather (the largest shard's), and each row is cut back to its rank's count when the job's verdicts are put together."""
    import numpy as np
    import bench
    r = _dry(3, world)
    per = r["replies_per_rank"]
    assert len(set(per)) > 1 and r["slots"] == max(per) and r["bitmap_bytes_per_rank"] == (max(per) + 7) // 8
    vr = r["variable_ranges"]
    assert vr[0][0] == 0 and vr[-1][1] == 10000 and all(vr[i][1] == vr[i + 1][0] for i in range(world - 1))
    total = int((9 + np.arange(10000) % 3).sum())
    assert r["replies_total"] == total == sum(per)
    want = bench.constructed_ok(0, total, 3)
    assert r["gathered_ok"] == r["sum_of_rank_ok"] == int(want.sum()) and r["verdict_sha256"] == bench.verdict_digest(want)


@pytest.mark.parametrize("world", (2, 8))
def test_default_line_of_a_multi_rank_run_carries_cfg4_and_cfg5(world):
    """`bench.py --gpus N` (N > 1) prints ONE line: cfg 2's, with the storm (cfg 4) and the share combine (cfg 5) behind it as
    `other_configs`, run in the same process group."""
    r = _run(["--gpus", str(world)])
    assert r["world_size"] == world and r["config"] == 2 and r["gather_rows"] == world
    oc = r["other_configs"]
    assert set(oc) == {"cfg4", "cfg5"} and "error" not in oc["cfg4"] and "error" not in oc["cfg5"]
    c4, c5 = oc["cfg4"], oc["cfg5"]
    assert c4["config"] == 4 and c4["world_size"] == world and c4["writes_per_step"] == c4["share_per_rank"] * world
    assert c4["gathered_ok"] == c4["sum_of_rank_ok"] and c4["ranks_whose_own_rows_match"] == world
    assert c5["config"] == 5 and c5["operation_ranges"][-1][1] == 10000 and c5["exchange_steps"] == 0
    # the N > 1 line names its rank count where the contract puts it (`config.parallelism` of a measured line; the dry line's
    # `config` is the config number, so the key sits beside it)
    assert ("x%d" % world) in r["parallelism"]


def test_measured_multi_rank_line_fits_the_bound_and_names_the_rank_count():
    """The N > 1 form of a MEASURED line (the one-rank forced-RCCL rehearsal of round 5, cfg 4 and cfg 5 behind the headline in the same
    process group) through the stdout emitter: at most LINE_MAX bytes, `config.parallelism` with the rank count, roofline and summary kept."""
    import bench
    for name in ("r05_bench_multi_rank_form_one_rank_forced_rccl.json", "r06_bench_multi_rank_form_one_rank_forced_rccl.json"):
        _check_multi_rank_record(bench, json.load(open(os.path.join(ROOT, "profiles", name))))
    # what the round-6 rehearsal really printed (BFTKV_FORCE_RCCL=1 BFTKV_BENCH_EXTRAS_IN_PROCESS=1 python bench.py --gpus 1 ...)
    line = open(os.path.join(ROOT, "profiles", "r06_bench_multi_rank_form_stdout_line.json")).read().strip()
    r = json.loads(line)
    assert len(line) < bench.LINE_MAX and "x1" in r["config"]["parallelism"] and {"cfg4", "cfg5"} <= set(r["summary"])
    assert r["roofline"]["launch_ms"] <= r["ms_per_step"] and r["summary"]["cfg4"]["roofline"]["launch_ms"] <= r["summary"]["cfg4"]["ms_per_step"]


def _check_multi_rank_record(bench, d):
    for world in (1, 8):
        d["n_gpus"] = world
        d["config"]["parallelism"] = "shard-by-write x%d, RCCL all-gather of verdict bitmaps on the verifier's stream" % world
        line = json.dumps(bench.compact_line(d, os.path.join(ROOT, "bench_full.json")))
        assert len(line) < bench.LINE_MAX == 6144
        r = json.loads(line)
        assert ("x%d" % world) in r["config"]["parallelism"] and r["n_gpus"] == world
        assert r["roofline"]["frac"] > 0 and {"cfg4", "cfg5"} <= set(r["summary"]) and r["full_record"] == "bench_full.json"
