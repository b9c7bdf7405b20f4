"""CPU, world_size 2 over gloo: the N>1 path of the verifier -- shard-by-write partitioning and the all-gather
of verdict bitmaps (bftkv_amd/dist.py).  Verdicts come from the oracle (there is no GPU here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bftkv_amd import dist as D


def test_shard_ranges_partition_the_items():
    for n in (0, 1, 7, 8, 10000, 1000003):
        for w in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) == D.max_shard(n, w) or n == 0


def test_pack_unpack_round_trip():
    rng = np.random.default_rng(0)
    for n in (0, 1, 7, 8, 9, 1001):
        ok = torch.from_numpy(rng.integers(0, 2, size=n).astype(np.uint8))
        bits = D.pack_verdicts(ok, n + 5)
        assert bits.numel() == (n + 5 + 7) // 8
        assert torch.equal(D.unpack_verdicts(bits, n), ok)
        assert D.unpack_verdicts(bits, n + 5)[n:].sum() == 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank derives the same corpus and quorum; each verifies (oracle stands in for the GPU) only its shard
        from corpus import build as cb
        from tests import helpers as H
        cl = cb.make_cluster(4)
        c = cb.make_write_corpus(cl, n_items, mutation_rates={cb.MUT_ONE_SHORT: 0.3, cb.MUT_BAD_MPI: 0.2})
        kr, quorum = H.oracle_keyring(cl), H.clique_quorum(cl)
        lo, hi = D.shard_range(n_items, rank, world)
        local = torch.tensor([1 if H.oracle_collective(kr, quorum, c, i).err is None else 0 for i in range(lo, hi)], dtype=torch.uint8)
        allv = D.allgather_verdicts(local, n_items)
        q.put((rank, lo, hi, local.tolist(), allv.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [21, 20, 16])      # ragged shards; equal shards (the bench's case) with and without padding bits
def test_allgather_verdicts_world2_gloo(n_items):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = []
    for rank, lo, hi, local, allv in res:
        assert (lo, hi) == D.shard_range(n_items, rank, world)
        full += local
    assert len(full) == n_items and 0 < sum(full) < n_items
    for _, _, _, _, allv in res:
        assert allv == full          # every rank holds every verdict, in global write order
