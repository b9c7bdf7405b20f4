"""CPU: the arithmetic behind bench.py's headline -- `value` counts the public-key operations the REFERENCE performs, computed
from the verifier's per-packet statuses; checked here against the Python oracle's own walk on a small mutated corpus."""
import numpy as np

import bench
from corpus import build as cb
from oracle import openpgp as pgp
from tests import helpers as H


def test_reference_pubkey_ops_counts_what_the_reference_verifies():
    cl = cb.make_cluster(7, dsa_fraction=0.3)
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    c = cb.make_write_corpus(cl, 40, mutation_rates={cb.MUT_BAD_MPI: 0.2, cb.MUT_UNKNOWN_ISSUER: 0.15, cb.MUT_DUP_SIGNER: 0.1,
                                                      cb.MUT_ONE_SHORT: 0.2, cb.MUT_BAD_TAG: 0.1})
    # what a verify-everything run of the verifier reports: the status of EVERY packet (the oracle with the early exit disabled)
    st, item, err, nver, want_ops = [], [], [], [], 0
    for i in range(c.n_items):
        r = H.oracle_collective(kr, q, c, i)                       # the reference: stops at the first sufficient prefix
        err.append(0 if r.err is None else 2)
        nver.append(len(r.verified))
        want_ops += sum(1 for s in r.statuses if s in (pgp.ST_OK, pgp.ST_BAD_SIG))
        pos, data, tbs = 0, c.ss_data(i), c.tbss(i)
        while pos < len(data):                                     # every packet, no exit
            step = pgp.check_detached_signature(kr.get_keyring(), tbs, data, pos)
            pos = step.pos
            st.extend(step.statuses)
            item.extend([i] * len(step.statuses))
    got = bench.reference_pubkey_ops(np.array(st, dtype=np.uint8), np.array(item, dtype=np.int64), np.array(err, dtype=np.uint8),
                                     np.array(nver, dtype=np.uint32), c.n_items)
    assert got == want_ops and 0 < want_ops < len(st)


def test_int_mac_block_bases():
    b = bench.int_mac_block(1e9, 2.0, 1.6, 1.5, 2200.0, 3)
    assert abs(b["achieved"] - 5e11) < 1 and b["per_launch_in_timed_region"]["launch_ms"] == 1.6 and b["single_flight"]["launch_ms"] == 1.5
    assert b["single_flight"]["frac"] > b["per_launch_in_timed_region"]["frac"] > b["frac"] and "3 batches" in b["basis"]
    assert "single_flight" not in bench.int_mac_block(1e9, 2.0, 1.6, None, None, 1)


def test_cfg1_leg_times_the_c_restatement_on_the_reference_shape():
    """BASELINE.json configs[0]: 4 replicas, 100 RSA-2048 signed writes on the CPU path (dry: without the GPU identity call)."""
    class D:
        dry, rank, world, local_rank = True, 0, 1, 0
    out = bench.bench_cfg1(bench.parse_args(["--config", "1", "--dry-run"]), D)
    assert out["device"] == "cpu" and out["unit"] == "verifies/s" and out["value"] > 0 and out["threads"] == 1
    assert "4-replica" in out["workload"] and "100 RSA-2048" in out["workload"] and out["all_cores"]["threads"] == bench.effective_cores()


def test_summarize_keeps_what_the_default_line_promises():
    full = {"metric": "m", "value": 1.0, "unit": "u", "steps": 2, "warmup": 1, "ms_per_step": 3.0, "scaling": "weak", "dtype": "u32", "data": "synthetic",
            "config": {"workload": "w"}, "int_mac": {"achieved": 1, "frac": 0.5, "frac_of_theoretical": 0.4, "basis": "b", "peak": 2},
            "roofline": {"bound": "hbm", "kernel": "k", "achieved": 1, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": 5, "traffic_source": "p",
                         "launch_ms": 1.0, "note": "n"},
            "cpu_baseline": {"value": 2.0, "unit": "u", "cores": 16, "threads": 64, "kind": "port", "sample": "s", "gpu_verdicts_identical_to_cpu": True},
            "kernel_ms": {"k_rsa_modexp": 1.0, "measured": "text"}, "verdicts_match_construction": True}
    s = bench.summarize(full)
    assert s["workload"] == "w" and s["int_mac"]["frac"] == 0.5 and s["roofline"]["traffic"] == 5 and s["roofline"]["frac"] == 0.1
    assert s["cpu_baseline"] == {"value": 2.0, "unit": "u", "cores": 16, "threads": 64, "kind": "port"}
    assert s["identity"] == {"gpu_verdicts_identical_to_cpu": True} and s["kernel_ms"] == {"k_rsa_modexp": 1.0}
    assert bench.summarize(None) is None


def test_dsa_mac_count_follows_the_table_width():
    """bench.py prices a DSA verification by the width the library built its tables at (bftkv_gpu_dsa_window_bits): 2 * ceil(256 / bits) - 1
    Montgomery products of 11,552 limb MACs."""
    assert [bench.macs_per_dsa_verify(b) // 11552 for b in (4, 8, 16, 17, 18, 19, 20)] == [127, 63, 31, 31, 29, 27, 25]
    assert bench.macs_per_dsa_verify(0) == 0 and bench.macs_per_dsa_verify(18) == 29 * 11552


def test_summarize_keeps_what_the_default_line_shows_per_config():
    import json
    import os
    full = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_cfg3_bench_18bit_tables.json")))
    s = bench.summarize(full)
    assert s["metric"] == full["metric"] and s["value"] == full["value"] and s["workload"] == full["config"]["workload"]
    assert s["dsa_tables"] == {"window_bits": 18, "products_per_verify": 29, "note": full["dsa_tables"]["note"]}
    assert set(s["int_mac"]) == {"achieved", "frac", "frac_of_theoretical", "basis"} and s["roofline"]["kernel"] == "k_dsa_modexp"
    assert all(isinstance(v, (int, float)) for v in s["kernel_ms"].values())
    assert bench.summarize(None) is None


def test_verifier_gather_check_with_unequal_shards():
    """cfg 3's ranks hold different numbers of replies: every rank contributes the LARGEST shard's byte count to the all-gather
    (Verifier.slots = max over ranks) and check_gather compares this rank's row on its own length and the population count of ALL
    rows -- padding bits are zero -- with the all-reduced count.  Two fake ranks, 13 and 21 replies."""
    import torch
    from bftkv_amd import dist as BD
    n_items = [13, 21]
    rng = np.random.default_rng(3)
    errs = [np.where(rng.random(n) < 0.3, 2, 0).astype(np.uint8) for n in n_items]
    slots = max(n_items)
    rows = [BD.pack_verdicts(torch.from_numpy((e == 0).astype(np.uint8)), slots).numpy() for e in errs]
    gathered = np.concatenate(rows)
    total_ok = int(sum((e == 0).sum() for e in errs))

    torch_mod = torch

    class FakeDist:
        dry, world, torch = True, 2, torch_mod

        def __init__(self, rank):
            self.rank = rank

        def max_int(self, v):
            return slots

        def sum_ints(self, vals):
            return [total_ok]
    for r in range(2):
        D = FakeDist(r)
        V = bench.Verifier(D, None, n_items[r], None, None, None, np.zeros(1, dtype=np.uint64), n_ctx=1)
        assert V.slots == slots and (V.slots + 7) // 8 == rows[0].size
        assert V.check_gather(errs[r], gathered)
        # a flipped bit in the OTHER rank's row, or this rank's row shifted by one, is noticed
        bad = gathered.copy(); bad[(1 - r) * rows[0].size] ^= 1
        assert not V.check_gather(errs[r], bad)
        assert not V.check_gather(np.roll(errs[r], 1) if errs[r].any() and not (np.roll(errs[r], 1) == errs[r]).all() else 1 - errs[r], gathered)


def test_line_summary_digests_every_committed_line_shape():
    """The digest at the end of the line (bench.line_summary) over real lines of every config: cfg 2's default line with its
    other_configs, cfg 3's, cfg 5's with its own `serving` shape (one share-combine per call, per scheme) -- and it never raises."""
    import glob
    import json
    import os
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    lines = [json.load(open(p)) for p in sorted(glob.glob(os.path.join(prof, "r05_*bench*.json")))]
    lines += [json.load(open(os.path.join(prof, "r05_cfg3_dsa64.json")))["policy_run_line"]]
    assert len(lines) >= 6
    seen = set()
    for d in lines:
        s = bench.line_summary(d)
        assert s["value"] == d["value"] and s["metric"] == d["metric"]
        seen.update(s)
        json.dumps(s)
    assert {"serving_verify_calls_per_s", "serving_ops_per_s_256_callers", "single_flight_ms_per_step", "host_buffers_ms_per_call", "cfg4", "cfg5"} <= seen
    assert "error" not in bench.line_summary({"metric": "m", "value": 1.0, "serving": {"runs": []}, "other_configs": {"cfg9": "text"}})


def test_stdout_line_is_bounded_and_carries_what_the_driver_parses():
    """Round 5's default line had grown to 25 KB and the driver's parser returned nothing (BENCH_r05.json: parsed null).  The stdout
    emitter (bench.compact_line) over that very record, and over every other committed full record: at most LINE_MAX = 6144 bytes,
    json round-trips, the contract's keys, `roofline.frac`, `cpu_baseline.value`, `config.workload` present; value and ms_per_step
    keep every digit."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = json.load(open(os.path.join(root, "profiles", "r05_bench_default.json")))
    assert len(json.dumps(full)) > 20000
    line = json.dumps(bench.compact_line(full, os.path.join(root, "bench_full.json")))
    assert len(line) < 6144 == bench.LINE_MAX
    r = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in r, k
    assert r["value"] == full["value"] and r["ms_per_step"] == full["ms_per_step"]
    assert r["roofline"]["frac"] > 0 and r["roofline"]["bound"] == "hbm" and r["roofline"]["kernel"] == "k_rsa_modexp"
    assert r["cpu_baseline"]["value"] > 0 and r["cpu_baseline"]["cores"] >= 1 and r["cpu_baseline"]["kind"] == "port"
    assert r["cpu_baseline"]["gpu_verdicts_identical_to_cpu"] is True and r["cpu_baseline"]["untuned"] is False
    assert "10000 RSA-2048 signed writes" in r["config"]["workload"] and r["int_mac"]["frac"] > 0.5
    assert r["quorum_verdicts_per_sec"] > 0                       # BASELINE.json's metric names both rates
    s = r["summary"]
    assert {"cfg1", "cfg3", "cfg4", "cfg5", "host_buffers", "serving_verify_calls_per_s"} <= set(s)
    assert s["cfg5"]["single_flight_ms_per_step"] > s["cfg5"]["ms_per_step"]            # both in the same cell
    assert s["cfg3"]["identity"]["read_answers_identical_to_oracle"] is True and s["cfg3"]["cpu_baseline"]["value"] > 0
    assert r["full_record"] == "bench_full.json"
    # every other committed full record, whatever its config
    for p in sorted(glob.glob(os.path.join(root, "profiles", "r0[56]_*bench*.json"))):
        d = json.load(open(p))
        if "metric" not in d:
            continue
        t = json.dumps(bench.compact_line(d, None))
        assert len(t) < bench.LINE_MAX, p
        assert json.loads(t)["value"] == d["value"]
    # a record bloated far beyond anything seen still fits: digests are dropped in order of importance, never the contract's keys
    fat = json.loads(json.dumps(full))
    fat["other_configs"].update({"cfg%d" % k: dict(fat["other_configs"]["cfg5"]) for k in range(6, 30)})
    t = json.dumps(bench.compact_line(fat, None))
    assert len(t) < bench.LINE_MAX and json.loads(t)["roofline"]["frac"] > 0 and json.loads(t)["cpu_baseline"]["value"] > 0


def test_roofline_launch_ms_is_a_single_flight_duration_in_every_committed_r06_line():
    """Round 5's cfg-3 entry carried an overlapped three-in-flight span (62.7 ms) as the kernel's launch_ms against a 44.8 ms step.  Every
    roofline entry of the round-6 records: launch_ms <= ms_per_step of the basis it names (the step itself, or -- cfg 5, whose headline
    keeps 16 independent steps in flight -- the single-flight step it is measured in); overlapped spans sit under in_flight_*."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = 0
    for p in sorted(glob.glob(os.path.join(root, "profiles", "r06_*bench*.json"))):
        d = json.load(open(p))
        entries = [d] + [v for v in (d.get("other_configs") or {}).values() if isinstance(v, dict)]
        entries += [v for v in (d.get("summary") or {}).values() if isinstance(v, dict) and "roofline" in v]
        for e in entries:
            rf = e.get("roofline")
            if not rf or rf.get("launch_ms") is None or e.get("ms_per_step") is None:
                continue
            basis = rf.get("single_flight_ms_per_step") or e.get("single_flight_ms_per_step") or e["ms_per_step"]
            assert rf["launch_ms"] <= max(basis, e["ms_per_step"]) * 1.001, (p, rf, e["ms_per_step"])
            seen += 1
    if not glob.glob(os.path.join(root, "profiles", "r06_*bench*.json")):
        import pytest
        pytest.skip("no round-6 record committed yet")
    assert seen >= 1
