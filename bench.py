#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on BASELINE.json configs[1]:
"64-replica quorum, 10k RSA-2048 signed writes, single MI355X batched verify".

One "step" = one pass of the whole hot path (packet parse -> SHA-256 -> RSA-2048 verify -> quorum
tally) over one batch of synthetic signed writes that is ALREADY RESIDENT IN HBM when the timed
region starts.  N>1: one process per GPU, every rank verifies its own shard of writes (weak scaling)
and the per-write verdict bitmaps are all-gathered over RCCL inside every step.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
INT_MAC_PEAK = 29.1e12         # measured v_mad_u64_u32 lane-ops/s on MI355X (tools/microbench, profiles/)
RSA_BYTES = 291                # algorithmic bytes per RSA-2048 signature verify
MADS_PER_VERIFY = 18 * 2 * 76 * 76   # 18 Montgomery products x (76x76 a*b + 76x76 m*n) limb MACs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--items", type=int, default=10000, help="signed writes per GPU per step")
    ap.add_argument("--replicas", type=int, default=64)
    ap.add_argument("--inflight", type=int, default=1, help="batches (steps) in flight per GPU, each on its own verifier context / "
                    "HIP streams.  2 overlaps the walk/parse of step i+1 and the compare/tally of step i-1 with the modexp of step i "
                    "(+6 %% throughput, profiles/r01_v8_*), but then two modexp kernels also share the GPU and a launch's duration no "
                    "longer says anything about the kernel: the default 1 keeps the roofline line meaningful")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--corpus-cache", default="", help="path prefix of an .npz cache of the generated corpus (profiling reruns)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from bftkv_amd import Context
    from corpus import build as cb

    ctx = Context(local_rank)
    n = args.replicas
    cl = cb.make_cluster(n)

    # ---- synthetic signed writes; RSA signatures made on this GPU (generic modexp kernel)
    mods, exps = cb.signer_tables(cl)   # replicas in order, then the client (corpus/build.py BatchSigner)

    def gpu_signer(em, key_index):
        return ctx.modexp(em, key_index.astype(np.uint32), mods, exps)

    t0 = time.time()
    cache = None
    if args.corpus_cache:
        cache = "%s.n%d.i%d.r%d.npz" % (args.corpus_cache, n, args.items, rank)
    if cache and os.path.exists(cache):
        z = np.load(cache)
        corpus = cb.WriteCorpus(cl, args.items, z["tb"], z["to"], z["sb"], z["so"], int(z["n_sigs"]), z["mut"])
    else:
        corpus = cb.make_write_corpus(cl, args.items, seed=cb.MASTER_SEED + rank, batch_signer=gpu_signer, with_client_sig=True)
        if cache:
            np.savez(cache, tb=corpus.tbss_blob, to=corpus.tbss_off, sb=corpus.ss_blob, so=corpus.ss_off,
                     n_sigs=corpus.n_sigs, mut=corpus.mutation)
    t_corpus = time.time() - t0

    # keyring + quorum through the C ABI (clique of all replicas, AUTH rule: wotqs.go:36-70); one verifier context per
    # batch in flight (each owns its HIP streams and device arena)
    keys = [{"key_id": r.key_id, "entity_id": r.key_id, "pk_algo": r.algo, "usable_sign": True,
             "n": r.n.to_bytes(256, "big"), "e": r.e.to_bytes(3, "big")} for r in cl.replicas]
    f, mn, thr, suff = cb.quorum_numbers(n)
    n_ctx = max(1, args.inflight)
    ctxs = [ctx] + [Context(local_rank) for _ in range(n_ctx - 1)]
    qhs = []
    for cx in ctxs:
        cx.keyring_set(keys)
        qhs.append(cx.quorum_create([(f, mn, thr, suff, [r.key_id for r in cl.replicas])]))
    qh = qhs[0]

    d_tbs = torch.from_numpy(corpus.tbss_blob).to(dev)
    d_tbs_off = torch.from_numpy(corpus.tbss_off.astype(np.int64)).to(dev)
    d_ss = torch.from_numpy(corpus.ss_blob).to(dev)
    d_ss_off = torch.from_numpy(corpus.ss_off.astype(np.int64)).to(dev)
    outs = [(torch.zeros(args.items, dtype=torch.uint8, device=dev), torch.zeros(args.items, dtype=torch.int32, device=dev),
             torch.zeros(args.items, dtype=torch.uint8, device=dev)) for _ in ctxs]
    from bftkv_amd import dist as D
    torch.cuda.synchronize()

    def submit(i):
        cx, (e, nv, vd) = ctxs[i % n_ctx], outs[i % n_ctx]
        cx.collective_verify_dev(qhs[i % n_ctx], args.items, d_tbs.data_ptr(), d_tbs_off.data_ptr(), d_ss.data_ptr(), d_ss_off.data_ptr(),
                                 int(corpus.ss_off[-1]), e.data_ptr(), nv.data_ptr(), vd.data_ptr())

    step_rsa_ms, step_total_ms = [], []

    def complete(i):
        ctxs[i % n_ctx].sync()
        tm_i = ctxs[i % n_ctx].last_timing()        # HIP events recorded on the kernels' own streams during this step
        step_rsa_ms.append(tm_i["rsa"])
        step_total_ms.append(tm_i["total"])
        if world > 1:
            # per-write verdict bitmap (1 bit per write), all-gathered over RCCL/xGMI so that every
            # rank holds every verdict -- as every replica of the reference reaches every decision
            return D.allgather_verdicts(outs[i % n_ctx][0] == 0, args.items * world)
        return None

    def run(k):
        """k steps, up to n_ctx batches in flight: step i+1 is submitted before step i is waited for."""
        for i in range(k):
            submit(i)
            if i >= n_ctx - 1:
                complete(i - (n_ctx - 1))
        for i in range(max(0, k - (n_ctx - 1)), k):
            complete(i)

    run(args.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    del step_rsa_ms[:], step_total_ms[:]
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    timed_rsa_ms, timed_total_ms = list(step_rsa_ms), list(step_total_ms)      # the K timed steps
    # phase breakdown of an isolated call: a few non-overlapped calls on one context
    rsa_ms, tot_ms = [], []
    for _ in range(3):
        submit(0)
        complete(0)
        tm = ctx.last_timing()
        rsa_ms.append(tm["rsa"])
        tot_ms.append(tm["total"])
    d_err, d_nver, d_verdict = outs[0]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([corpus.n_sigs, args.items], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_sigs, total_items = int(cnt[0].item()), int(cnt[1].item())
    else:
        total_sigs, total_items = corpus.n_sigs, args.items

    counters = ctx.last_counters()
    err = d_err.cpu().numpy()
    nver = d_nver.cpu().numpy()

    if rank == 0:
        verifies_per_s = total_sigs * args.steps / elapsed
        verdicts_per_s = total_items * args.steps / elapsed
        rsa_avg_s = float(np.mean(timed_rsa_ms)) * 1e-3      # average k_rsa_modexp launch duration over the timed region
        alg_bytes = int(corpus.tbss_off[-1]) + corpus.n_sigs * RSA_BYTES + (args.items + 7) // 8
        achieved = alg_bytes / rsa_avg_s / 1e9
        traffic, traffic_src = measured_traffic("k_rsa_modexp") if args.items == 10000 and n == 64 else (None, None)
        out = {
            "metric": "pgp_rsa2048_signature_verifies_per_sec",
            "value": verifies_per_s,
            "unit": "verifies/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "%d-replica quorum, %d RSA-2048 signed writes per GPU (cfg2 of BASELINE.json), "
                                   "%d signature packets per GPU, 1.0%% corrupt / 0.5%% unknown issuer / 0.5%% duplicate / "
                                   "1.0%% one-short" % (n, args.items, corpus.n_sigs),
                       "replicas": n, "writes_per_gpu": args.items, "sigs_per_gpu": corpus.n_sigs,
                       "parallelism": "shard-by-write x%d, RCCL all-gather of verdict bitmaps" % world,
                       "batches_in_flight": n_ctx},
            "quorum_verdicts_per_sec": verdicts_per_s,
            "sufficient_fraction": float((err == 0).mean()),
            "pubkey_ops_per_step_per_gpu": int(counters["pubkey_ops"]),
            "kernel_ms": {"k_rsa_modexp": float(np.mean(timed_rsa_ms)), "step_device_span": float(np.mean(timed_total_ms)),
                          "measured": "HIP events of the %d timed steps (%d in flight)" % (len(timed_rsa_ms), n_ctx),
                          "isolated_call": {"total": float(np.mean(tot_ms)), "k_rsa_modexp": float(np.mean(rsa_ms)), "phases": tm}},
            "roofline": {"bound": "hbm", "kernel": "k_rsa_modexp", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "path is integer-VALU bound, not HBM bound (DESIGN.md); see int_mac"},
            "int_mac": {"achieved": counters["pubkey_ops"] * MADS_PER_VERIFY / rsa_avg_s, "peak": INT_MAC_PEAK,
                        "frac": counters["pubkey_ops"] * MADS_PER_VERIFY / rsa_avg_s / INT_MAC_PEAK,
                        "unit": "u32xu32+u64 MAC/s (v_mad_u64_u32 lanes)"},
            "corpus_build_s": t_corpus,
        }
        if world == 1:
            # the same batch handed over in HOST buffers (what a cgo caller does): H2D copy + pipeline + D2H of the verdicts.
            # Reported beside the headline, never as `value` (inputs resident in HBM).
            hb = []
            for _ in range(3):
                t_h = time.perf_counter()
                e_h, _, _ = ctx.collective_verify(qh, corpus.tbss_blob, corpus.tbss_off, corpus.ss_blob, corpus.ss_off)
                hb.append(time.perf_counter() - t_h)
            assert (e_h == err).all()
            hb_s = min(hb)
            out["host_buffers"] = {"ms_per_step": hb_s * 1e3, "verifies_per_sec": corpus.n_sigs / hb_s,
                                   "bytes_over_pcie": int(corpus.tbss_off[-1]) + int(corpus.ss_off[-1]) + 16 * args.items,
                                   "note": "pageable host memory in, verdicts out; best of 3"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cl, corpus, err, nver)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for cx in ctxs:
        cx.close()


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_pmc_summary.json,
    made by tools/profile_bench.sh + tools/summarize_profile.py: separate rocprofv3 --pmc passes over this
    same command, gfx950 FETCH_SIZE correction applied).  None when no summary is present."""
    import glob
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        best = p
    if not best:
        return None, None
    with open(best) as f:
        d = json.load(f)
    k = d["kernels"].get("bftkv::" + kernel) or d["kernels"].get("bftkv::" + kernel + "<19>")
    return (k["hbm_bytes_corrected"] if k else None), os.path.relpath(best, ROOT)


def effective_cores():
    """CPUs this process may really use: affinity mask and cgroup quota, not just os.cpu_count()."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(cl, corpus, gpu_err, gpu_nver):
    """The reference-shaped CPU path (oracle/c/oracle.c, 'port') on this box's host cores, on the same
    writes; also the bit-exact verdict check of the GPU results (checker role only)."""
    from oracle.cbind import COracle
    from tests import helpers as H
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    co = COracle()
    co.set_keyring(kr)
    co.set_quorum(q)
    cores = effective_cores()
    n = corpus.n_items
    # single-thread rate on a bounded sample (~5 s of CPU work), then all usable cores on the whole batch
    m1 = min(n, 2000)
    sub = (corpus.tbss_blob, corpus.tbss_off[:m1 + 1], corpus.ss_blob, corpus.ss_off[:m1 + 1])
    t0 = time.perf_counter()
    _, _, ops1 = co.collective_verify(*sub, n_threads=1)
    t1 = time.perf_counter() - t0
    best, best_threads = None, cores
    cands = sorted({cores} | {t for t in (16, 32, 64, 128, 256) if t <= (os.cpu_count() or 1)})
    for nt in cands:
        for _ in range(2):
            t0 = time.perf_counter()
            cerr, cnver, ops = co.collective_verify(corpus.tbss_blob, corpus.tbss_off, corpus.ss_blob, corpus.ss_off, n_threads=nt)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, best_threads = dt, nt
    cores = best_threads
    identical = bool((cerr == gpu_err).all() and (cnver == gpu_nver).all())
    return {"value": ops / best, "unit": "verifies/s", "cores": cores, "kind": "port",
            "sample": "all %d writes of the GPU batch (%d public-key ops after the reference's early exit at suff=%d), "
                      "best thread count %d of a sweep up to %d logical CPUs (usable per affinity/cgroup: %d); OpenSSL libcrypto "
                      "bignum/SHA (faster than Go math/big)" % (n, ops, cl.suff, cores, os.cpu_count() or 1, effective_cores()),
            "verdicts_per_sec": n / best,
            "single_thread_verifies_per_sec": ops1 / t1,
            "gpu_verdicts_identical_to_cpu": identical}


if __name__ == "__main__":
    main()
