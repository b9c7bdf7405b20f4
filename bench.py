#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1|2|3|4|5]

  (no --config)         cfg 2 as the headline `value`; on one GPU also `other_configs`: cfg 1 / 5 / 3 / 4, each as
                        `python bench.py --config N` in a process of its own (shorter timed regions), summarised into the ONE line
  --config 1            configs[0]: 4-replica clique, 100 RSA-2048 signed writes on the CPU path (the C restatement; plumbing, no GPU number)
  --config 2            configs[1]: 64-replica quorum, 10k RSA-2048 signed writes per GPU, batched verify; beside the resident
                        headline: `end_to_end` (the same batch handed over in pageable host memory: pipelined copy + verify, alone and
                        from three callers at once), `serving` (one Verify per call through the micro-batcher), `cpu_baseline`
  --config 3            configs[2]: 64-replica quorum (half DSA-2048/256), 100k signed read replies over 10k variables,
                        reply verdicts on the GPU -> maxTimestampedValue per variable (protocol/client.go:181-205)
  --config 4            configs[3]: 256 replicas, 1M-write storm sharded over the ranks (strong scaling), RCCL all-gather
                        of the per-write verdict bitmaps
  --config 5            configs[4]: threshold share-combine, 10k operations per scheme, sharded by operation

One "step" = one pass of the whole hot path over one batch that is ALREADY RESIDENT IN HBM when the timed region starts.
N > 1: one process per GPU.  Started under torchrun (RANK / WORLD_SIZE in the environment) it is one rank; started
plainly with --gpus N it re-executes itself under `python -m torch.distributed.run` with N ranks.  torch.distributed is
used for the rendezvous, the barriers and the max-over-ranks of the elapsed time only; the data-path exchange (all-gather of
verdict bitmaps) is the library's own RCCL call on the verifier's HIP stream (bftkv_gpu_allgather_errs_dev).

`--dry-run` (CPU, gloo) exercises the launcher, the sharding and the exchange step with the verify call stubbed out by the
corpus' expected verdicts: it is what tests/test_bench_launcher.py runs; it prints no performance number.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Integer roof.  v_mad_u64_u32 is a quarter-rate VALU op: 16 lanes per SIMD and clock.
INT_MAC_THEORETICAL = 256 * 4 * 16 * 2.4e9      # CUs x SIMDs x lanes/clk x max clock (MI355X_MICROARCH.md chip table) = 39.3 T/s
INT_MAC_MEASURED = 29.66e12    # tools/microbench/valu_rates.hip at 8 waves/SIMD: 4.64 issue cycles per wave-instruction at the 2.10 GHz the
                               # part sustains under an all-MAC load (profiles/r02_valu_issue_rates_microbench.txt)
RSA_BYTES = 291                # algorithmic bytes per RSA-2048 signature verify
DSA_BYTES = 99                 # per DSA-2048/256 signature verify
# limb MACs actually executed (mont28.h): a general Montgomery product = 76x76 (a*b) + 76x76 (m*n) = 11,552; a squaring forms
# only the triangles of a*a: 4 lanes x (4 x 190 + 1,444) = 8,816.  e = 65537: 16 squarings + 2 products.
MACS_PER_RSA_VERIFY = 16 * 8816 + 2 * 11552
LINE_MAX = 6144                # bytes of the ONE stdout line (round 5's had grown to 25 KB and the driver's parser returned nothing)


def macs_per_dsa_verify(table_bits):
    """Limb MACs of one DSA verification from fixed-base tables of `table_bits`-bit windows (k_dsa_modexp): g^u1 y^u2 is one table
    entry per window of u1 and of u2, the first taken as it is and the last stored in plain form -- 2 * ceil(256 / bits) - 1
    general Montgomery products (18 bits: 29, 16: 31, 19: 27, 8: 63) of 11,552 MACs."""
    if not table_bits:
        return 0
    return (2 * ((256 + table_bits - 1) // table_bits) - 1) * 11552


def macs_per_calculate_r(k=8, windows=64, win_bits=4, ent=15):
    """Limb MACs of one CalculateR (k_multiexp twice: prod_j Ri^lj with k bases, then the final power with one): per base one
    product into Montgomery form and ent - 1 table powers, one Montgomery one, win_bits squarings per window, one table product
    per base and window, one product leaving the domain."""
    general = (k * ent + 1 + windows * k + 1) + (ent + 1 + windows + 1)
    squarings = 2 * windows * win_bits
    return general * 11552 + squarings * 8816


# ------------------------------------------------------------------------------------------------------------------
# launcher and process group
# ------------------------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 200 for cfg 2 -- 0.6 s of timed region, long past the "
                    "fill and drain of the batches in flight --, 10 for the other configs)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=None, choices=(1, 2, 3, 4, 5), help="default: cfg 2 as the headline and, on one GPU, "
                    "cfg 1 / 3 / 4 / 5 behind it as `other_configs`; --config N runs that config alone with its full line")
    ap.add_argument("--no-other-configs", action="store_true", help="default run: the cfg-2 line only")
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds of CPU work the cpu_baseline leg may sweep thread counts for")
    ap.add_argument("--items", type=int, default=0, help="cfg 2: signed writes per GPU and step (default 10000); cfg 3: variables "
                    "(default 10000, ~10 replies each); cfg 4: writes of the whole storm (default 1000000); cfg 5: operations per scheme (10000)")
    ap.add_argument("--replicas", type=int, default=0, help="clique size (default 64; 256 for cfg 4)")
    ap.add_argument("--distinct", type=int, default=2500, help="cfg 4: distinctly signed writes the resident batch is tiled from")
    ap.add_argument("--chunk", type=int, default=125000, help="cfg 4: writes per verifier call (the resident batch)")
    ap.add_argument("--inflight", type=int, default=None, help="batches (steps) in flight per GPU, each on its own context; default 3 "
                    "(cfg 2, 3), 16 (cfg 5 with its 16 hardware queues: its CalculateR is a chain that leaves most issue slots of one step empty; 4 with the runtime's default 4 queues), "
                    "cfg 4 always runs one call at a time.  cfg 2: batches in flight per GPU, each on its own verifier context "
                    "(its own arena and streams).  With more than one, walk/parse and compare/tally/exchange of one step run under "
                    "the modexp of a neighbour -- what a fed verifier does; the machine-filling modexps themselves take turns "
                    "(the library's per-device turnstile), so a launch's duration still measures the kernel.  3 is the default: "
                    "under a saturating modexp the dozen small kernels of a step's tail and head only get SIMD slots when a round "
                    "of modexp blocks retires (every ~0.3 ms), which takes about one modexp's duration -- two in flight leave gaps "
                    "(3.03 ms per step), three do not (2.91; one: 3.2).  The line also carries the single-flight figures "
                    "(kernel_ms.single_flight, int_mac.single_flight)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-serving", action="store_true", help="cfg 2 / cfg 5: skip the one-call-per-operation leg (tools/serving/batcher_load.c, threshold_load.c)")
    ap.add_argument("--no-end-to-end", action="store_true", help="cfg 2: skip the host-buffer legs (profiling runs: their piece-sized launches "
                    "would mix into the per-kernel averages of the resident step)")
    ap.add_argument("--soak-seconds", type=float, default=6.0, help="after the timed region: keep running the same step, untimed for "
                    "the headline, for about this long (reported as `sustained`); an activity sampler with a period of seconds "
                    "otherwise never sees a timed region of tens of milliseconds.  0 disables")
    ap.add_argument("--corpus-cache", default="", help="path prefix of an .npz cache of the generated corpus (profiling reruns)")
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo: launcher, sharding and exchange step only (no GPU, no number)")
    ap.add_argument("--full-json", default=None, help="where the FULL record goes (sweeps, per-thread serving runs, timelines, notes, commands, "
                    "every other config's record): default bench_full.json next to bench.py (bench_full_cfgN.json with --config N).  stdout "
                    "carries ONE line of at most %d bytes -- the contract's keys, roofline, cpu_baseline, int_mac and a per-config summary" % LINE_MAX)
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = 200 if (args.config in (None, 2) and not args.dry_run) else 64 if args.config == 5 else 10
    if args.config == 5 and not args.dry_run:
        # Steps of cfg 5 are chains on too few waves to fill the chip (docs/history.md, round 4): what buys throughput is MANY steps in
        # flight, each on its own context, and their streams on hardware queues of their own.  The HIP runtime multiplexes all streams
        # onto GPU_MAX_HW_QUEUES queues (4 by default; read when the runtime starts, so it is set here, before torch / HIP are
        # loaded).  Measured (tools/runs/r04/gpu_r4z.sh): 4 queues x 4 steps in flight 7.6 ms per step, 16 x 8: 6.2, 16 x 12: 5.8, 16 x 16:
        # 5.7 (24 or 32 queues are worse: a context's three streams then collide differently).
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    if args.inflight is None:
        many = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) >= 16
        args.inflight = (16 if many else 4) if args.config == 5 else 3
    return args


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch(args):
    """--gpus N without a torchrun environment: start N ranks of this script, one per GPU, and pass their output through."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class Dist:
    """Rank bookkeeping; torch.distributed only for rendezvous, barriers and the reduction of the timing."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dry = args.dry_run
        self.dist = None
        # "library": bftkv_gpu_allgather_errs_dev on the verifier's stream; "torch": the fallback of comm_init (BFTKV_BENCH_EXCHANGE=torch forces it)
        self.exchange = "torch" if os.environ.get("BFTKV_BENCH_EXCHANGE") == "torch" else "library"
        if self.dry:
            self.dev = torch.device("cpu")
        else:
            self.dev = torch.device("cuda", self.local_rank)
            torch.cuda.set_device(self.dev)
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.dry:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=self.dev)
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def sync(self):
        if not self.dry:
            self.torch.cuda.synchronize()

    def max_float(self, v):
        if not self.dist:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_ints(self, vals):
        if not self.dist:
            return [int(v) for v in vals]
        t = self.torch.tensor(list(vals), dtype=self.torch.int64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [int(v) for v in t.tolist()]

    def max_int(self, v):
        if not self.dist:
            return int(v)
        t = self.torch.tensor([int(v)], dtype=self.torch.int64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.item())

    def gather_ints(self, vals):
        """every rank's list of ints, rank-major: [world][len(vals)]"""
        if not self.dist:
            return [[int(v) for v in vals]]
        t = self.torch.tensor(list(vals), dtype=self.torch.int64, device=self.dev)
        out = self.torch.empty(self.world * t.numel(), dtype=self.torch.int64, device=self.dev)
        self.dist.all_gather_into_tensor(out, t)
        return out.view(self.world, -1).tolist()

    def comm_init(self, ctxs):
        """The library's own RCCL communicator on every verifier context: rank 0 draws the unique id
        (bftkv_gpu_comm_unique_id), torch.distributed carries it to the other ranks, every rank joins."""
        from bftkv_amd import Context
        torch = self.torch
        if self.exchange == "torch":
            return
        failed = None
        try:
            for cx in ctxs:
                uid = torch.zeros(128, dtype=torch.uint8)
                if self.rank == 0:
                    uid = torch.from_numpy(Context.comm_unique_id().copy())
                if self.dist:
                    u = uid.to(self.dev)
                    self.dist.broadcast(u, src=0)
                    uid = u.cpu()
                cx.comm_init(self.world, self.rank, uid.numpy())
            # before anything is timed: every rank's row of a gathered, rank-stamped buffer must be that rank's, on every context
            # (bftkv_gpu_comm_selftest; collective).
            for k, cx in enumerate(ctxs):
                cx.comm_selftest(4096)
        except Exception as e:      # noqa: BLE001
            failed = e
            sys.stderr.write("bench.py: rank %d of %d: the library's own RCCL exchange could not be set up: %s\n" % (self.rank, self.world, e))
        # Every rank learns whether ANY rank failed.  The multi-GPU path has never met more than one GPU (one per box here): if the
        # library's communicator cannot be had, the exchange step falls back to torch.distributed's all-gather of the same bitmaps
        # (still RCCL over xGMI, but host-synchronised per step) and the line says so -- a number with a caveat instead of none.
        if self.world > 1:
            if self.sum_ints([1 if failed else 0])[0]:
                self.exchange = "torch"
                if self.rank == 0:
                    sys.stderr.write("bench.py: exchange step through torch.distributed.all_gather_into_tensor (fallback)\n")
                return
        elif failed:
            raise failed
        if self.world > 1:
            path, pre = Context.comm_library()
            seen = self.sum_ints([1])[0]
            if seen != self.world:
                raise RuntimeError("bench.py: %d ranks answered, world size %d" % (seen, self.world))
            if self.rank == 0:
                sys.stderr.write("bench.py: RCCL for the exchange step: %s (%s)\n" % (path, "already loaded by the process" if pre else "loaded by libbftkv_gpu"))

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------
# shared pieces of the verify workloads (cfg 2, 3, 4)
# ------------------------------------------------------------------------------------------------------------------
def gpu_signers(ctx, cl):
    """RSA signatures and DSA nonce powers of the corpus generator computed on this GPU (generic modexp kernel)."""
    from corpus import build as cb
    mods, exps = cb.signer_tables(cl)   # replicas in order, then the client

    def rsa(em, key_index):
        return ctx.modexp(em, key_index.astype(np.uint32), mods, exps)
    dsa = None
    if any(r.algo == cb.PK_DSA for r in cl.replicas):
        ps, gs = cb.dsa_pow_tables(cl)

        def dsa(key_index, ks):
            base = np.ascontiguousarray(gs[key_index])
            ex = np.stack([np.frombuffer(int(k).to_bytes(32, "big"), dtype=np.uint8) for k in ks])
            out = ctx.modexp_ops(base, key_index.astype(np.uint32), ps, ex)
            return [int.from_bytes(out[i].tobytes(), "big") for i in range(len(ks))]
    return rsa, dsa


def abi_keys_of(cl):
    """bftkv_gpu_pubkey records of the clique members (what the cgo shim extracts from the node's keyring)."""
    from corpus import build as cb
    keys = []
    for r in cl.replicas:
        if r.algo == cb.PK_RSA:
            keys.append({"key_id": r.key_id, "entity_id": r.key_id, "pk_algo": r.algo, "usable_sign": True,
                         "n": r.n.to_bytes(256, "big"), "e": r.e.to_bytes(3, "big")})
        else:
            keys.append({"key_id": r.key_id, "entity_id": r.key_id, "pk_algo": r.algo, "usable_sign": True,
                         "n": r.p.to_bytes(256, "big"), "e": r.q.to_bytes(32, "big"), "g": r.g.to_bytes(256, "big"),
                         "y": r.y.to_bytes(256, "big")})
    return keys


class Verifier:
    """n_ctx verifier contexts over one resident batch: submit(i) launches the pipeline and the exchange step of batch i on
    context i % n_ctx, complete(i) waits for it and collects the HIP-event durations of that step."""

    def __init__(self, D, cl, n_items, tb, to, sb, so, n_ctx=1, ctx0=None, ss_len=None):
        from bftkv_amd import Context
        from corpus import build as cb
        self.D, self.n_items, self.n_ctx = D, n_items, n_ctx
        # every rank contributes the same number of bitmap bits: the LARGEST shard (cfg 3's ranks hold different reply counts; an
        # all-gather with per-rank counts would be a different collective on every rank)
        self.slots = D.max_int(n_items)
        self.rsa_ms, self.dsa_ms, self.hash_ms, self.total_ms = [], [], [], []
        self.gathers = 0
        self.ss_len = int(so[-1]) if ss_len is None else ss_len
        torch = D.torch
        if D.dry:
            self.ctxs = []
            return
        # one root context holds the key table (and the DSA window tables: GBs per key); the other batches in flight run on
        # FORKS of it (bftkv_gpu_ctx_fork: own streams, arena, events, mailbox; the root's keys and quorum handles)
        root = ctx0 or Context(D.local_rank)
        f, mn, thr, suff = cb.quorum_numbers(cl.n)
        root.keyring_set(abi_keys_of(cl))
        qh = root.quorum_create([(f, mn, thr, suff, [r.key_id for r in cl.replicas])])
        self.ctxs = [root] + [root.fork() for _ in range(n_ctx - 1)]
        self.qhs = [qh] * n_ctx
        D.comm_init(self.ctxs)
        dev = D.dev
        self.d_tbs = torch.from_numpy(tb).to(dev) if isinstance(tb, np.ndarray) else tb
        self.d_ss = torch.from_numpy(sb).to(dev) if isinstance(sb, np.ndarray) else sb
        self.d_tbs_off = torch.from_numpy(to.astype(np.int64)).to(dev) if isinstance(to, np.ndarray) else to
        self.d_ss_off = torch.from_numpy(so.astype(np.int64)).to(dev) if isinstance(so, np.ndarray) else so
        nbytes = (self.slots + 7) // 8
        self.outs = [(torch.zeros(n_items, dtype=torch.uint8, device=dev), torch.zeros(n_items, dtype=torch.int32, device=dev),
                      torch.zeros(n_items, dtype=torch.uint8, device=dev), torch.zeros(D.world * nbytes, dtype=torch.uint8, device=dev))
                     for _ in self.ctxs]
        torch.cuda.synchronize()

    def submit(self, i):
        k = i % self.n_ctx
        cx, (e, nv, vd, bits) = self.ctxs[k], self.outs[k]
        cx.collective_verify_dev(self.qhs[k], self.n_items, self.d_tbs.data_ptr(), self.d_tbs_off.data_ptr(), self.d_ss.data_ptr(),
                                 self.d_ss_off.data_ptr(), self.ss_len, e.data_ptr(), nv.data_ptr(), vd.data_ptr())
        # exchange step, enqueued behind the tally on the verifier's own stream: no host synchronisation in between
        if self.D.exchange == "library":
            cx.allgather_errs_dev(e.data_ptr(), self.n_items, self.slots, bits.data_ptr())
        else:       # fallback (Dist.comm_init): wait for the verdicts, pack and gather with torch.distributed -- same bitmap layout
            from bftkv_amd import dist as BD
            cx.sync()
            local = BD.pack_verdicts(e == 0, self.slots)
            if self.D.dist:
                self.D.dist.all_gather_into_tensor(bits, local)
            else:
                bits.copy_(local)
            self.D.torch.cuda.current_stream().synchronize()
        self.gathers += 1

    def complete(self, i):
        k = i % self.n_ctx
        self.ctxs[k].sync()
        tm = self.ctxs[k].last_timing()        # HIP events recorded on the kernels' own streams during this step
        self.rsa_ms.append(tm["rsa"]); self.dsa_ms.append(tm["dsa"]); self.hash_ms.append(tm["hash"]); self.total_ms.append(tm["total"])
        self.last_tm = tm

    def run(self, k):
        """k batches, up to n_ctx in flight: batch i+1 is submitted before batch i is waited for."""
        for i in range(k):
            self.submit(i)
            if i >= self.n_ctx - 1:
                self.complete(i - (self.n_ctx - 1))
        for i in range(max(0, k - (self.n_ctx - 1)), k):
            self.complete(i)

    def reset_timing(self):
        del self.rsa_ms[:], self.dsa_ms[:], self.hash_ms[:], self.total_ms[:]

    def results(self, k=0):
        e, nv, vd, bits = self.outs[k]
        return e.cpu().numpy(), nv.cpu().numpy(), bits.cpu().numpy()

    def check_gather(self, err, bits):
        """Every rank's row of the gathered bitmap must be that rank's verdicts; this rank checks its own row and the
        population count of all rows against the all-reduced count of accepted writes."""
        nbytes = (self.slots + 7) // 8
        rows = np.unpackbits(bits.reshape(self.D.world, nbytes), axis=1, bitorder="little")[:, :self.slots]     # (padding bits are zero)
        own_ok = bool((rows[self.D.rank][:self.n_items] == (err == 0)).all())
        total_ok = self.D.sum_ints([int((err == 0).sum())])[0]
        return own_ok and int(rows.sum()) == total_ok

    def close(self):
        for cx in reversed(self.ctxs):      # forks before their root
            cx.close()


def dry_exchange(D, ok_local, slots):
    """Stub of bftkv_gpu_allgather_errs_dev for --dry-run: same bitmap layout (bftkv_amd/dist.py), gloo instead of RCCL."""
    from bftkv_amd import dist as BD
    torch = D.torch
    bits = BD.pack_verdicts(torch.from_numpy(ok_local.astype(np.uint8)), slots)
    if D.world == 1:
        return bits.numpy()
    out = torch.empty(D.world * bits.numel(), dtype=torch.uint8)
    D.dist.all_gather_into_tensor(out, bits)
    return out.numpy()


def constructed_ok(lo, hi, salt):
    """--dry-run: the verdict 'constructed' for the unit (write, reply) with GLOBAL index i in [lo, hi) -- a fixed function of
    the index, so that a test can rebuild the verdict vector of the whole job and compare it with what the ranks gathered."""
    i = np.arange(lo, hi, dtype=np.uint64)
    return (((i * np.uint64(2654435761) + np.uint64(salt)) >> np.uint64(7)) % np.uint64(61)) != 0


def verdict_digest(ok):
    import hashlib
    return hashlib.sha256(np.packbits(np.asarray(ok, dtype=np.uint8), bitorder="little").tobytes()).hexdigest()


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


def c_oracle_for(cl):
    from oracle.cbind import COracle
    from tests import helpers as H
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    co = COracle()
    co.set_keyring(kr)
    co.set_quorum(q)
    return co


CPU_SWEEP = {}          # what the last cpu_collective call swept (goes into cpu_baseline.sweep)


def cpu_collective(cl, tb, to, sb, so, budget_s=25.0):
    """The reference-shaped CPU path (oracle/c/oracle.c: per-signature re-hash, per-signature IsSufficient, early exit) on
    this box's host cores over the given items: checker of the GPU verdicts and the reported CPU baseline.  Thread counts are
    swept within a time budget -- but never fewer than two of them: the usable cores and the next wider candidate are always
    timed, so that a short budget cannot leave the baseline at an untuned first guess; what was timed and whether the budget cut
    the sweep short is recorded (CPU_SWEEP).  Returns (err, n_verified, public-key ops, best seconds, threads, single-thread ops/s)."""
    co = c_oracle_for(cl)
    n = len(to) - 1
    m1 = max(1, min(n, n // 50))
    t0 = time.perf_counter()
    _, _, ops1 = co.collective_verify(tb, to[:m1 + 1], sb, so[:m1 + 1], n_threads=1)
    t1 = time.perf_counter() - t0
    cores = effective_cores()
    cands = sorted({cores} | {t for t in (16, 32, 64, 128, 256) if t <= (os.cpu_count() or 1)})
    cands = [t for t in cands if t >= cores] or [cores]          # fewer threads than usable cores are never the best
    best = None
    spent = 0.0
    res = None
    timed = []
    for k, nt in enumerate(cands):
        if k >= 2 and spent + best[0] > budget_s:
            break
        t0 = time.perf_counter()
        cerr, cnver, ops = co.collective_verify(tb, to, sb, so, n_threads=nt)
        dt = time.perf_counter() - t0
        spent += dt
        timed.append({"threads": nt, "seconds": dt})
        if best is None or dt < best[0]:
            best = (dt, nt)
        res = (cerr, cnver, ops)
    CPU_SWEEP.clear()
    # untuned: the budget ended the sweep while the widest count timed was still the best one (a wider one might have been better)
    CPU_SWEEP.update({"candidates": cands, "timed": timed, "truncated_by_budget": len(timed) < len(cands), "budget_s": budget_s,
                      "untuned": bool(len(timed) < len(cands) and best[1] == timed[-1]["threads"])})
    return res[0], res[1], res[2], best[0], best[1], ops1 / t1


def measured_traffic(cfg, kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary of this config (profiles/*_pmc_summary.json,
    made by tools/profile_bench.sh + tools/summarize_profile.py: separate rocprofv3 --pmc passes over this same command,
    gfx950 FETCH_SIZE correction applied).  None when no summary is present."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cfg%d_pmc_summary.json" % cfg)))
    if not cands and cfg == 2:
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r01_v?_pmc_summary.json")))
    if not cands:
        return None, None
    with open(cands[-1]) as f:
        d = json.load(f)
    for name, k in d["kernels"].items():
        if name.split("<")[0] == "bftkv::" + kernel:
            return k["hbm_bytes_corrected"], os.path.relpath(cands[-1], ROOT)
    return None, os.path.relpath(cands[-1], ROOT)


def roofline(cfg, kernel, alg_bytes, launch_ms, note, **extra):
    """`launch_ms` is the dominant kernel's SINGLE-FLIGHT launch duration (HIP events on its own stream, nothing of a neighbouring step
    beside it) wherever the timed region keeps steps in flight whose kernels overlap: a span measured across overlapped launches is
    not the kernel's cost and goes under `in_flight_span_ms` (extra)."""
    achieved = alg_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms else 0.0
    traffic, src = measured_traffic(cfg, kernel)
    out = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": traffic, "traffic_source": src, "algorithmic_bytes_per_launch": int(alg_bytes),
           "launch_ms": launch_ms, "note": note}
    out.update(extra)
    return out


def int_mac(macs, launch_ms, sclk_mhz=None):
    a = macs / (launch_ms * 1e-3) if launch_ms else 0.0
    out = {"achieved": a, "peak": INT_MAC_MEASURED, "frac": a / INT_MAC_MEASURED, "unit": "u32xu32+u64 MAC/s (v_mad_u64_u32 lanes)",
           "peak_source": "tools/microbench/valu_rates.hip on MI355X, 8 waves/SIMD (profiles/r02_valu_issue_rates_microbench.txt): 4.64 issue "
                          "cycles per wave64 v_mad_u64_u32 at the 2.10 GHz (s_memtime / s_memrealtime) the part sustains under an all-MAC load",
           "peak_theoretical": INT_MAC_THEORETICAL, "frac_of_theoretical": a / INT_MAC_THEORETICAL,
           "theoretical_source": "256 CU x 4 SIMD x 16 lanes/clk (quarter-rate VALU op) x 2.4 GHz max clock"}
    if sclk_mhz:
        out["sclk_mhz_in_kernel"] = sclk_mhz
        out["sclk_source"] = "s_memtime / s_memrealtime over the first wave of k_rsa_modexp (bftkv_gpu_last_sclk_mhz)"
        out["peak_at_observed_clock"] = 256 * 4 * 64 / 4.64 * sclk_mhz * 1e6
        out["frac_at_observed_clock"] = a / out["peak_at_observed_clock"]
        out["observed_clock_note"] = "4.64 issue cycles per wave64 v_mad_u64_u32 (microbench) at the clock this launch ran at"
    return out


def int_mac_block(macs_per_step, ms_step, launch_ms, single_ms, sclk_mhz, in_flight):
    """cfg 2's integer-MAC line.  Main figures: the MACs of a step over the WALL time of a step in the timed region -- what the chip
    sustains over everything (walk, parse, hashing, tallies and the exchange included).  Beside it: the same MACs over the average
    duration of the timed region's k_rsa_modexp launches (each with the small kernels of the neighbouring batches beside it) and
    over the kernel's single-flight launch duration (the kernel with only its own hash stream beside it)."""
    out = int_mac(macs_per_step, ms_step, sclk_mhz)
    out["basis"] = "MACs of the public-key kernels per step / ms_per_step of the timed region (%d batches in flight)" % in_flight
    pl = int_mac(macs_per_step, launch_ms)
    out["per_launch_in_timed_region"] = {"launch_ms": launch_ms, "achieved": pl["achieved"], "frac": pl["frac"], "frac_of_theoretical": pl["frac_of_theoretical"]}
    if single_ms:
        sf = int_mac(macs_per_step, single_ms)
        out["single_flight"] = {"launch_ms": single_ms, "achieved": sf["achieved"], "frac": sf["frac"], "frac_of_theoretical": sf["frac_of_theoretical"]}
    return out


def reference_pubkey_ops(st, item, err, nver, n_items):
    """Public-key operations the REFERENCE performs on this batch (crypto_pgp.go:485-500): per item the packets it examines
    -- all of them, or up to the packet that made IsSufficient true -- that reach rsa.VerifyPKCS1v15 / dsa.Verify, i.e. end in
    ST_OK or ST_BAD_SIG.  Computed from the verifier's per-packet statuses, error bytes and exit counts (identical to the CPU
    restatement's where that ran: cpu_baseline.gpu_verdicts_identical_to_cpu).  The two-phase plan verifies a few packets more
    (its margin); they are not counted here."""
    ok = (st == 0)
    cum = np.cumsum(ok, dtype=np.int64)
    first = np.searchsorted(item, np.arange(n_items), side="left")
    before_item = np.concatenate([[0], cum])[first]              # OK packets before the item's first record
    ok_before = cum - ok - before_item[item]                      # OK packets of the same item in front of this record
    examined = (err[item] != 0) | (ok_before < nver[item].astype(np.int64))
    return int((examined & (ok | (st == 8))).sum())


def soak(D, run_steps, seconds, ms_per_step, units_per_step):
    """The same step, repeated for ~`seconds` after the timed region (see --soak-seconds)."""
    if seconds <= 0 or ms_per_step <= 0:
        return None
    k = max(1, int(seconds * 1e3 / ms_per_step))
    D.barrier(); D.sync()
    t0 = time.perf_counter()
    run_steps(k)
    D.sync(); D.barrier()
    dt = D.max_float(time.perf_counter() - t0)
    return {"steps": k, "seconds": dt, "ms_per_step": dt / k * 1e3, "value": units_per_step * k / dt,
            "note": "same step as the timed region, run right after it; not the headline"}


def timed_region(D, run, steps, warmup, reset=None):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
    run(warmup)
    D.barrier()
    D.sync()
    if reset:
        reset()
    t0 = time.perf_counter()
    run(steps)
    D.sync()
    D.barrier()
    return D.max_float(time.perf_counter() - t0)


def base_line(args, D, metric, unit, value, elapsed, dtype, workload, extra_cfg, scaling="weak"):
    cfg = {"workload": workload, "parallelism": "one process per GPU x%d" % D.world}
    cfg.update(extra_cfg)
    return {"metric": metric, "value": value, "unit": unit, "n_gpus": D.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": dtype, "data": "synthetic", "config": cfg}


# ------------------------------------------------------------------------------------------------------------------
# cfg 2: 64-replica quorum, 10k RSA-2048 signed writes per GPU
# ------------------------------------------------------------------------------------------------------------------
def load_or_make(args, tag, rank, make):
    """corpus cache for profiling reruns (--corpus-cache): arrays of a WriteCorpus-like object"""
    if not args.corpus_cache:
        return make()
    path = "%s.%s.r%d.npz" % (args.corpus_cache, tag, rank)
    if os.path.exists(path):
        z = np.load(path, allow_pickle=False)
        return {k: z[k] for k in z.files}
    d = make()
    np.savez(path, **d)
    return d


def cert_serving_leg(threads="1,64,256", clients=64):
    """One Issuer(sig) + VerifyWithCertificate per CALL for clients OUTSIDE the server's keyring -- Server.sign's verification
    (protocol/server.go:199-207), bftkv_gpu_batcher_cert_verify -- measured like serving_leg: plain-C load generator
    (tools/serving/cert_load.c) in its own process on a corpus signed on the CPU (tools/serving/make_cert_corpus.py).  None / an
    "error" entry when a tool is missing or fails: a side measurement never takes the line down."""
    import shutil
    import tempfile
    if shutil.which("gcc") is None:
        return None
    tmp = tempfile.mkdtemp(prefix="bftkv_cert_serving")
    try:
        exe, path, lib_dir = os.path.join(tmp, "cert_load"), os.path.join(tmp, "certs.bin"), os.path.join(ROOT, "bftkv_amd")
        cc = subprocess.run(["gcc", "-O2", "-std=gnu99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "serving", "cert_load.c"),
                             "-L", lib_dir, "-lbftkv_gpu", "-lpthread", "-Wl,-rpath," + lib_dir, "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if cc.returncode != 0:
            return {"error": "gcc: " + cc.stderr.decode(errors="replace")[-300:]}
        mk = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "serving", "make_cert_corpus.py"), path, str(clients)], stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, timeout=120)
        if mk.returncode != 0:
            return {"error": "make_cert_corpus: " + mk.stderr.decode(errors="replace")[-300:]}
        r = subprocess.run([exe, path, "256", "0", threads, "1.0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        if r.returncode != 0:
            return {"error": "cert_load rc=%d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:])}
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        return {"what": "one bftkv_gpu_batcher_cert_verify per call (%d clients outside the keyring, %d-byte certificates, RSA-2048): first sight "
                        "of a certificate = compound call on the root (registration + what ReadEntity verifies), later requests = one staged "
                        "signature verification on a lane; every answer checked" % (d["clients"], d["certificate_bytes"]),
                "first_sight_ms_per_certificate": d["first_sight"]["ms_per_certificate"], "wrong_answers_first_sight": d["first_sight"]["wrong"],
                "runs": [{"caller_threads": x["threads"], "calls_per_s": x["calls_per_s"], "latency_ms": x["latency_ms"], "wrong_answers": x["wrong"],
                          "device_calls": x["device_calls"]} for x in d["runs"]],
                "tool": "tools/serving/cert_load.c"}
    except Exception as e:      # noqa: BLE001
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def serving_leg(args, cl, z, want_ok, n_writes=4096, threads="1,64,256", lanes=0):
    """One CollectiveSignature.Verify per CALL from many caller threads through the micro-batcher -- the shape the reference
    actually has (one goroutine per request, transport/http/http.go:85,143 -> protocol/server.go:562-620) -- measured by the
    plain-C load generator tools/serving/batcher_load.c in its own process, on the first writes of this run's batch.  Reported
    beside the headline (resident batches), never as `value`.  None when there is no C compiler or the tool fails."""
    import shutil
    import struct
    import tempfile
    if shutil.which("gcc") is None:
        return None
    from corpus import build as cb
    n = min(n_writes, len(z["to"]) - 1)
    tmp = tempfile.mkdtemp(prefix="bftkv_serving")
    try:
        exe = os.path.join(tmp, "batcher_load")
        lib_dir = os.path.join(ROOT, "bftkv_amd")
        cc = subprocess.run(["gcc", "-O2", "-std=gnu99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "serving", "batcher_load.c"),
                             "-L", lib_dir, "-lbftkv_gpu", "-lpthread", "-Wl,-rpath," + lib_dir, "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if cc.returncode != 0:
            return {"error": "gcc: " + cc.stderr.decode(errors="replace")[-300:]}
        f, mn, thr, suff = cb.quorum_numbers(cl.n)
        path = os.path.join(tmp, "load.bin")
        with open(path, "wb") as fh:
            fh.write(struct.pack("<I", cl.n))
            for r in cl.replicas:
                fh.write(struct.pack("<Q", r.key_id) + r.n.to_bytes(256, "big") + r.e.to_bytes(4, "big"))
            fh.write(struct.pack("<iiiiI", f, mn, thr, suff, n))
            fh.write(z["to"][:n + 1].astype("<u8").tobytes() + z["so"][:n + 1].astype("<u8").tobytes())
            fh.write(z["tb"][:int(z["to"][n])].tobytes() + z["sb"][:int(z["so"][n])].tobytes())
            fh.write(want_ok[:n].astype(np.uint8).tobytes())
        r = subprocess.run([exe, path, "256", "0", str(lanes), threads], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        if r.returncode != 0:
            return {"error": "batcher_load rc=%d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:])}
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        sigs_per_write = float(z["so"][n]) / n / 287.0
        runs = [{"caller_threads": x["threads"], "verify_calls_per_s": x["verify_calls_per_s"],
                 "signatures_per_s": x["verify_calls_per_s"] * float(np.mean(z["sig_count"][:n])) if "sig_count" in z else None,
                 "latency_ms": x["latency_ms"], "wrong_answers": x["wrong"], "device_calls": x["device_calls"],
                 "cpu_cores_busy": x["cpu_cores_busy"]} for x in d["runs"]]
        return {"what": "one bftkv_gpu_batcher_collective_verify per call (one write, ~%.0f signature packets, %.1f KB payload) from N caller "
                        "threads, plain-C load generator in its own process, 2 s per point after 0.7 s of warm-up; every answer checked "
                        "against the corpus' constructed verdict" % (sigs_per_write, float(z["to"][n]) / n / 1e3),
                "lanes": d["lanes"], "max_items_per_batch": d["max_items"], "writes": n, "runs": runs,
                "usable_host_cores": effective_cores(), "tool": "tools/serving/batcher_load.c"}
    except Exception as e:      # noqa: BLE001  (a side measurement must not take the bench line down)
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def write_corpus_arrays(c):
    return {"tb": c.tbss_blob, "to": c.tbss_off, "sb": c.ss_blob, "so": c.ss_off, "n_sigs": np.array(c.n_sigs),
            "expected_valid": c.expected_valid, "sig_count": c.sig_count}


def bench_cfg2(args, D):
    from corpus import build as cb
    n = args.replicas or 64
    items = args.items or 10000
    cl = cb.make_cluster(n)
    t0 = time.time()
    ctx0 = None
    if D.dry:
        signer = None
    else:
        from bftkv_amd import Context
        ctx0 = Context(D.local_rank)
        signer, _ = gpu_signers(ctx0, cl)
    rates = {cb.MUT_ONE_SHORT: 0.3, cb.MUT_BAD_MPI: 0.2} if D.dry else None      # the dry run's few items must not all agree
    z = load_or_make(args, "cfg2.n%d.i%d" % (n, items), D.rank, lambda: write_corpus_arrays(
        cb.make_write_corpus(cl, items, seed=cb.MASTER_SEED + D.rank, batch_signer=signer, with_client_sig=True, mutation_rates=rates)))
    n_sigs = int(z["n_sigs"])
    t_corpus = time.time() - t0
    V = Verifier(D, cl, items, z["tb"], z["to"], z["sb"], z["so"], n_ctx=max(1, args.inflight), ctx0=ctx0)
    want_ok = z["expected_valid"] >= cl.suff

    if D.dry:
        gathered = []

        def run(k):
            for _ in range(k):
                gathered.append(dry_exchange(D, want_ok, items))
                V.gathers += 1
        elapsed = timed_region(D, run, args.steps, args.warmup)
        rows = np.unpackbits(gathered[-1].reshape(D.world, -1), axis=1, bitorder="little")[:, :items]
        tot = D.sum_ints([int(want_ok.sum())])[0]
        return {"dry_run": True, "config": 2, "n_gpus": D.world, "world_size": D.world, "steps": args.steps, "warmup": args.warmup,
                "parallelism": "shard-by-write x%d, all-gather of verdict bitmaps" % D.world, "allgathers_in_step_loop": V.gathers, "gather_rows": int(rows.shape[0]),
                "gathered_ok": int(rows.sum()), "sum_of_rank_ok": tot,
                "own_row_matches": bool((rows[D.rank] == want_ok).all()), "elapsed_s": elapsed} if D.rank == 0 else None

    V.run(V.n_ctx)               # one untimed call per verifier context: its arena is allocated at its first call
    elapsed = timed_region(D, V.run, args.steps, args.warmup, V.reset_timing)
    sclk = V.ctxs[0].last_sclk_mhz()
    timed_rsa, timed_total, timed_hash = list(V.rsa_ms), list(V.total_ms), list(V.hash_ms)
    iso = []
    for _ in range(3):           # phase breakdown of an isolated call: non-overlapped calls on one context
        V.submit(0); V.complete(0)
        iso.append(dict(V.last_tm))
    err, nver, bits = V.results(0)
    gather_ok = V.check_gather(err, bits)
    counters = V.ctxs[0].last_counters()
    st_all, st_item = V.ctxs[0].last_statuses()
    ref_ops = reference_pubkey_ops(st_all, st_item, err, nver, items)
    total_sigs, total_items, total_ref_ops, total_done_ops = D.sum_ints([n_sigs, items, ref_ops, int(counters["pubkey_ops"])])
    ms_step = elapsed / args.steps * 1e3
    sustained = soak(D, V.run, args.soak_seconds, ms_step, total_ref_ops)
    if D.rank == 0:
        rsa_ms = float(np.mean(timed_rsa))
        iso_rsa = float(np.mean([t["rsa"] for t in iso]))
        alg_bytes = int(z["to"][-1]) + ref_ops * RSA_BYTES + (items + 7) // 8      # payload once + 291 B per verify + 1 bit
        out = base_line(args, D, "pgp_rsa2048_signature_verifies_per_sec", "verifies/s", total_ref_ops * args.steps / elapsed, elapsed, "u32",
                        "%d-replica quorum, %d RSA-2048 signed writes per GPU (cfg2 of BASELINE.json), %d signature packets per GPU, "
                        "1.0%% corrupt / 0.5%% unknown issuer / 0.5%% duplicate / 1.0%% one-short" % (n, items, n_sigs),
                        {"replicas": n, "writes_per_gpu": items, "sigs_per_gpu": n_sigs, "batches_in_flight": V.n_ctx,
                         "parallelism": "shard-by-write x%d, RCCL all-gather of verdict bitmaps on the verifier's stream" % D.world})
        out.update({
            "value_counts": "RSA-2048 public-key verifications on the REFERENCE's operation count: per write the packets "
                            "PGPCollectiveSignature.Verify examines before IsSufficient stops it (crypto_pgp.go:485-500) that reach "
                            "rsa.VerifyPKCS1v15; the verifier performs pubkey_ops_per_step_per_gpu (its plan margin on top), parses and "
                            "hash-tag checks every packet of the batch",
            "reference_pubkey_ops_per_step_per_gpu": ref_ops,
            "pubkey_ops_per_step_per_gpu": int(counters["pubkey_ops"]),
            "pubkey_ops_performed_per_sec": total_done_ops * args.steps / elapsed,
            "packets_per_sec": total_sigs * args.steps / elapsed,
            "quorum_verdicts_per_sec": total_items * args.steps / elapsed,
            "sufficient_fraction": float((err == 0).mean()),
            "verdicts_match_construction": bool(((err == 0) == want_ok).all()),
            "allgather": {"calls_in_timed_region": args.steps, "bytes_per_rank": (items + 7) // 8, "rows_consistent": gather_ok,
                          "via": "bftkv_gpu_allgather_errs_dev (library RCCL, verifier stream)" if D.exchange == "library" else
                                 "torch.distributed.all_gather_into_tensor (FALLBACK: the library's communicator could not be set up; host-synchronised per step)"},
            "kernel_ms": {"k_rsa_modexp": rsa_ms, "k_rsa_modexp_min_max": [float(np.min(timed_rsa)), float(np.max(timed_rsa))],
                          "hash_stream": float(np.mean(timed_hash)), "step_device_span": float(np.mean(timed_total)),
                          "step_over_modexp": ms_step / iso_rsa if iso_rsa else None,
                          "step_over_modexp_is": "ms_per_step of the timed region over the kernel's single-flight launch duration",
                          "measured": "HIP events of the %d timed steps (%d in flight; the modexps of different contexts take turns "
                                      "at the library's turnstile, the duration runs from a launch's turn to its end)" % (len(timed_rsa), V.n_ctx),
                          "single_flight": {k: float(np.mean([t[k] for t in iso])) for k in iso[0]},
                          "single_flight_is": "3 non-overlapped calls on one context right after the timed region"},
            "roofline": roofline(2, "k_rsa_modexp", alg_bytes, iso_rsa or rsa_ms, "path is integer-VALU bound, not HBM bound (DESIGN.md); see "
                                 "int_mac; launch_ms = the kernel's single-flight launch duration (HIP events on its stream, 3 non-overlapped "
                                 "calls right after the timed region); in_flight_launch_ms = average over the timed region's launches (%d "
                                 "batches in flight, the modexps taking turns at the turnstile)" % V.n_ctx,
                                 in_flight_launch_ms=rsa_ms, launch_ms_basis="single_flight"),
            "int_mac": int_mac_block(counters["pubkey_ops"] * MACS_PER_RSA_VERIFY, ms_step, rsa_ms, iso_rsa, sclk, V.n_ctx),
            "sustained": sustained,
            "corpus_build_s": t_corpus,
        })
        if D.world == 1 and not args.no_end_to_end:
            # the same batch handed over in HOST buffers (what a cgo caller does): H2D copy + pipeline + D2H of the verdicts.
            # Reported beside the headline, never as `value` (inputs resident in HBM).  The library cuts such a batch into pieces
            # and verifies piece k while the pieces behind it cross PCIe (capi.hip collective_verify_pipelined); `unsplit` is the
            # same call with that switched off (copy, then verify).
            def host_call(cx, reps, fresh=False):
                ts = []
                for _ in range(reps):
                    # fresh: buffers the runtime has never seen (a Go caller's slices are new memory every time)
                    bufs = [np.copy(z[k]) for k in ("tb", "to", "sb", "so")] if fresh else [z[k] for k in ("tb", "to", "sb", "so")]
                    t_h = time.perf_counter()
                    e_h, nv_h, _ = cx.collective_verify(V.qhs[0], *bufs)
                    ts.append(time.perf_counter() - t_h)
                    assert (e_h == err).all() and (nv_h == nver).all()
                return ts
            c0 = V.ctxs[0]
            host_call(c0, 2)                                  # worker arenas are allocated at the first pipelined call
            hb = host_call(c0, 7)
            trace = c0.host_pipeline_trace()
            hb_fresh = host_call(c0, 4, fresh=True)
            c0.set_host_pipeline(0, "ring")
            host_call(c0, 1)
            hb_ring = host_call(c0, 5)
            trace_ring = c0.host_pipeline_trace()
            hb_ring_fresh = host_call(c0, 3, fresh=True)
            c0.set_host_pipeline(1)
            hb1 = host_call(c0, 3)
            hb1_fresh = host_call(c0, 3, fresh=True)
            c0.set_host_pipeline(0)
            pcie_bytes = int(z["to"][-1]) + int(z["so"][-1]) + 16 * items + 7 * items
            # three callers at once, each on its own context with its own host slices (the shim's goroutines)
            import threading
            t3 = {}

            n_callers = min(3, V.n_ctx)
            N3 = 16          # calls per caller in the contended legs (4 left the figure to the phase the callers happened to start in)
            gate = threading.Barrier(n_callers)

            def caller(k):
                host_call(V.ctxs[k], 2)          # this context's workers and arenas are allocated at its first pipelined call
                gate.wait()                      # ... every caller's, before anybody's clock starts
                t0 = time.perf_counter()
                host_call(V.ctxs[k], N3)
                t3[k] = (t0, time.perf_counter())
            ths = [threading.Thread(target=caller, args=(k,)) for k in range(n_callers)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            span3 = max(b for _, b in t3.values()) - min(a for a, _ in t3.values())
            # The same batch with the client certificate taken out of the PCIe stream (bftkv_gpu_collective_verify_segments): TBSS
            # ends in chunk(Cert) (packet/packet.go:192-212), the SAME bytes behind every write -- sent once, laid out on the device.
            from bftkv_amd import host as HM
            pb, po, shb, sho, seg = HM.split_tails(z["tb"], z["to"], [cl.client.entity])

            def seg_call(cx, reps):
                ts = []
                for _ in range(reps):
                    t_h = time.perf_counter()
                    e_h, nv_h, _ = cx.collective_verify_segments(V.qhs[0], pb, po, shb, sho, seg, z["sb"], z["so"])
                    ts.append(time.perf_counter() - t_h)
                    assert (e_h == err).all() and (nv_h == nver).all()
                return ts
            seg_call(c0, 2)
            hs = seg_call(c0, 7)
            trace_seg = c0.host_pipeline_trace()
            t3s = {}
            gate2 = threading.Barrier(n_callers)

            def caller_seg(k):
                seg_call(V.ctxs[k], 2)
                gate2.wait()
                t0 = time.perf_counter()
                seg_call(V.ctxs[k], N3)
                t3s[k] = (t0, time.perf_counter())
            ths = [threading.Thread(target=caller_seg, args=(k,)) for k in range(n_callers)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            span3s = max(b for _, b in t3s.values()) - min(a for a, _ in t3s.values())
            seg_bytes = int(po[-1]) + int(sho[-1]) + int(z["so"][-1]) + 20 * items + 7 * items
            segments = {"what": "bftkv_gpu_collective_verify_segments: payload = prefix || the client's certificate, sent once (every answer "
                                "checked against the resident call)",
                        "ms_per_call_alone": min(hs) * 1e3, "ms_per_call_alone_median": float(np.median(hs)) * 1e3,
                        "ms_per_call_three_callers": span3s / (N3 * len(t3s)) * 1e3, "verifies_per_sec_three_callers": ref_ops * N3 * len(t3s) / span3s,
                        "bytes_over_pcie": seg_bytes, "pcie_floor_ms_at_63GBps": seg_bytes / 63e9 * 1e3, "shared_tails": 1,
                        "tail_bytes": int(sho[-1]), "timeline_us": trace_seg}
            out["end_to_end"] = {
                "ms_per_step": min(hb) * 1e3, "ms_per_step_median": float(np.median(hb)) * 1e3,
                "verifies_per_sec": ref_ops / min(hb), "packets_per_sec": n_sigs / min(hb),
                "bytes_over_pcie": pcie_bytes, "pcie_floor_ms_at_63GBps": pcie_bytes / 63e9 * 1e3,
                "over_pcie_floor": min(hb) / (pcie_bytes / 63e9),
                "fresh_buffers_ms_per_step": min(hb_fresh) * 1e3,
                "copy": "hipMemcpyAsync from the caller's pageable memory, piece by piece, on a helper thread (the runtime pins the pages in "
                        "place and remembers them; fresh_buffers = arrays copied anew before every call); ring = the library's page-locked staging "
                        "ring instead (4 helper threads memcpy, DMA follows)",
                "ring_ms_per_step": min(hb_ring) * 1e3, "ring_fresh_buffers_ms_per_step": min(hb_ring_fresh) * 1e3, "ring_timeline_us": trace_ring,
                "unsplit_ms_per_step": min(hb1) * 1e3, "unsplit_fresh_buffers_ms_per_step": min(hb1_fresh) * 1e3,
                "timeline_us": trace,
                "three_callers": {"calls": N3 * len(t3), "ms_per_call": span3 / (N3 * len(t3)) * 1e3, "verifies_per_sec": ref_ops * N3 * len(t3) / span3},
                "segments": segments,
                "note": "bftkv_gpu_collective_verify on pageable host memory in, verdicts out (best of 7; verdicts and exit counts checked "
                        "against the resident call every time); never the headline"}
            out["host_buffers"] = {"ms_per_step": min(hb) * 1e3, "verifies_per_sec": n_sigs / min(hb), "bytes_over_pcie": pcie_bytes,
                                   "note": "see end_to_end (kept under its round-3 name; verifies_per_sec here counts packets)"}
        if D.world == 1 and not args.no_serving:
            out["serving"] = serving_leg(args, cl, z, want_ok)
            if isinstance(out["serving"], dict):
                out["serving"]["request_certificates"] = cert_serving_leg()
        if D.world == 1 and not args.no_cpu_baseline:
            cerr, cnver, ops, best, nt, st1 = cpu_collective(cl, z["tb"], z["to"], z["sb"], z["so"], budget_s=args.cpu_budget)
            out["cpu_baseline"] = {
                "value": ops / best, "unit": "verifies/s", "cores": effective_cores(), "threads": nt, "kind": "port",
                "sample": "all %d writes of the GPU batch (%d public-key ops after the reference's early exit at suff=%d), best thread count "
                          "%d of a sweep up to %d logical CPUs (usable per affinity/cgroup: %d); OpenSSL libcrypto bignum/SHA (faster than "
                          "Go math/big)" % (items, ops, cl.suff, nt, os.cpu_count() or 1, effective_cores()),
                "verdicts_per_sec": items / best, "single_thread_verifies_per_sec": st1,
                "gpu_verdicts_identical_to_cpu": bool((cerr == err).all() and (cnver == nver).all()),
                "reference_op_count_identical_to_cpu": bool(ops == ref_ops),
                "sweep": dict(CPU_SWEEP)}
    V.close()
    return out if D.rank == 0 else None


# ------------------------------------------------------------------------------------------------------------------
# cfg 3: 100k mixed RSA / DSA signed read replies -> maxTimestampedValue per variable
# ------------------------------------------------------------------------------------------------------------------
def dry_cfg3(args, D, lo, hi, n_vars_total):
    """--dry-run of cfg 3's rank split: variables [lo, hi) of this rank, 9..11 replies per variable (by variable index), reply
    verdicts constructed by GLOBAL reply index; ranks hold DIFFERENT reply counts, so the bitmap rows are sized by the largest
    (what Verifier does with D.max_int) and each row is cut back to its rank's count when the job's vector is assembled."""
    per_var = lambda a, b: 9 + (np.arange(a, b, dtype=np.int64) % 3)
    first = int(per_var(0, lo).sum())                # global index of this rank's first reply
    n_replies = int(per_var(lo, hi).sum())
    slots = D.max_int(n_replies)
    ok = constructed_ok(first, first + n_replies, 3)
    gathered, gathers = [], [0]

    def run(k):
        for _ in range(k):
            gathered.append(dry_exchange(D, ok, slots))
            gathers[0] += 1
    elapsed = timed_region(D, run, args.steps, args.warmup)
    counts = D.gather_ints([lo, hi, n_replies])
    rows = np.unpackbits(gathered[-1].reshape(D.world, -1), axis=1, bitorder="little")
    full = np.concatenate([rows[r][:counts[r][2]] for r in range(D.world)])
    tot = D.sum_ints([int(ok.sum())])[0]
    if D.rank != 0:
        return None
    return {"dry_run": True, "config": 3, "n_gpus": D.world, "world_size": D.world, "steps": args.steps, "warmup": args.warmup,
            "scaling": "strong", "variables_total": n_vars_total, "variable_ranges": [c[:2] for c in counts], "replies_per_rank": [c[2] for c in counts],
            "replies_total": int(full.size), "slots": slots, "bitmap_bytes_per_rank": (slots + 7) // 8, "allgathers_in_step_loop": gathers[0],
            "gathered_ok": int(full.sum()), "sum_of_rank_ok": tot, "verdict_sha256": verdict_digest(full), "elapsed_s": elapsed}


def bench_cfg3(args, D):
    from corpus import build as cb
    from bftkv_amd import dist as BD
    if not D.dry:
        from bftkv_amd import Context, host as HM
    n = args.replicas or 64
    n_vars_total = args.items or 10000
    lo, hi = BD.shard_range(n_vars_total, D.rank, D.world)      # variables are independent: shard by variable
    n_vars = hi - lo
    if D.dry:
        return dry_cfg3(args, D, lo, hi, n_vars_total)
    cl = cb.make_cluster(n, dsa_fraction=0.5)
    ctx0 = Context(D.local_rank)
    rsa_signer, dsa_pow = gpu_signers(ctx0, cl)
    t0 = time.time()
    rc = None
    cache = "%s.cfg3.n%d.v%d.r%d.pickle" % (args.corpus_cache, n, n_vars, D.rank) if args.corpus_cache else None
    if cache and os.path.exists(cache):
        import pickle
        with open(cache, "rb") as fh:
            rc = pickle.load(fh)
    if rc is None:
        rc = cb.make_read_corpus(cl, n_vars, seed=cb.MASTER_SEED + D.rank, batch_signer=rsa_signer, dsa_batch_pow=dsa_pow)
        if cache:
            import pickle
            with open(cache, "wb") as fh:
                pickle.dump(rc, fh, protocol=4)
    t_corpus = time.time() - t0
    n_replies = len(rc.reply_var)
    V = Verifier(D, cl, n_replies, rc.tbss_blob, rc.tbss_off, rc.ss_blob, rc.ss_off, n_ctx=max(1, args.inflight), ctx0=ctx0)
    # the read quorum: the n_storage storage nodes under the READ rule (wotqs.go:36-70: threshold = f + 1)
    ns = len(rc.storage_ids)
    f = (ns - 1) // 3
    q_read = HM.Quorum.from_qcs([(f, 3 * f + 1, f + 1, f + (ns - f) // 2 + 1, rc.storage_ids)])
    vals = np.frombuffer(b"".join(rc.write_value), dtype=np.uint8).reshape(len(rc.write_value), -1)
    vlen = vals.shape[1]
    reply_t = rc.write_t[rc.reply_write]
    winners = [None]

    # replies grouped by variable once (they do not change from step to step); each step only brings new error bytes
    order = np.argsort(rc.reply_var, kind="stable")
    roff = np.zeros(n_vars + 1, dtype=np.uint64)
    roff[1:] = np.cumsum(np.bincount(rc.reply_var, minlength=n_vars), dtype=np.uint64)
    peers_s = np.ascontiguousarray(rc.reply_peer[order]).astype(np.uint64, copy=False)
    ts_s = np.ascontiguousarray(reply_t[order]).astype(np.uint64, copy=False)
    vb_s = np.ascontiguousarray(vals[rc.reply_write[order]].reshape(-1))
    vo_s = np.arange(n_replies + 1, dtype=np.uint64) * np.uint64(vlen)
    in_order = bool((order == np.arange(n_replies)).all())

    def tally(err):
        """Client.Read's fold over the replies whose <x,v,t,sig,ss> verified (protocol/client.go:181-205) through the host
        mirror (bftkv_host_max_timestamped_value_masked): replies that fail verification are dropped as failures."""
        e = np.ascontiguousarray(err if in_order else err[order], dtype=np.uint8)
        return HM.max_timestamped_value_masked(q_read, n_vars, peers_s, ts_s, vb_s, vo_s, roff, e)

    def finish(j):
        V.complete(j)
        err = V.outs[j % V.n_ctx][0].cpu().numpy()      # 1 byte per reply back to the host
        winners[0] = tally(err)                         # (on the host, while the next steps' kernels run)

    def run(k):
        for i in range(k):                              # up to n_ctx steps in flight (see cfg 2)
            V.submit(i)
            if i >= V.n_ctx - 1:
                finish(i - (V.n_ctx - 1))
        for j in range(max(0, k - (V.n_ctx - 1)), k):
            finish(j)

    run(V.n_ctx)                 # one untimed step per verifier context: its arena is allocated at its first call
    elapsed = timed_region(D, run, args.steps, args.warmup, V.reset_timing)
    sclk = V.ctxs[0].last_sclk_mhz()
    timed_rsa, timed_dsa, timed_hash, timed_total = list(V.rsa_ms), list(V.dsa_ms), list(V.hash_ms), list(V.total_ms)
    iso = []
    for _ in range(3):           # single-flight: non-overlapped calls on one context (the in-flight spans above overlap each other)
        V.submit(0); V.complete(0)
        iso.append(dict(V.last_tm))
    err, nver, bits = V.results(0)
    gather_ok = V.check_gather(err, bits)
    counters = V.ctxs[0].last_counters()
    n_dsa_ops = int(counters["dsa_ops"])
    st_all, st_item = V.ctxs[0].last_statuses()
    ref_ops = reference_pubkey_ops(st_all, st_item, err, nver, n_replies)
    tot_sigs, tot_replies, tot_vars, tot_ref_ops, tot_done_ops = D.sum_ints([rc.n_sigs, n_replies, n_vars, ref_ops, int(counters["pubkey_ops"])])
    win = winners[0]
    sustained = soak(D, run, args.soak_seconds, elapsed / args.steps * 1e3, tot_ref_ops)
    if D.rank == 0:
        rsa_ms, dsa_ms = float(np.mean(timed_rsa)), float(np.mean(timed_dsa))                 # spans with V.n_ctx steps in flight
        iso_rsa, iso_dsa = float(np.mean([t["rsa"] for t in iso])), float(np.mean([t["dsa"] for t in iso]))
        n_rsa_ops = int(counters["pubkey_ops"]) - n_dsa_ops
        dsa_bits = V.ctxs[0].dsa_window_bits()        # the width the tables were built at decides the MAC count of a verification
        dom = "k_dsa_modexp" if iso_dsa >= iso_rsa else "k_rsa_modexp"
        alg_bytes = int(rc.tbss_off[-1]) + n_rsa_ops * RSA_BYTES + n_dsa_ops * DSA_BYTES + (n_replies + 7) // 8
        out = base_line(args, D, "pgp_mixed_rsa_dsa_signature_verifies_per_sec", "verifies/s", tot_ref_ops * args.steps / elapsed, elapsed, "u32",
                        "%d-replica quorum (half RSA-2048, half DSA-2048/256), %d signed read replies <x,v,t,sig,ss> over %d variables per GPU "
                        "(cfg3 of BASELINE.json: %d distinct stored packets, 2 timestamps per variable, 1%% of the variables with a conflicting "
                        "value), %d signature packets; reply verdicts on the GPU, then maxTimestampedValue per variable over the %d-node read "
                        "quorum" % (n, n_replies, n_vars, rc.writes.n_items, rc.n_sigs, ns),
                        {"replicas": n, "replies_per_gpu": n_replies, "variables_per_gpu": n_vars, "sigs_per_gpu": rc.n_sigs,
                         "batches_in_flight": V.n_ctx, "parallelism": "shard-by-variable x%d, RCCL all-gather of reply-verdict bitmaps" % D.world})
        out.update({
            "value_counts": "RSA-2048 / DSA-2048 public-key verifications on the REFERENCE's operation count (packets "
                            "PGPCollectiveSignature.Verify examines before IsSufficient stops it that reach the public-key operation)",
            "reference_pubkey_ops_per_step_per_gpu": ref_ops,
            "pubkey_ops_performed_per_sec": tot_done_ops * args.steps / elapsed, "packets_per_sec": tot_sigs * args.steps / elapsed,
            "reply_verdicts_per_sec": tot_replies * args.steps / elapsed, "read_verdicts_per_sec": tot_vars * args.steps / elapsed,
            "pubkey_ops_per_step_per_gpu": {"rsa": n_rsa_ops, "dsa": n_dsa_ops},
            "sustained": sustained,
            "reads_answered_fraction": float(np.mean(win >= 0)), "replies_accepted_fraction": float((err == 0).mean()),
            "allgather": {"calls_in_timed_region": args.steps, "bytes_per_rank": (n_replies + 7) // 8, "rows_consistent": gather_ok},
            "kernel_ms": {"k_rsa_modexp": rsa_ms, "k_dsa_mul+k_dsa_modexp": dsa_ms, "hash_stream": float(np.mean(timed_hash)),
                          "step_device_span": float(np.mean(timed_total)),
                          "measured": "HIP events of the %d timed steps, %d in flight: SPANS that overlap the neighbouring steps' kernels, not "
                                      "kernel costs (those are single_flight)" % (len(timed_rsa), V.n_ctx),
                          "single_flight": {k: float(np.mean([t[k] for t in iso])) for k in iso[0]},
                          "single_flight_is": "3 non-overlapped calls on one context right after the timed region",
                          "last_call": V.last_tm},
            "roofline": roofline(3, dom, alg_bytes, max(iso_rsa, iso_dsa), "integer-VALU bound; see int_mac; launch_ms = the dominant kernel's "
                                 "single-flight duration (HIP events, one call on the device)",
                                 in_flight_span_ms=max(rsa_ms, dsa_ms), launch_ms_basis="single_flight"),
            "int_mac": int_mac_block(n_rsa_ops * MACS_PER_RSA_VERIFY + n_dsa_ops * macs_per_dsa_verify(dsa_bits), elapsed / args.steps * 1e3,
                                     rsa_ms + dsa_ms, iso_rsa + iso_dsa, sclk, V.n_ctx),
            "dsa_tables": {"window_bits": dsa_bits, "products_per_verify": macs_per_dsa_verify(dsa_bits) // 11552,
                           "gb_allocated": V.ctxs[0].dsa_table_bytes()[0] / 1e9,       # the arena as the library holds it (bftkv_gpu_dsa_table_bytes)
                           "dsa_keys": sum(1 for r in cl.replicas if r.algo == cb.PK_DSA),
                           "gb_pinned": (sum(1 for r in cl.replicas if r.algo == cb.PK_DSA) * 2 * ((256 + dsa_bits - 1) // dsa_bits) *
                                         ((1 << dsa_bits) - 1) * 304 / 1e9) if dsa_bits else 0.0,
                           "note": "width chosen by the library for this keyring and the free HBM (bftkv_gpu_dsa_window_bits: the widest of 18 / 16 / "
                                   "15 / 14 / 13 / 12 / 10 / 8 bits whose tables fit); bftkv_gpu_set_dsa_table_budget bounds it (32 keys: 8 GB -> 14 "
                                   "bits, 24 GB -> 16); gb_pinned = the tables of the ring's keys, gb_allocated = the arena with its growth margin"},
            "corpus_build_s": t_corpus,
        })
        if D.world == 1 and not args.no_cpu_baseline:
            # the CPU path on the DISTINCT stored packets (a reply is a byte-identical copy of one of them), mapped to the replies
            w = rc.writes
            cerr_w, cnver_w, ops_w, best, nt, st1 = cpu_collective(cl, w.tbss_blob, w.tbss_off, w.ss_blob, w.ss_off, budget_s=args.cpu_budget)
            cerr, cnver = cerr_w[rc.reply_write], cnver_w[rc.reply_write]
            # read tally restated by the oracle (oracle/collective.py max_timestamped_value) over the CPU verdicts
            from oracle import collective as col
            from oracle import wotqs
            qo = wotqs.WotQ([wotqs.QC(nodes=list(rc.storage_ids), f=f, min=3 * f + 1, threshold=f + 1, suff=f + (ns - f) // 2 + 1)])
            okc = cerr == 0
            t_c0 = time.perf_counter()
            want = []
            order = np.argsort(rc.reply_var, kind="stable")
            bounds = np.searchsorted(rc.reply_var[order], np.arange(n_vars + 1))
            for j in range(n_vars):
                rs = [(int(rc.reply_peer[r]), int(reply_t[r]), rc.write_value[rc.reply_write[r]]) for r in order[bounds[j]:bounds[j + 1]] if okc[r]]
                want.append(col.max_timestamped_value(rs, qo))
            t_tally = time.perf_counter() - t_c0
            got = []
            acc_s = err[order] == 0                                   # winners index the accepted replies in variable order
            okg = order[np.nonzero(acc_s)[0]]
            first = np.concatenate([[0], np.cumsum(np.bincount(rc.reply_var[order][acc_s], minlength=n_vars))])
            for j in range(n_vars):
                r = okg[first[j] + win[j]] if win[j] >= 0 else -1
                got.append(None if r < 0 else (rc.write_value[rc.reply_write[r]], int(reply_t[r])))
            reads_same = all((a is None and b is None) or (a is not None and b is not None and a[0] == b[0] and a[1] == b[1])
                             for a, b in zip(got, want))
            ops_all = int(cnver_w.astype(np.int64)[rc.reply_write].sum())    # lower bound of the CPU's ops over all replies
            out["cpu_baseline"] = {
                "value": ops_w / best, "unit": "verifies/s", "cores": effective_cores(), "threads": nt, "kind": "port",
                "sample": "the %d distinct stored packets the %d replies are copies of (%d public-key ops after the early exit, about half "
                          "DSA), best thread count %d (usable per affinity/cgroup: %d); OpenSSL libcrypto; read tally: Python oracle, %.2f s "
                          "for %d variables" % (w.n_items, n_replies, ops_w, nt, effective_cores(), t_tally, n_vars),
                "reply_verdicts_per_sec": w.n_items / best, "single_thread_verifies_per_sec": st1,
                "gpu_verdicts_identical_to_cpu": bool((cerr == err).all() and (cnver == nver).all()),
                "read_answers_identical_to_oracle": bool(reads_same), "min_cpu_pubkey_ops_all_replies": ops_all, "sweep": dict(CPU_SWEEP)}
    V.close()
    return out if D.rank == 0 else None


# ------------------------------------------------------------------------------------------------------------------
# cfg 4: 256 replicas, 1M-write storm sharded over the ranks
# ------------------------------------------------------------------------------------------------------------------
def cfg4_split(args, world):
    """The storm's split (strong scaling): every rank takes total // world writes and verifies them as `calls` calls over a resident
    batch of `chunk` = `tiles` x `distinct` writes.  1,000,000 writes: 1 GPU 8 calls of 125,000; 2 GPUs 4; 4 GPUs 2; 8 GPUs 1 -- the
    all-gather of a call moves chunk / 8 bytes per rank (8 GPUs: 15,625 B each, 125 KB gathered)."""
    total_writes = args.items or 1000000
    share = total_writes // world
    chunk = min(args.chunk, share)
    distinct = min(args.distinct, chunk)
    tiles = chunk // distinct
    chunk = tiles * distinct
    calls = max(1, share // chunk)
    return total_writes, share, chunk, distinct, tiles, calls


def dry_cfg4(args, D):
    """--dry-run of cfg 4's rank split and exchange: rank r owns the writes [r x share, (r+1) x share) of the storm, call c of a step
    covers chunk writes of them; verdicts constructed by GLOBAL write index, gathered per call like bftkv_gpu_allgather_errs_dev
    gathers them (gloo instead of RCCL), assembled in write order."""
    total_writes, share, chunk, distinct, tiles, calls = cfg4_split(args, D.world)
    lo = D.rank * chunk * calls
    gathered, gathers = [], [0]

    def run(k):
        for _ in range(k):
            for c in range(calls):
                gathered.append(dry_exchange(D, constructed_ok(lo + c * chunk, lo + (c + 1) * chunk, 4), chunk))
                gathers[0] += 1
    elapsed = timed_region(D, run, args.steps, args.warmup)
    rows = [np.unpackbits(g.reshape(D.world, -1), axis=1, bitorder="little")[:, :chunk] for g in gathered[-calls:]]
    full = np.concatenate([rows[c][r] for r in range(D.world) for c in range(calls)])
    own = np.concatenate([rows[c][D.rank] for c in range(calls)])
    tot = D.sum_ints([int(constructed_ok(lo, lo + chunk * calls, 4).sum())])[0]
    own_ok = D.sum_ints([int((own == constructed_ok(lo, lo + chunk * calls, 4)).all())])[0]
    if D.rank != 0:
        return None
    return {"dry_run": True, "config": 4, "n_gpus": D.world, "world_size": D.world, "steps": args.steps, "warmup": args.warmup, "scaling": "strong",
            "writes_requested": total_writes, "writes_per_step": chunk * calls * D.world, "share_per_rank": chunk * calls, "writes_per_call": chunk,
            "calls_per_step": calls, "tiles": tiles, "distinct": distinct, "bitmap_bytes_per_rank_per_call": (chunk + 7) // 8,
            "gathered_bytes_per_call": D.world * ((chunk + 7) // 8), "allgathers_in_step_loop": gathers[0],
            "gathered_ok": int(full.sum()), "sum_of_rank_ok": tot, "ranks_whose_own_rows_match": own_ok,
            "verdict_sha256": verdict_digest(full), "elapsed_s": elapsed}


def bench_cfg4(args, D):
    if D.dry:
        return dry_cfg4(args, D)
    from corpus import build as cb
    from bftkv_amd import Context
    torch = D.torch
    n = args.replicas or 256
    total_writes, share, chunk, distinct, tiles, calls = cfg4_split(args, D.world)
    cl = cb.make_cluster(n)
    ctx0 = Context(D.local_rank)
    rsa_signer, _ = gpu_signers(ctx0, cl)
    t0 = time.time()
    rates = {cb.MUT_BAD_MPI: 0.001, cb.MUT_UNKNOWN_ISSUER: 0.0005, cb.MUT_DUP_SIGNER: 0.0005, cb.MUT_ONE_SHORT: 0.001}
    z = load_or_make(args, "cfg4.n%d.d%d" % (n, distinct), D.rank, lambda: write_corpus_arrays(
        cb.make_write_corpus(cl, distinct, seed=cb.MASTER_SEED + D.rank, batch_signer=rsa_signer, mutation_rates=rates)))
    t_corpus = time.time() - t0
    n_sigs_base = int(z["n_sigs"])
    # the resident batch: `tiles` copies of the distinct writes, laid out back to back in HBM (distinct memory per copy)
    dev = D.dev
    d_tb = torch.from_numpy(z["tb"]).to(dev).repeat(tiles)
    d_sb = torch.from_numpy(z["sb"]).to(dev).repeat(tiles)

    def tile_off(off):
        o = torch.from_numpy(off.astype(np.int64)).to(dev)
        step = int(off[-1])
        body = (o[:-1].unsqueeze(0) + torch.arange(tiles, device=dev, dtype=torch.int64).unsqueeze(1) * step).reshape(-1)
        return torch.cat([body, torch.tensor([tiles * step], device=dev, dtype=torch.int64)])
    d_to, d_so = tile_off(z["to"]), tile_off(z["so"])
    # One call in flight: a second one (measured, 1124.3 vs 1123.8 ms per step) gains nothing here -- the 9 ms of walk / parse /
    # compare per call are machine WORK, not latency, and while a 129 ms modexp has blocks pending the dispatcher hands every
    # freed SIMD slot to it: the neighbour's small kernels run when it has drained, exactly as they do with one call in flight.
    n_fl = 1
    V = Verifier(D, cl, chunk, d_tb, d_to, d_sb, d_so, n_ctx=n_fl, ctx0=ctx0, ss_len=tiles * int(z["so"][-1]))

    def run(k):
        V.run(k * calls)                                # the rank's share of the storm: `calls` resident batches per step, n_fl in flight

    V.run(n_fl)                  # one untimed call per verifier context: its arena is allocated at its first call

    elapsed = timed_region(D, run, args.steps, args.warmup, V.reset_timing)
    sclk = V.ctxs[0].last_sclk_mhz()
    err, nver, bits = V.results(0)
    gather_ok = V.check_gather(err, bits)
    counters = V.ctxs[0].last_counters()
    n_sigs_call = n_sigs_base * tiles
    st_all, st_item = V.ctxs[0].last_statuses()
    ref_ops_call = reference_pubkey_ops(st_all, st_item, err, nver, chunk)
    del st_all, st_item
    tot_sigs, tot_writes, tot_ref_ops, tot_done_ops = D.sum_ints([n_sigs_call * calls, chunk * calls, ref_ops_call * calls,
                                                                  int(counters["pubkey_ops"]) * calls])
    if D.rank == 0:
        rsa_ms, hash_ms = float(np.mean(V.rsa_ms)), float(np.mean(V.hash_ms))
        alg_bytes = tiles * int(z["to"][-1]) + ref_ops_call * RSA_BYTES + (chunk + 7) // 8
        want_ok = np.tile(z["expected_valid"] >= cl.suff, tiles)
        out = base_line(args, D, "pgp_rsa2048_signature_verifies_per_sec", "verifies/s", tot_ref_ops * args.steps / elapsed, elapsed, "u32",
                        "%d-replica quorum, write storm of %d signed writes per step over %d GPU(s) (cfg4 of BASELINE.json): each rank verifies "
                        "its %d writes as %d call(s) over a resident batch of %d writes = %d tiles of %d distinctly signed writes "
                        "(171..256 signatures and a %.1f KB payload each; mutations at 0.1%%), %d signature packets per call" %
                        (n, chunk * calls * D.world, D.world, chunk * calls, calls, chunk, tiles, distinct,
                         float(z["to"][-1]) / distinct / 1024, n_sigs_call),
                        {"replicas": n, "writes_per_step": chunk * calls * D.world, "writes_per_gpu_per_step": chunk * calls,
                         "writes_per_call": chunk, "sigs_per_call": n_sigs_call, "distinct_writes": distinct, "tiles": tiles,
                         "calls_in_flight": V.n_ctx, "parallelism": "shard-by-write x%d, RCCL all-gather of verdict bitmaps per call" % D.world}, scaling="strong")
        out.update({
            "data": "synthetic: %d distinctly signed writes tiled %dx in HBM (every copy is parsed, hashed and verified again; nothing is "
                    "cached across items)" % (distinct, tiles),
            "value_counts": "RSA-2048 public-key verifications on the REFERENCE's operation count (see reference_pubkey_ops_per_call)",
            "reference_pubkey_ops_per_call": ref_ops_call,
            "pubkey_ops_performed_per_sec": tot_done_ops * args.steps / elapsed, "packets_per_sec": tot_sigs * args.steps / elapsed,
            "quorum_verdicts_per_sec": tot_writes * args.steps / elapsed,
            "pubkey_ops_per_call": int(counters["pubkey_ops"]),
            "sufficient_fraction": float((err == 0).mean()),
            "verdicts_match_construction": bool(((err == 0) == want_ok).all()),
            "allgather": {"calls_in_timed_region": args.steps * calls, "bytes_per_rank": (chunk + 7) // 8, "rows_consistent": gather_ok},
            "kernel_ms": {"k_rsa_modexp": rsa_ms, "hash_stream": hash_ms, "call_device_span": float(np.mean(V.total_ms)),
                          "measured": "HIP events of the %d timed calls" % len(V.rsa_ms), "last_call": V.last_tm},
            "roofline": roofline(4, "k_rsa_modexp", alg_bytes, rsa_ms, "integer-VALU bound; see int_mac"),
            "int_mac": int_mac_block(counters["pubkey_ops"] * MACS_PER_RSA_VERIFY * calls, elapsed / args.steps * 1e3, rsa_ms * calls, None, sclk, V.n_ctx),
            "corpus_build_s": t_corpus,
        })
        if D.world == 1 and not args.no_cpu_baseline:
            cerr, cnver, ops, best, nt, st1 = cpu_collective(cl, z["tb"], z["to"], z["sb"], z["so"], budget_s=args.cpu_budget)
            out["cpu_baseline"] = {
                "value": ops / best, "unit": "verifies/s", "cores": effective_cores(), "threads": nt, "kind": "port",
                "sample": "the %d distinct writes of the batch (%d public-key ops after the early exit at suff=%d; the reference re-hashes the "
                          "%.1f KB payload for every signature), best thread count %d (usable per affinity/cgroup: %d); OpenSSL libcrypto" %
                          (distinct, ops, cl.suff, float(z["to"][-1]) / distinct / 1024, nt, effective_cores()),
                "verdicts_per_sec": distinct / best, "single_thread_verifies_per_sec": st1,
                "gpu_verdicts_identical_to_cpu": bool((np.tile(cerr, tiles) == err).all() and (np.tile(cnver, tiles) == nver).all()),
                "sweep": dict(CPU_SWEEP)}
    V.close()
    return out if D.rank == 0 else None


# ------------------------------------------------------------------------------------------------------------------
# cfg 5: threshold share-combine, 10k operations per scheme
# ------------------------------------------------------------------------------------------------------------------
def threshold_serving_leg(tc, res, n, threads="1,64,256", seconds=1.0):
    """ONE share-combine operation per CALL from many caller threads through the micro-batcher (bftkv_gpu_batcher_modmul_product /
    _lagrange_combine / _dsa_calculate_r) -- the shape of Client.DistSign (protocol/client.go:509-546: one ThresholdProcess per signature)
    -- measured by the plain-C load generator tools/serving/threshold_load.c in its own process on the first `n` operations of this
    run's corpus; every answer is compared byte for byte with the batched entry points' result for the same operation (`res`, which
    the cpu_baseline leg compares with oracle/c/threshold.c).  Beside the headline, never `value`."""
    import shutil
    import struct
    import tempfile
    if shutil.which("gcc") is None:
        return None
    from bftkv_amd._native import _ints_to_be
    tmp = tempfile.mkdtemp(prefix="bftkv_thserving")
    try:
        exe = os.path.join(tmp, "threshold_load")
        lib_dir = os.path.join(ROOT, "bftkv_amd")
        cc = subprocess.run(["gcc", "-O2", "-std=gnu99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "serving", "threshold_load.c"),
                             "-L", lib_dir, "-lbftkv_gpu", "-lpthread", "-Wl,-rpath," + lib_dir, "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if cc.returncode != 0:
            return {"error": "gcc: " + cc.stderr.decode(errors="replace")[-300:]}
        flat = lambda rows: [v for r in rows for v in r]
        i32 = lambda a: np.ascontiguousarray(a[:n], dtype="<i4").tobytes()
        path = os.path.join(tmp, "ops.bin")
        with open(path, "wb") as fh:
            fh.write(struct.pack("<I", n))
            fh.write(tc.rsa_n.to_bytes(256, "big") + _ints_to_be(flat(tc.rsa_factors[:n]), 256).tobytes() + res["rsa"][:n].tobytes())
            fh.write(tc.sss_mod.to_bytes(256, "big") + i32(tc.sss_xs) + _ints_to_be(flat(tc.sss_ys[:n]), 256).tobytes() + res["sss"][:n].tobytes())
            fh.write(tc.dsa_q.to_bytes(32, "big") + i32(tc.s_xs) + _ints_to_be(flat(tc.s_ys[:n]), 32).tobytes() + res["s"][:n].tobytes())
            fh.write(tc.dsa_p.to_bytes(256, "big") + i32(tc.r_xs) + _ints_to_be(flat(tc.r_ri[:n]), 256).tobytes() +
                     _ints_to_be(flat(tc.r_vi[:n]), 32).tobytes() + res["r"][:n].tobytes() + res["st_r"][:n].tobytes())
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)          # the serving process runs on the runtime's defaults, like the cfg-2 leg
        r = subprocess.run([exe, path, "256", "0", threads, str(seconds)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, env=env)
        if r.returncode != 0:
            return {"error": "threshold_load rc=%d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:])}
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        return {"what": "one bftkv_gpu_batcher_{modmul_product,lagrange_combine,dsa_calculate_r} per call from N caller threads (plain-C load "
                        "generator in its own process, %.1f s per point after 0.3 s of warm-up, runtime-default hardware queues); every answer "
                        "compared byte for byte with the batched entry point's; `distsign_mix` = RSA product : CalculateR : calculateS in turn"
                        % seconds,
                "lanes": d["lanes"], "max_items_per_batch": d["max_items"], "ops_per_scheme": n,
                "runs": [{"scheme": x["scheme"], "caller_threads": x["threads"], "ops_per_s": x["ops_per_s"], "latency_ms": x["latency_ms"],
                          "wrong_answers": x["wrong"], "device_calls": x["device_calls"]} for x in d["runs"]],
                "usable_host_cores": effective_cores(), "tool": "tools/serving/threshold_load.c"}
    except Exception as e:      # noqa: BLE001  (a side measurement must not take the bench line down)
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def bench_cfg5(args, D):
    from corpus import build as cb
    from bftkv_amd import dist as BD
    if not D.dry:
        from bftkv_amd import Context
        from bftkv_amd._native import _ints_to_be, _ptr
    torch = D.torch
    n_total = args.items or 10000
    lo, hi = BD.shard_range(n_total, D.rank, D.world)    # operations are independent: shard by operation, no exchange step
    N = hi - lo
    if D.dry:
        # --dry-run: the rank split alone -- contiguous shards that tile [0, n_total), no exchange step (a combine's result goes back
        # to the one client that asked for it), per-rank corpus seeds
        def run(k):
            pass
        elapsed = timed_region(D, run, args.steps, args.warmup)
        ranges = D.gather_ints([lo, hi, cb.MASTER_SEED + D.rank])
        tot = D.sum_ints([3 * N])[0]
        if D.rank != 0:
            return None
        return {"dry_run": True, "config": 5, "n_gpus": D.world, "world_size": D.world, "steps": args.steps, "warmup": args.warmup, "scaling": "strong",
                "ops_per_scheme": n_total, "operation_ranges": [r[:2] for r in ranges], "corpus_seeds": [r[2] for r in ranges],
                "scheme_ops_per_step": tot, "exchange_steps": 0, "elapsed_s": elapsed}
    gold = os.path.join(ROOT, "tests", "golden")
    kat = json.load(open(os.path.join(gold, "threshold_kat.json")))
    k0 = json.load(open(os.path.join(gold, "keys_dsa2048.json")))["keys"][0]
    as_int = lambda v: int(v, 16) if isinstance(v, str) else int(v)
    t0 = time.time()
    # shares per combine: the reference's own parameters (n = 10 nodes: SSS k = 7, threshold DSA 2t = 8) -- or, with --replicas n, a
    # dealing to n nodes with k = f + 1 = 2t shares (wot.newQC's READ threshold, wotqs.go:36-70), whose Lagrange coefficients leave
    # the kernels' 31-bit fast path (64 nodes: k = 22); the RSA tree stays at 10 partial signatures
    n_nodes = args.replicas or 10
    K_SSS, K_DSA = (7, 8) if n_nodes == 10 else ((n_nodes - 1) // 3 + 1,) * 2
    tc = cb.make_threshold_corpus(N, int(kat["rsa"]["n"], 16), int(kat["sss"]["pb"], 16), as_int(k0["p"]), as_int(k0["q"]),
                                  seed=cb.MASTER_SEED + D.rank, n_shares=n_nodes, sss_k=K_SSS, dsa_2t=K_DSA)
    t_corpus = time.time() - t0
    ctx = Context(D.local_rank)
    # CalculateR is a chain of ~900 dependent products per operation: 10,000 operations are 625 waves on 1,024 SIMDs, so ONE step
    # leaves most issue slots empty.  Steps are independent: they rotate over n_ctx contexts (own streams), whose kernels share
    # the SIMDs -- the same batches-in-flight form as cfg 2, here buying occupancy rather than hiding heads and tails.
    n_ctx = max(1, args.inflight)
    ctxs = [ctx] + [Context(D.local_rank) for _ in range(n_ctx - 1)]
    lib, dev = ctx.lib, D.dev
    for cx in ctxs:         # the corpus deals to n nodes (x = 1..n, as sss.Distribute numbers them); n = 10: 10^7 stays within the Lagrange fast path
        cx._check(lib.bftkv_gpu_set_lagrange_x_bound(cx.h, n_nodes), "set_lagrange_x_bound")
    flat = lambda rows: [v for r in rows for v in r]
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # inputs resident in HBM: big-endian numbers as the reference serialises them (big.Int.Bytes, left-padded)
    d = {"rsa_f": up(_ints_to_be(flat(tc.rsa_factors), 256)), "sss_x": up(tc.sss_xs), "sss_y": up(_ints_to_be(flat(tc.sss_ys), 256)),
         "s_x": up(tc.s_xs), "s_y": up(_ints_to_be(flat(tc.s_ys), 32)), "r_x": up(tc.r_xs), "r_ri": up(_ints_to_be(flat(tc.r_ri), 256)),
         "r_vi": up(_ints_to_be(flat(tc.r_vi), 32))}
    outs = [{"rsa": torch.zeros((N, 256), dtype=torch.uint8, device=dev), "sss": torch.zeros((N, 256), dtype=torch.uint8, device=dev),
             "s": torch.zeros((N, 32), dtype=torch.uint8, device=dev), "r": torch.zeros((N, 32), dtype=torch.uint8, device=dev),
             "st_sss": torch.zeros(N + 8, dtype=torch.uint8, device=dev), "st_s": torch.zeros(N + 8, dtype=torch.uint8, device=dev),
             "st_r": torch.zeros(N + 8, dtype=torch.uint8, device=dev)} for _ in ctxs]
    o = outs[0]
    m_rsa, m_sss = _ints_to_be([tc.rsa_n], 256), _ints_to_be([tc.sss_mod], 256)
    m_q, m_p = _ints_to_be([tc.dsa_q], 32), _ints_to_be([tc.dsa_p], 256)
    P = lambda t: t.data_ptr()
    import ctypes as C
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps + args.warmup + 4)]
    streams = [torch.cuda.ExternalStream(cx.lib.bftkv_gpu_stream(cx.h), device=dev) for cx in ctxs]
    spans = []
    step_no = [0]

    def one_step():
        e = ev[step_no[0] % len(ev)]
        k = step_no[0] % n_ctx
        ctx, h, stream, o = ctxs[k], ctxs[k].h, streams[k], outs[k]
        step_no[0] += 1
        e[0].record(stream)
        # RSA: S = prod of the 10 partial signatures mod N (rsa.go:318-329)
        ctx._check(lib.bftkv_gpu_modmul_product_dev(h, N, 10, P(d["rsa_f"]), 256, None, 1, _ptr(m_rsa), P(o["rsa"])), "modmul_product_dev")
        e[1].record(stream)
        # SSS: calculateSecret over k = 7 shares mod the 2048-bit prime (sss.go:69-107)
        ctx._check(lib.bftkv_gpu_lagrange_combine_dev(h, N, K_SSS, P(d["sss_x"]), P(d["sss_y"]), 256, None, 1, _ptr(m_sss), P(o["sss"]), P(o["st_sss"])), "lagrange_dev")
        e[2].record(stream)
        # threshold DSA: calculateS over 2t = 8 shares mod q (dsa_core.go:389-403) ...
        ctx._check(lib.bftkv_gpu_lagrange_combine_dev(h, N, K_DSA, P(d["s_x"]), P(d["s_y"]), 32, None, 1, _ptr(m_q), P(o["s"]), P(o["st_s"])), "lagrange_dev")
        e[3].record(stream)
        # ... and CalculateR over 2t = 8 partial r's (dsa.go:33-52)
        ctx._check(lib.bftkv_gpu_dsa_calculate_r_dev(h, N, K_DSA, P(d["r_x"]), P(d["r_ri"]), 256, P(d["r_vi"]), 32, None, 1, _ptr(m_p), _ptr(m_q),
                                                     P(o["r"]), P(o["st_r"])), "calculate_r_dev")
        e[4].record(stream)
        return e

    def run(k):
        es = [one_step() for _ in range(k)]
        for cx in ctxs:
            cx.sync()
        for e in es:
            spans.append([e[i].elapsed_time(e[i + 1]) for i in range(4)])

    run(n_ctx)                   # one untimed step per context (scratch and modulus tables are allocated at the first call)
    elapsed = timed_region(D, run, args.steps, args.warmup, lambda: spans.clear())
    res = {k: o[k].cpu().numpy() for k in o}
    for other in outs[1:]:
        for k in o:
            assert torch.equal(other[k], o[k]), "contexts disagree on " + k
    tot_ops = D.sum_ints([3 * N])[0]
    sustained = soak(D, run, args.soak_seconds, elapsed / args.steps * 1e3, tot_ops)
    # ONE batch at a time -- the workload BASELINE configs[4] names, without the steps in flight that the headline needs: each step
    # is waited for before the next is issued
    n_single = 6
    spans_timed = [list(x) for x in spans]
    spans.clear()
    t0 = time.perf_counter()
    for _ in range(n_single):
        run(1)
    single_ms = D.max_float(time.perf_counter() - t0) / n_single * 1e3
    single_r_ms = float(np.mean(np.array(spans), axis=0)[3])
    if D.rank == 0:
        sp = np.mean(np.array(spans_timed), axis=0)
        names = ["rsa_combine_n10", "sss_calculate_secret_k%d_2048" % K_SSS, "dsa_calculate_s_2t%d_q256" % K_DSA, "dsa_calculate_r_2t%d_2048_256" % K_DSA]
        # dominant: CalculateR.  Algorithmic bytes per op: 2t x (|p| + |q|) in, |q| out
        alg_r = N * (K_DSA * (256 + 32) + 32)
        alg_all = N * ((10 + 1) * 256 + (K_SSS + 1) * 256 + K_SSS * 4 + (K_DSA + 1) * 32 + K_DSA * 4 + K_DSA * (256 + 32) + K_DSA * 4 + 32)
        out = base_line(args, D, "threshold_share_combine_ops_per_sec", "ops/s", tot_ops * args.steps / elapsed, elapsed, "u32",
                        "threshold share-combine (cfg5 of BASELINE.json): per step %d operations of each scheme over %d GPU(s) -- RSA calculateSignature "
                        "(product of 10 partial signatures mod the 2048-bit N of rsa/test.pkcs8), SSS calculateSecret (k=7 of n=10, mod the 2048-bit "
                        "prime of sss_test.go), threshold-DSA combine = calculateS (2t=8, 256-bit q) + CalculateR (2t=8, 2048/256-bit group); one "
                        "operation = one scheme-level combine (3 per index); %d independent steps in flight on %d contexts with GPU_MAX_HW_QUEUES=%s "
                        "(set before the HIP runtime loads: a service must export it itself) -- ONE step at a time is `single_flight`%s"
                        % (n_total, D.world, n_ctx, n_ctx, os.environ.get("GPU_MAX_HW_QUEUES", "4 (runtime default)"),
                           "" if n_nodes == 10 else "; THIS RUN (--replicas %d): dealings to %d nodes, SSS k = %d and threshold DSA 2t = %d shares per "
                           "combine instead of 7 / 8 -- Lagrange coefficients on exact big integers" % (n_nodes, n_nodes, K_SSS, K_DSA)),
                        {"ops_per_scheme": n_total, "ops_per_scheme_per_gpu": N, "schemes": 3, "steps_in_flight": n_ctx,
                         "parallelism": "shard-by-operation x%d, no exchange step (results go back to the one client that asked)" % D.world},
                        scaling="strong")
        out.update({
            "per_scheme_ops_per_sec_per_gpu": {names[i]: N / (sp[i] * 1e-3) for i in range(4)},
            "kernel_ms": {names[i]: float(sp[i]) for i in range(4)},
            "roofline": roofline(5, "k_multiexp", alg_r, single_r_ms,
                                 "launch_ms = single-flight span of the CalculateR call (k_lagrange_inv/terms, 2 x k_multiexp, k_u256_inv_modq, "
                                 "k_limbs_mod_q) with ONE step on the device; compare it with single_flight.ms_per_step -- the headline's "
                                 "ms_per_step has %d independent steps in flight, whose chains overlap on the SIMDs (in_flight_span_ms is such an "
                                 "overlapped span, not a cost); 10k operations are 625 waves: latency-, not bandwidth- or MAC-bound" % n_ctx,
                                 in_flight_span_ms=float(sp[3]), launch_ms_basis="single_flight", single_flight_ms_per_step=single_ms),
            "int_mac": int_mac_block(N * macs_per_calculate_r(K_DSA, 64), elapsed / args.steps * 1e3, float(sp[3]), None, None, n_ctx),
            "int_mac_counts": "CalculateR only (the dominant call): %d limb MACs per operation = 2 x k_multiexp (8 bases then the final "
                              "power; 4-bit windows over the 64 windows of a 256-bit q: 715 general products and 512 squarings)" % macs_per_calculate_r(K_DSA, 64),
            "algorithmic_bytes_per_step": alg_all,
            "sustained": sustained,
            "corpus_build_s": t_corpus,
        })
        sf = int_mac(N * macs_per_calculate_r(K_DSA, 64), single_ms)
        out["single_flight"] = {"what": "the same step with ONE batch of %d operations per scheme on the device at a time (no steps in flight): "
                                        "what a lone caller of the batched entry points sees" % N,
                                "steps": n_single, "ms_per_step": single_ms, "value": tot_ops / (single_ms * 1e-3), "unit": "ops/s",
                                "calculate_r_ms": single_r_ms,
                                "int_mac": {"achieved": sf["achieved"], "frac": sf["frac"], "frac_of_theoretical": sf["frac_of_theoretical"]}}
        if D.world == 1 and not args.no_serving and n_nodes == 10:      # (the load generator's file layout is the reference's k = 7 / 2t = 8)
            out["serving"] = threshold_serving_leg(tc, res, min(N, 2048))
        if D.world == 1 and not args.no_cpu_baseline:
            # the reference's combine arithmetic restated in C on OpenSSL bignums (oracle/c/threshold.c), threads over operations:
            # all N operations of every scheme, checked against the GPU's bytes; thread counts swept like cfg 2's baseline
            from oracle.cbind import CThreshold
            ct = CThreshold()
            h = {k: d[k].cpu().numpy() for k in ("rsa_f", "sss_y", "s_y", "r_ri", "r_vi")}
            cores = effective_cores()
            cands = sorted({cores} | {t for t in (16, 32, 64, 128) if t <= (os.cpu_count() or 1)})
            best = None
            cands = [t for t in cands if t >= cores] or [cores]
            timed_nt = []
            for k_, nt in enumerate(cands):
                if k_ >= 2 and best[0] * 2 > args.cpu_budget:       # (never fewer than two thread counts: ADVICE r04)
                    break
                timed_nt.append(nt)
                t0 = time.perf_counter()
                c_rsa = ct.rsa_combine(h["rsa_f"], 10, 256, tc.rsa_n, n_threads=nt)
                t1 = time.perf_counter()
                c_sss, c_st_sss = ct.lagrange_combine(tc.sss_xs, h["sss_y"], 256, tc.sss_mod, n_threads=nt)
                t2 = time.perf_counter()
                c_s, c_st_s = ct.lagrange_combine(tc.s_xs, h["s_y"], 32, tc.dsa_q, n_threads=nt)
                t3 = time.perf_counter()
                c_r, c_st_r = ct.dsa_calculate_r(tc.r_xs, h["r_ri"], 256, h["r_vi"], 32, tc.dsa_p, tc.dsa_q, n_threads=nt)
                t4 = time.perf_counter()
                if best is None or t4 - t0 < best[0]:
                    best = (t4 - t0, nt, (t1 - t0, t2 - t1, t3 - t2, t4 - t3))
            t0 = time.perf_counter()
            S1 = min(N, 500)
            ct.dsa_calculate_r(tc.r_xs[:S1], h["r_ri"][:S1 * K_DSA], 256, h["r_vi"][:S1 * K_DSA], 32, tc.dsa_p, tc.dsa_q, n_threads=1)
            t_r1 = (time.perf_counter() - t0) / S1
            ok_r = c_st_r == 0
            same = bool((c_rsa == res["rsa"]).all() and (c_sss == res["sss"]).all() and (c_s == res["s"]).all() and
                        ((res["st_r"][:N] != 0) == (c_st_r != 0)).all() and (c_r[ok_r] == res["r"][ok_r]).all() and
                        not c_st_sss.any() and not c_st_s.any() and not res["st_sss"][:N].any() and not res["st_s"][:N].any())
            bt = best[2]
            out["cpu_baseline"] = {"value": 3.0 * N / best[0], "unit": "ops/s", "cores": effective_cores(), "threads": best[1], "kind": "port",
                                   "sample": "oracle/c/threshold.c (the reference's Lagrange / combine steps on OpenSSL BN_mod_exp_mont / BN_mod_mul / "
                                             "BN_mod_inverse, faster than Go math/big) over all %d operations of every scheme, best thread count %d of a "
                                             "sweep (usable per affinity/cgroup: %d): %.1f / %.1f / %.1f / %.1f ms for RSA / SSS / calculateS / "
                                             "CalculateR; one thread: %.0f us per CalculateR" %
                                             (N, best[1], cores, bt[0] * 1e3, bt[1] * 1e3, bt[2] * 1e3, bt[3] * 1e3, t_r1 * 1e6),
                                   "per_scheme_ops_per_sec": {names[i]: N / bt[i] for i in range(4)},
                                   "gpu_results_identical_to_cpu": same,
                                   "sweep": {"candidates": cands, "timed": [{"threads": t} for t in timed_nt], "truncated_by_budget": len(timed_nt) < len(cands),
                                             "untuned": bool(len(timed_nt) < len(cands) and best[1] == timed_nt[-1]), "budget_s": args.cpu_budget}}
    for cx in reversed(ctxs):
        cx.close()
    return out if D.rank == 0 else None


# ------------------------------------------------------------------------------------------------------------------
# cfg 1: 4-replica clique, 100 RSA-2048 signed writes on the CPU path (BASELINE.json configs[0]: plumbing, no GPU number)
# ------------------------------------------------------------------------------------------------------------------
def bench_cfg1(args, D):
    """configs[0] is the reference's own CPU-runnable case.  No Go toolchain exists in this image, so what is timed is the C
    restatement of PGPCollectiveSignature.Verify (oracle/c/oracle.c on OpenSSL: per-signature re-hash of the payload, per-signature
    IsSufficient, early exit -- the reference's algorithm shape) over the 100 writes, on one thread (a Go server verifies one write
    per request goroutine) and on the box's cores.  The same 100 writes then go through the verifier as ONE call, only as the
    identity check of the plumbing (verdicts and exit counts); its latency is reported, not a throughput."""
    from corpus import build as cb
    n, items = 4, 100
    cl = cb.make_cluster(n)
    z = write_corpus_arrays(cb.make_write_corpus(cl, items, seed=cb.MASTER_SEED, with_client_sig=True,
                                                 mutation_rates={cb.MUT_BAD_MPI: 0.03, cb.MUT_ONE_SHORT: 0.03, cb.MUT_UNKNOWN_ISSUER: 0.02}))
    co = c_oracle_for(cl)
    reps = []
    for _ in range(5):
        t0 = time.perf_counter()
        cerr, cnver, ops = co.collective_verify(z["tb"], z["to"], z["sb"], z["so"], n_threads=1)
        reps.append(time.perf_counter() - t0)
    t1 = min(reps)
    cores = effective_cores()
    t0 = time.perf_counter()
    co.collective_verify(z["tb"], z["to"], z["sb"], z["so"], n_threads=cores)
    tn = time.perf_counter() - t0
    out = {"metric": "pgp_rsa2048_signature_verifies_per_sec", "value": ops / t1, "unit": "verifies/s", "device": "cpu",
           "workload": "4-replica wotqs clique, %d RSA-2048 signed writes (cfg1 of BASELINE.json), %d signature packets, %d public-key ops after "
                       "the early exit at suff=%d" % (items, int(z["n_sigs"]), ops, cl.suff),
           "ms_per_step": t1 * 1e3, "writes_per_sec": items / t1, "threads": 1,
           "all_cores": {"threads": cores, "verifies_per_sec": ops / tn, "writes_per_sec": items / tn},
           "kind": "port", "what": "oracle/c/oracle.c (C on OpenSSL libcrypto; the Go reference cannot be built here), best of 5 passes"}
    if not D.dry:
        from bftkv_amd import Context
        from tests import helpers as H
        ctx = Context(D.local_rank)
        ctx.keyring_set(abi_keys_of(cl))
        qh = ctx.quorum_create(H.abi_qcs(H.clique_quorum(cl)))
        lat = []
        for _ in range(5):
            t0 = time.perf_counter()
            err, nver, _ = ctx.collective_verify(qh, z["tb"], z["to"], z["sb"], z["so"])
            lat.append(time.perf_counter() - t0)
        out["gpu_identity"] = {"gpu_verdicts_identical_to_cpu": bool((cerr == err).all() and (cnver == nver).all()),
                               "sufficient_fraction": float((err == 0).mean()),
                               "one_call_ms_host_buffers": min(lat) * 1e3}
        ctx.close()
    return out


def summarize(out):
    """The fields of a config's full line that go into the default line's `other_configs`."""
    if out is None:
        return None
    keep = {k: out[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "scaling", "dtype", "data") if k in out}
    keep["workload"] = out["config"]["workload"]
    im = out.get("int_mac") or {}
    keep["int_mac"] = {k: im.get(k) for k in ("achieved", "frac", "frac_of_theoretical", "basis")}
    rf = out.get("roofline") or {}
    keep["roofline"] = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "launch_ms")}
    keep["roofline"].update({k: rf[k] for k in ("launch_ms_basis", "in_flight_span_ms", "in_flight_launch_ms", "single_flight_ms_per_step") if k in rf})
    cb_ = out.get("cpu_baseline")
    if cb_:
        keep["cpu_baseline"] = {k: cb_[k] for k in ("value", "unit", "cores", "threads", "kind") if k in cb_}
        if "sweep" in cb_:
            keep["cpu_baseline"]["threads_timed"] = [t["threads"] for t in cb_["sweep"]["timed"]]
            keep["cpu_baseline"]["sweep_truncated_by_budget"] = cb_["sweep"]["truncated_by_budget"]
            keep["cpu_baseline"]["untuned"] = bool(cb_["sweep"].get("untuned", False))
        keep["identity"] = {k: v for k, v in cb_.items() if "identical" in k}
    for k in ("verdicts_match_construction", "kernel_ms", "per_scheme_ops_per_sec_per_gpu", "packets_per_sec", "quorum_verdicts_per_sec",
              "reply_verdicts_per_sec", "read_verdicts_per_sec", "dsa_tables", "single_flight", "serving"):
        if k in out:
            keep[k] = out[k]
    if "kernel_ms" in keep:
        keep["kernel_ms"] = {k: v for k, v in keep["kernel_ms"].items() if isinstance(v, (int, float))}
    return keep


def other_configs(args, D):
    """cfg 1, 3, 4, 5 at full size behind the cfg-2 headline of a default run: each in a FRESH process running exactly
    `python bench.py --config N` (shorter timed regions, no soak, bounded CPU legs).  A process of its own because that is what the
    per-config numbers of DESIGN.md are (cfg 5 read 9.2 ms per step in the process that had just run cfg 2 -- its streams land on
    the runtime's hardware queues differently -- and 7.6 alone), and because a failure or a hang of one config is then recorded
    in its entry and can never take the headline down."""
    res = {}
    # (timed regions long enough that the fill and drain of the steps in flight do not show: cfg 5 keeps 16 steps in flight)
    plan = [(1, []), (5, ["--steps", "64", "--warmup", "16"]), (3, ["--steps", "20", "--warmup", "3"]), (4, ["--steps", "2", "--warmup", "1"])]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    import tempfile
    full = {}
    for cfg, extra in plan:
        # (cfg 5 keeps its `serving` leg: one share-combine per call through the micro-batcher, ~20 s)
        cmd = [sys.executable, os.path.abspath(__file__), "--config", str(cfg), "--gpus", "1", "--soak-seconds", "0",
               "--cpu-budget", str(min(args.cpu_budget, 10.0))] + ([] if cfg == 5 else ["--no-serving"]) + extra + \
              (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
        t0 = time.time()
        key = "cfg%d" % cfg
        fd, side = tempfile.mkstemp(prefix="bench_full_cfg%d_" % cfg, suffix=".json")       # the child's FULL record (its stdout line is the short one)
        os.close(fd)
        try:
            p = subprocess.run(cmd + ["--full-json", side], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                res[key] = {"error": "rc %d" % p.returncode, "stderr_tail": p.stderr[-1200:]}
            else:
                with open(side) as fh:
                    out = json.load(fh)
                full[key] = out
                res[key] = out if cfg == 1 else summarize(out)
        except subprocess.TimeoutExpired:
            res[key] = {"error": "timeout after 900 s"}
        except Exception as e:          # noqa: BLE001 -- reported in the line
            res[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            try:
                os.unlink(side)
            except OSError:
                pass
        res[key]["wall_s"] = time.time() - t0
        res[key]["command"] = " ".join(["python", "bench.py"] + cmd[2:])
    OTHER_FULL.clear()
    OTHER_FULL.update(full)
    return res


OTHER_FULL = {}          # the complete records of the configs run behind the headline (full record only)


def line_summary(out):
    """The line's figures in one small object (placed last in the line): headline, the legs measured beside it, every other config."""
    def g(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    s = {"metric": out.get("metric"), "value": out.get("value"), "ms_per_step": out.get("ms_per_step"), "n_gpus": out.get("n_gpus"),
         "roofline_frac": g(out, "roofline", "frac"), "int_mac_frac": g(out, "int_mac", "frac"),
         "cpu_baseline": g(out, "cpu_baseline", "value"), "identity": {k: v for k, v in (out.get("cpu_baseline") or {}).items() if "identical" in k}}
    if "end_to_end" in out:
        s["host_buffers_ms_per_call"] = {"alone": g(out, "end_to_end", "ms_per_step"), "three_callers": g(out, "end_to_end", "three_callers", "ms_per_call"),
                                         "pcie_floor": g(out, "end_to_end", "pcie_floor_ms_at_63GBps")}
    runs = g(out, "serving", "runs")
    if runs and "verify_calls_per_s" in runs[0]:          # cfg 2: one Verify per call
        s["serving_verify_calls_per_s"] = {str(r["caller_threads"]): r["verify_calls_per_s"] for r in runs}
        s["serving_p99_ms"] = {str(r["caller_threads"]): r["latency_ms"]["p99"] for r in runs}
        cr = g(out, "serving", "request_certificates", "runs")
        if cr:                                            # Server.sign's Issuer + VerifyWithCertificate, one per call
            s["serving_cert_verify_calls_per_s"] = {str(r["caller_threads"]): r["calls_per_s"] for r in cr}
    elif runs:                                            # cfg 5: one share-combine per call, per scheme
        s["serving_ops_per_s_256_callers"] = {r["scheme"]: r["ops_per_s"] for r in runs if r.get("caller_threads") == 256}
    if out.get("single_flight"):
        s["single_flight_ms_per_step"] = g(out, "single_flight", "ms_per_step")
    for k, v in (out.get("other_configs") or {}).items():
        if isinstance(v, dict):
            e = {x: v.get(x) for x in ("value", "unit", "ms_per_step", "error") if v.get(x) is not None}
            if v.get("single_flight"):
                e["single_flight_ms_per_step"] = v["single_flight"].get("ms_per_step")
            th = [r for r in g(v, "serving", "runs") or [] if r.get("caller_threads") == 256]
            if th:
                e["one_op_per_call_256_callers_ops_per_s"] = {r["scheme"]: r["ops_per_s"] for r in th}
            s[k] = e
    return s


def _sig(v, nd=5):
    """floats to `nd` significant digits (the full record keeps every digit); containers recursively"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        return float("%.*g" % (nd, v)) if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _sig(x, nd) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, nd) for x in v]
    return str(v)


def _cut(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def config_digest(v):
    """One config's few figures for the stdout line: value, ms_per_step, the dominant kernel's roofline entry (single-flight launch_ms),
    int_mac fraction, CPU baseline (with `untuned` where the budget cut its sweep short of a maximum), identity flags."""
    if not isinstance(v, dict):
        return None
    e = {x: v.get(x) for x in ("value", "unit", "ms_per_step", "error") if v.get(x) is not None}
    if "error" in e:
        e["error"] = _cut(e["error"], 160)
    rf = v.get("roofline") or {}
    if rf:
        e["roofline"] = {k: rf.get(k) for k in ("kernel", "frac", "launch_ms", "traffic") if rf.get(k) is not None}
        for k in ("in_flight_span_ms", "in_flight_launch_ms"):
            if rf.get(k) is not None:
                e["roofline"][k] = rf[k]
    im = v.get("int_mac") or {}
    if im.get("frac") is not None:
        e["int_mac_frac"] = im["frac"]
        sf = (im.get("single_flight") or {}).get("frac")
        if sf is not None:
            e["int_mac_frac_single_flight"] = sf
    cb_ = v.get("cpu_baseline") or {}
    if cb_.get("value") is not None:
        e["cpu_baseline"] = {k: cb_[k] for k in ("value", "cores", "threads", "kind") if k in cb_}
        untuned = cb_.get("untuned", (cb_.get("sweep") or {}).get("untuned"))
        if untuned:
            e["cpu_baseline"]["untuned"] = True
    ident = dict(v.get("identity") or {})
    ident.update({k: x for k, x in cb_.items() if "identical" in k})
    ident.update({k: x for k, x in (v.get("gpu_identity") or {}).items() if "identical" in k})
    if ident:
        e["identity"] = ident
    sfl = v.get("single_flight") or {}
    if sfl.get("ms_per_step") is not None:
        e["single_flight_ms_per_step"] = sfl["ms_per_step"]
        if sfl.get("value") is not None:
            e["single_flight_value"] = sfl["value"]
    for k in ("quorum_verdicts_per_sec", "reply_verdicts_per_sec", "read_verdicts_per_sec", "threads"):
        if v.get(k) is not None:
            e[k] = v[k]
    dt = v.get("dsa_tables") or {}
    if dt:
        e["dsa_tables"] = {k: dt[k] for k in ("window_bits", "dsa_keys", "gb_pinned") if k in dt}
    th = [r for r in ((v.get("serving") or {}).get("runs") or []) if isinstance(r, dict) and r.get("caller_threads") == 256 and "scheme" in r]
    if th:
        e["one_op_per_call_256_callers_ops_per_s"] = {_cut(r["scheme"], 28): r["ops_per_s"] for r in th}
    return e


def compact_line(out, full_path=None):
    """The ONE stdout line: at most LINE_MAX bytes.  The contract's keys; `roofline` (dominant kernel, single-flight launch_ms),
    `cpu_baseline`, `int_mac`; the two rates BASELINE.json's metric names (signature verifies/s = `value`, quorum verdicts/s);
    `summary` with one digest per config and the serving / host-buffer headline figures.  Everything else is in the full record
    (`full_record`).  Dry-run lines (tests of the launcher; no number) pass through unchanged."""
    if out.get("dry_run"):
        return out
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in out}
    if "data" in line:
        line["data"] = _cut(line["data"], 160)
    cfg = dict(out.get("config") or {})
    if "workload" in cfg:
        cfg["workload"] = _cut(cfg["workload"], 240)
    if "parallelism" in cfg:
        cfg["parallelism"] = _cut(cfg["parallelism"], 100)
    line["config"] = cfg
    for k in ("packets_per_sec", "quorum_verdicts_per_sec", "reply_verdicts_per_sec", "read_verdicts_per_sec", "verdicts_match_construction",
              "reference_pubkey_ops_per_step_per_gpu", "device", "threads", "writes_per_sec"):
        if k in out:
            line[k] = out[k]
    ag = out.get("allgather") or {}
    if ag:
        line["exchange"] = {"rows_consistent": ag.get("rows_consistent"), "bytes_per_rank": ag.get("bytes_per_rank"),
                            "via": "library RCCL on the verifier's stream" if str(ag.get("via", "")).startswith("bftkv_gpu") else "torch.distributed (fallback)"}
    rf = out.get("roofline")
    if rf:
        line["roofline"] = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "launch_ms_basis",
                                                   "in_flight_launch_ms", "in_flight_span_ms", "single_flight_ms_per_step",
                                                   "algorithmic_bytes_per_launch", "traffic_source") if k in rf}
    cb_ = out.get("cpu_baseline")
    if cb_:
        c = {k: cb_[k] for k in ("value", "unit", "cores", "threads", "kind") if k in cb_}
        c["sample"] = _cut(cb_.get("sample", ""), 200)
        c.update({k: v for k, v in cb_.items() if "identical" in k})
        if "single_thread_verifies_per_sec" in cb_:
            c["single_thread"] = cb_["single_thread_verifies_per_sec"]
        sw = cb_.get("sweep") or {}
        if sw:
            c["threads_timed"] = [t["threads"] for t in sw.get("timed", [])]
            c["untuned"] = bool(sw.get("untuned", False))
        line["cpu_baseline"] = c
    im = out.get("int_mac")
    if im:
        line["int_mac"] = {k: im[k] for k in ("achieved", "peak", "frac", "peak_theoretical", "frac_of_theoretical", "sclk_mhz_in_kernel") if k in im}
        line["int_mac"]["unit"] = "u32 MAC/s (v_mad_u64_u32 lanes)"
        line["int_mac"]["basis"] = "ms_per_step"
        if im.get("single_flight"):
            line["int_mac"]["single_flight_frac"] = im["single_flight"].get("frac")
    s = {}
    if "end_to_end" in out:
        ee = out["end_to_end"]
        s["host_buffers"] = {"ms_per_call_alone": ee.get("ms_per_step"), "ms_per_call_three_callers": (ee.get("three_callers") or {}).get("ms_per_call"),
                             "verifies_per_sec_three_callers": (ee.get("three_callers") or {}).get("verifies_per_sec"),
                             "pcie_floor_ms": ee.get("pcie_floor_ms_at_63GBps"), "bytes_over_pcie": ee.get("bytes_over_pcie")}
        if ee.get("segments"):
            s["host_buffers"]["segments"] = {k: ee["segments"].get(k) for k in ("ms_per_call_alone", "ms_per_call_three_callers", "verifies_per_sec_three_callers",
                                                                                 "bytes_over_pcie", "pcie_floor_ms_at_63GBps")}
    sv = out.get("serving") if isinstance(out.get("serving"), dict) else {}
    runs = sv.get("runs") or []
    if runs and "verify_calls_per_s" in runs[0]:
        s["serving_verify_calls_per_s"] = {str(r["caller_threads"]): r["verify_calls_per_s"] for r in runs}
        s["serving_p99_ms"] = {str(r["caller_threads"]): r["latency_ms"]["p99"] for r in runs}
        cr = (sv.get("request_certificates") or {}).get("runs") if isinstance(sv.get("request_certificates"), dict) else None
        if cr:
            s["serving_cert_verify_calls_per_s"] = {str(r["caller_threads"]): r["calls_per_s"] for r in cr}
    elif runs:
        s["serving_ops_per_s_256_callers"] = {_cut(r["scheme"], 28): r["ops_per_s"] for r in runs if r.get("caller_threads") == 256}
    if (out.get("single_flight") or {}).get("ms_per_step") is not None:
        s["single_flight_ms_per_step"] = out["single_flight"]["ms_per_step"]
    if (out.get("kernel_ms") or {}).get("single_flight"):
        s["kernel_ms_single_flight"] = {k: v for k, v in out["kernel_ms"]["single_flight"].items() if isinstance(v, (int, float))}
    if out.get("sustained"):
        s["sustained"] = {k: out["sustained"].get(k) for k in ("steps", "ms_per_step", "value")}
    if out.get("dsa_tables"):
        s["dsa_tables"] = {k: out["dsa_tables"][k] for k in ("window_bits", "dsa_keys", "gb_pinned") if k in out["dsa_tables"]}
    oc = out.get("other_configs") or {}
    for k, v in oc.items():
        if isinstance(v, dict) and v.get("dry_run"):
            s[k] = v                                    # (dry-run rehearsal of the N > 1 extras: small records, kept whole)
        elif isinstance(v, dict):
            s[k] = config_digest(v)
        elif k == "error":
            s["other_configs_error"] = _cut(v, 160)
    line["summary"] = s
    if full_path:
        line["full_record"] = os.path.relpath(full_path, ROOT) if os.path.abspath(full_path).startswith(ROOT + os.sep) else full_path
    exact = {k: line[k] for k in ("value", "ms_per_step") if k in line}       # the contract's own figures keep every digit
    line = _sig(line)
    line.update(exact)
    # a bound, not a hope: drop the least important digests until the line fits
    for victim in ("kernel_ms_single_flight", "sustained", "serving_p99_ms", "serving_cert_verify_calls_per_s", "dsa_tables"):
        if len(json.dumps(line)) <= LINE_MAX:
            break
        line["summary"].pop(victim, None)
    if len(json.dumps(line)) > LINE_MAX:
        for k in list(line["summary"]):
            if k.startswith("cfg") and isinstance(line["summary"][k], dict):
                line["summary"][k] = {x: line["summary"][k][x] for x in ("value", "unit", "ms_per_step", "error") if x in line["summary"][k]}
    if len(json.dumps(line)) > LINE_MAX:
        line["config"] = {"workload": _cut(line["config"].get("workload", ""), 120)}
        line["summary"] = {"truncated": True}
    return line


def multi_rank_extras(args, D, emit):
    """N > 1, default run: BASELINE's actual multi-GPU configs behind the cfg-2 headline, IN this process group -- cfg 4 (the
    1 M-write storm split over the ranks, one RCCL all-gather of verdict bitmaps per call; strong scaling) and cfg 5 (10 k combines
    per scheme sharded by operation, no exchange; strong scaling), with short timed regions.  (On one GPU `other_configs` runs every
    config in a process of its own; ranks of a torchrun group cannot spawn a second group, so here they run in line: cfg 5 then sees
    the runtime's default 4 hardware queues and keeps 4 steps in flight.)  A failure on any rank is recorded in the entry -- all
    ranks agree on it before the next config starts -- and never takes the headline down; `emit` prints the line early if the
    extras do not come back in time."""
    import threading
    res = {}
    # (cfg 5: the HIP runtime of this process started with its default 4 hardware queues -- 4 steps in flight is their optimum)
    plan = [(4, ["--steps", "2", "--warmup", "1"]), (5, ["--steps", "32", "--warmup", "8", "--inflight", "4"])]
    timer = None
    if not D.dry and D.rank == 0:
        def give_up():
            res.setdefault("error", "the configs behind the headline did not finish within 900 s; line printed by the watchdog")
            emit(res)
            os._exit(1)      # ends the torchrun group: the other ranks are stuck in a collective
        timer = threading.Timer(900.0, give_up)
        timer.daemon = True
        timer.start()
    try:
        for cfg, extra in plan:
            key = "cfg%d" % cfg
            argv = ["--config", str(cfg), "--gpus", str(D.world), "--soak-seconds", "0", "--no-serving", "--no-cpu-baseline"] + extra
            if D.dry:
                argv += ["--dry-run"] + (["--items", str(args.items * 64)] if cfg == 4 and args.items else []) + \
                        (["--distinct", "8", "--chunk", "64"] if cfg == 4 and args.items else [])
            a2 = parse_args(argv)
            t0 = time.time()
            out, err = None, None
            try:
                out = {4: bench_cfg4, 5: bench_cfg5}[cfg](a2, D)
            except Exception as e:          # noqa: BLE001 -- reported in the line
                err = "%s: %s" % (type(e).__name__, str(e)[:300])
            bad = D.sum_ints([1 if err else 0])[0]          # every rank learns whether ANY rank failed
            if D.rank == 0:
                if bad:
                    res[key] = {"error": err or "a rank other than 0 failed", "ranks_failed": bad}
                else:
                    res[key] = out if D.dry else summarize(out)
                res[key]["wall_s"] = time.time() - t0
                res[key]["command"] = "in process, after the headline: bench.py " + " ".join(argv)
            if bad:
                break           # the group's state is unknown after a failure: nothing further is attempted
    finally:
        if timer:
            timer.cancel()
    return res


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch(args))
    # stdout carries exactly ONE line, the JSON.  Libraries write there too (RCCL prints a version banner on stdout when the first
    # communicator is created -- seen on the GPU box), so file descriptor 1 is pointed at stderr for the life of the process and the
    # line goes to the descriptor stdout had.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    D = Dist(args)
    try:
        default_run = args.config is None
        everything = (default_run and D.world == 1 and not args.dry_run and not args.no_other_configs and
                      os.environ.get("BFTKV_BENCH_EXTRAS_IN_PROCESS") != "1")
        if args.config is None:
            args.config = 2
        out = {1: bench_cfg1, 2: bench_cfg2, 3: bench_cfg3, 4: bench_cfg4, 5: bench_cfg5}[args.config](args, D)
        printed = [False]

        def emit(extra=None):
            if out is not None and D.rank == 0 and not printed[0]:
                printed[0] = True
                if extra is not None:
                    out["other_configs"] = extra
                if args.dry_run:
                    os.write(json_fd, (json.dumps(out) + "\n").encode())
                    return
                try:
                    out["summary"] = line_summary(out)
                except Exception as e:                    # noqa: BLE001 -- a digest must never take the record down
                    out["summary"] = {"error": repr(e)[:200]}
                # the FULL record: a side file and stderr; stdout gets the short line (LINE_MAX bytes at most)
                path = args.full_json or os.path.join(ROOT, "bench_full.json" if default_run else "bench_full_cfg%d.json" % args.config)
                full = dict(out)
                if OTHER_FULL:
                    full["other_configs_full"] = dict(OTHER_FULL)
                try:
                    with open(path, "w") as fh:
                        json.dump(full, fh)
                        fh.write("\n")
                except OSError as e:
                    sys.stderr.write("bench.py: full record not written to %s: %s\n" % (path, e))
                    path = None
                # (on ONE stderr line that does not start with a brace: a reader that takes the last line beginning with "{" of whatever
                # it captured must find the stdout line, not this one)
                sys.stderr.write("bench.py: full record%s: %s\n" % (" (%s)" % path if path else "", json.dumps(out)))
                sys.stderr.flush()
                try:
                    line = json.dumps(compact_line(out, path))
                except Exception as e:                    # noqa: BLE001 -- the contract's keys at the very least
                    keep = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                                    "scaling", "vs_baseline", "dtype", "data")}
                    keep["config"] = {"workload": _cut((out.get("config") or {}).get("workload", ""), 200)}
                    keep["compact_line_error"] = repr(e)[:200]
                    line = json.dumps(keep)
                os.write(json_fd, (line + "\n").encode())
        if out is not None and everything:
            out["other_configs"] = other_configs(args, D)
            out["other_configs_note"] = ("BASELINE.json configs[0] (cfg1, CPU restatement) and configs[2..4] (cfg3/4/5 at full size on this "
                                         "GPU, each `python bench.py --config N` in a process of its own with shorter timed regions); the headline `value` is cfg2's")
        # (BFTKV_BENCH_EXTRAS_IN_PROCESS=1: the N > 1 form of the extras on ONE rank -- the rehearsal of exactly what the driver's
        # multi-GPU run executes after the headline, on the 1-GPU box; with BFTKV_FORCE_RCCL=1 its exchanges go through real
        # one-rank communicators)
        rehearse = os.environ.get("BFTKV_BENCH_EXTRAS_IN_PROCESS") == "1" and not args.dry_run
        if rehearse and out is not None:
            out.pop("other_configs", None)
        if default_run and (D.world > 1 or rehearse) and not args.no_other_configs:
            extra = multi_rank_extras(args, D, emit)
            if out is not None:
                out["other_configs"] = extra
                out["other_configs_note"] = ("BASELINE.json configs[3] (cfg4: the write storm split over the %d ranks) and configs[4] (cfg5: combines "
                                             "sharded by operation), run in this process group after the headline; the headline `value` is cfg2's" % D.world)
        emit()
    finally:
        D.close()


if __name__ == "__main__":
    main()
