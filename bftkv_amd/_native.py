"""ctypes binding of libbftkv_gpu.so (include/bftkv_gpu.h).

The product path has NO CPU fallback: if the HIP library is missing or no GPU is present, loading /
``Context()`` raises.  Nothing here imports ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbftkv_gpu.so")


class NativeError(RuntimeError):
    pass


class PubKey(C.Structure):
    _fields_ = [("key_id", C.c_uint64), ("entity_id", C.c_uint64), ("pk_algo", C.c_uint8),
                ("usable_sign", C.c_uint8), ("reserved", C.c_uint8 * 6),
                ("n", C.c_void_p), ("n_len", C.c_uint32), ("e", C.c_void_p), ("e_len", C.c_uint32),
                ("g", C.c_void_p), ("g_len", C.c_uint32), ("y", C.c_void_p), ("y_len", C.c_uint32)]


class QC(C.Structure):
    _fields_ = [("f", C.c_int32), ("min", C.c_int32), ("threshold", C.c_int32), ("suff", C.c_int32),
                ("node_ids", C.c_void_p), ("n_nodes", C.c_uint32)]


EXPORTS = [
    "bftkv_gpu_init", "bftkv_gpu_destroy", "bftkv_gpu_last_error", "bftkv_gpu_error_string",
    "bftkv_gpu_keyring_set", "bftkv_gpu_quorum_create", "bftkv_gpu_quorum_destroy",
    "bftkv_gpu_collective_verify", "bftkv_gpu_collective_verify_dev", "bftkv_gpu_sync",
    "bftkv_gpu_signature_verify", "bftkv_gpu_last_statuses", "bftkv_gpu_last_counters",
    "bftkv_gpu_signers", "bftkv_gpu_quorum_tally", "bftkv_gpu_modexp", "bftkv_gpu_last_timing",
    "bftkv_gpu_stream", "bftkv_gpu_modmul_product", "bftkv_gpu_lagrange_combine", "bftkv_gpu_dsa_calculate_r", "bftkv_gpu_selftest_reduce",
    "bftkv_gpu_comm_unique_id", "bftkv_gpu_comm_init", "bftkv_gpu_allgather_verdicts", "bftkv_gpu_sss_distribute", "bftkv_gpu_modinv",
    "bftkv_gpu_batcher_create", "bftkv_gpu_batcher_destroy", "bftkv_gpu_batcher_collective_verify",
    "bftkv_gpu_batcher_signature_verify", "bftkv_gpu_batcher_stats", "bftkv_gpu_set_dsa_window_bits", "bftkv_gpu_dsa_window_bits", "bftkv_gpu_set_dsa_table_budget", "bftkv_gpu_dsa_table_bytes", "bftkv_gpu_collective_verify_segments", "bftkv_gpu_message_verify", "bftkv_gpu_batcher_message_verify",
    "bftkv_gpu_modexp_ops", "bftkv_gpu_allgather_errs_dev", "bftkv_gpu_set_early_exit", "bftkv_gpu_last_sclk_mhz", "bftkv_gpu_modmul_product_dev", "bftkv_gpu_lagrange_combine_dev",
    "bftkv_gpu_dsa_calculate_r_dev", "bftkv_gpu_sss_distribute_dev", "bftkv_gpu_modinv_dev",
    "bftkv_gpu_ctx_fork", "bftkv_gpu_batcher_create_lanes", "bftkv_gpu_batcher_times", "bftkv_gpu_comm_library", "bftkv_gpu_comm_selftest",
    "bftkv_gpu_collective_verify_small", "bftkv_gpu_signature_verify_small", "bftkv_gpu_set_hash_policy", "bftkv_gpu_signers_fenced",
    "bftkv_gpu_set_host_pipeline", "bftkv_gpu_batcher_cert_verify", "bftkv_gpu_host_pipeline_trace",
    "bftkv_gpu_batcher_cert_entity", "bftkv_gpu_set_lagrange_x_bound", "bftkv_gpu_batcher_modmul_product", "bftkv_gpu_batcher_lagrange_combine", "bftkv_gpu_batcher_dsa_calculate_r", "bftkv_gpu_batcher_modexp",
]

_lib = None


def load_library() -> C.CDLL:
    """dlopen the in-tree HIP library; raises NativeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError("%s not built -- run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64p, u8p = C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p
    lib.bftkv_gpu_init.argtypes = [C.c_int, C.POINTER(vp)]
    lib.bftkv_gpu_destroy.argtypes = [vp]
    lib.bftkv_gpu_destroy.restype = None
    lib.bftkv_gpu_last_error.argtypes = [vp]
    lib.bftkv_gpu_last_error.restype = C.c_char_p
    lib.bftkv_gpu_error_string.argtypes = [C.c_int]
    lib.bftkv_gpu_error_string.restype = C.c_char_p
    lib.bftkv_gpu_keyring_set.argtypes = [vp, C.POINTER(PubKey), u32]
    lib.bftkv_gpu_set_dsa_window_bits.argtypes = [vp, u32]
    lib.bftkv_gpu_dsa_window_bits.argtypes = [vp, C.POINTER(u32)]
    lib.bftkv_gpu_set_dsa_table_budget.argtypes = [vp, C.c_uint64]
    lib.bftkv_gpu_dsa_table_bytes.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(u32)]
    lib.bftkv_gpu_set_hash_policy.argtypes = [vp, C.c_int, C.c_int]
    lib.bftkv_gpu_message_verify.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp, C.c_uint64, vp, vp, vp]
    lib.bftkv_gpu_quorum_create.argtypes = [vp, C.POINTER(QC), u32, C.POINTER(C.c_int)]
    lib.bftkv_gpu_quorum_destroy.argtypes = [vp, C.c_int]
    lib.bftkv_gpu_collective_verify.argtypes = [vp, C.c_int, u32, u8p, u64p, u8p, u64p, u8p, vp, u8p, u8p]
    lib.bftkv_gpu_collective_verify_segments.argtypes = [vp, C.c_int, u32, u8p, u64p, u8p, u64p, u32, vp, u8p, u64p, u8p, vp, u8p, u8p]
    lib.bftkv_gpu_collective_verify_dev.argtypes = [vp, C.c_int, u32, u8p, u64p, u8p, u64p, C.c_uint64, u8p, vp, u8p, u8p]
    lib.bftkv_gpu_collective_verify_small.argtypes = [vp, C.c_int, u32, u8p, u64p, u8p, u64p, u8p, u8p]
    lib.bftkv_gpu_signature_verify_small.argtypes = [vp, u32, u8p, u64p, u8p, u64p, u64p, u8p, u8p]
    lib.bftkv_gpu_sync.argtypes = [vp]
    lib.bftkv_gpu_signature_verify.argtypes = [vp, u32, u8p, u64p, u8p, u64p, u64p, u8p, u8p]
    lib.bftkv_gpu_last_statuses.argtypes = [vp, u8p, vp, u32, C.POINTER(u32)]
    lib.bftkv_gpu_last_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.bftkv_gpu_signers.argtypes = [vp, u32, u8p, u64p, u64p, u64p, C.c_uint64]
    lib.bftkv_gpu_signers_fenced.argtypes = [vp, u32, u8p, u64p, u64p, u64p, C.c_uint64, u8p]
    lib.bftkv_gpu_quorum_tally.argtypes = [vp, C.c_int, u32, u64p, u64p, u8p]
    lib.bftkv_gpu_modexp.argtypes = [vp, u32, u8p, u32, vp, u32, u8p, u8p, u32, u8p]
    lib.bftkv_gpu_last_timing.argtypes = [vp, C.POINTER(C.c_float)]
    lib.bftkv_gpu_modmul_product.argtypes = [vp, u32, u32, u8p, u32, vp, u32, u8p, u8p]
    lib.bftkv_gpu_lagrange_combine.argtypes = [vp, u32, u32, vp, u8p, u32, vp, u32, u8p, u8p, u8p]
    lib.bftkv_gpu_dsa_calculate_r.argtypes = [vp, u32, u32, vp, u8p, u32, u8p, u32, vp, u32, u8p, u8p, u8p, u8p]
    lib.bftkv_gpu_stream.argtypes = [vp]
    lib.bftkv_gpu_batcher_create.argtypes = [vp, u32, u32]
    lib.bftkv_gpu_batcher_create.restype = vp
    lib.bftkv_gpu_batcher_create_lanes.argtypes = [vp, u32, u32, u32]
    lib.bftkv_gpu_batcher_create_lanes.restype = vp
    lib.bftkv_gpu_ctx_fork.argtypes = [vp, C.POINTER(vp)]
    lib.bftkv_gpu_batcher_destroy.argtypes = [vp]
    lib.bftkv_gpu_batcher_destroy.restype = None
    lib.bftkv_gpu_batcher_collective_verify.argtypes = [vp, C.c_int, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, u8p, u8p]
    lib.bftkv_gpu_batcher_signature_verify.argtypes = [vp, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, vp, u8p, u8p]
    lib.bftkv_gpu_batcher_cert_verify.argtypes = [vp, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, u8p, u8p, vp, u8p]
    lib.bftkv_gpu_set_lagrange_x_bound.argtypes = [vp, u32]
    lib.bftkv_gpu_batcher_cert_entity.argtypes = [vp, C.c_char_p, C.c_uint64, u8p, u8p, vp, u8p, vp, vp, vp, u32, vp]
    lib.bftkv_gpu_batcher_message_verify.argtypes = [vp, C.c_char_p, C.c_uint64, vp, vp, vp, vp, C.c_uint64, vp, vp, vp]
    lib.bftkv_gpu_batcher_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.bftkv_gpu_batcher_modmul_product.argtypes = [vp, u32, u8p, u32, u8p, u8p, u8p]
    lib.bftkv_gpu_batcher_lagrange_combine.argtypes = [vp, u32, vp, u8p, u32, u8p, u8p, u8p]
    lib.bftkv_gpu_batcher_dsa_calculate_r.argtypes = [vp, u32, vp, u8p, u32, u8p, u32, u8p, u8p, u8p, u8p]
    lib.bftkv_gpu_batcher_modexp.argtypes = [vp, u8p, u32, u8p, u32, u8p, u8p, u8p]
    lib.bftkv_gpu_batcher_times.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.bftkv_gpu_sss_distribute.argtypes = [vp, u32, u32, u32, u8p, u32, vp, u32, u8p, u8p]
    lib.bftkv_gpu_modinv.argtypes = [vp, u32, u8p, u32, vp, u32, u8p, u8p, u8p]
    lib.bftkv_gpu_comm_unique_id.argtypes = [u8p]
    lib.bftkv_gpu_comm_init.argtypes = [vp, C.c_int, C.c_int, u8p]
    lib.bftkv_gpu_comm_library.argtypes = [C.c_char_p, u32, C.POINTER(C.c_int)]
    lib.bftkv_gpu_comm_selftest.argtypes = [vp, u32, C.POINTER(u32)]
    lib.bftkv_gpu_allgather_verdicts.argtypes = [vp, u8p, C.c_uint64, u8p]
    lib.bftkv_gpu_stream.restype = vp
    lib.bftkv_gpu_set_early_exit.argtypes = [vp, C.c_int]
    lib.bftkv_gpu_set_host_pipeline.argtypes = [vp, u32]
    lib.bftkv_gpu_host_pipeline_trace.argtypes = [vp, C.POINTER(C.c_float), u32, C.POINTER(u32)]
    lib.bftkv_gpu_last_sclk_mhz.argtypes = [vp, C.POINTER(C.c_float)]
    lib.bftkv_gpu_modexp_ops.argtypes = lib.bftkv_gpu_modexp.argtypes
    lib.bftkv_gpu_allgather_errs_dev.argtypes = [vp, u8p, u32, u32, u8p]
    lib.bftkv_gpu_modmul_product_dev.argtypes = lib.bftkv_gpu_modmul_product.argtypes
    lib.bftkv_gpu_lagrange_combine_dev.argtypes = lib.bftkv_gpu_lagrange_combine.argtypes
    lib.bftkv_gpu_dsa_calculate_r_dev.argtypes = lib.bftkv_gpu_dsa_calculate_r.argtypes
    lib.bftkv_gpu_sss_distribute_dev.argtypes = lib.bftkv_gpu_sss_distribute.argtypes
    lib.bftkv_gpu_modinv_dev.argtypes = lib.bftkv_gpu_modinv.argtypes
    for name in EXPORTS:
        if name not in ("bftkv_gpu_destroy", "bftkv_gpu_last_error", "bftkv_gpu_error_string", "bftkv_gpu_stream",
                        "bftkv_gpu_batcher_create", "bftkv_gpu_batcher_create_lanes", "bftkv_gpu_batcher_destroy"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def _ptr(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One verifier context = one GPU + one HIP stream (bftkv_gpu_init)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.bftkv_gpu_init(device, C.byref(h))
        if rc != 0:
            raise NativeError("bftkv_gpu_init(device=%d) failed with %d: no usable MI355X / HIP runtime "
                              "(there is no CPU fallback)" % (device, rc))
        self.h = h
        self._keep = []

    def close(self):
        if self.h:
            self.lib.bftkv_gpu_destroy(self.h)
            self.h = None

    def set_hash_policy(self, hash_id: int, state: int):
        """bftkv_gpu_set_hash_policy: MD5 (1) / RIPEMD-160 (3): 0 unknown (fenced), 1 available, 2 not available."""
        self._check(self.lib.bftkv_gpu_set_hash_policy(self.h, hash_id, state), "set_hash_policy")

    def fork(self) -> "Context":
        """bftkv_gpu_ctx_fork: a context with its own streams and arena over this one's key table and quorums
        (verify calls only; close it before its root)."""
        h = C.c_void_p()
        self._check(self.lib.bftkv_gpu_ctx_fork(self.h, C.byref(h)), "ctx_fork")
        f = Context.__new__(Context)
        f.lib, f.h, f._keep, f.root = self.lib, h, [], self
        return f

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise NativeError("%s failed (%d): %s" % (what, rc, self.lib.bftkv_gpu_last_error(self.h).decode()))

    # ---- keyring
    def keyring_set(self, keys):
        """keys: iterable of dicts {key_id, entity_id, pk_algo, usable_sign, n, e[, g, y]} with
        big-endian ``bytes`` material, in keyring order."""
        keys = list(keys)
        arr = (PubKey * max(1, len(keys)))()
        bufs = []
        for i, k in enumerate(keys):
            arr[i].key_id = k["key_id"]
            arr[i].entity_id = k.get("entity_id", k["key_id"])
            arr[i].pk_algo = k["pk_algo"]
            arr[i].usable_sign = 1 if k.get("usable_sign", True) else 0
            for name in ("n", "e", "g", "y"):
                b = k.get(name) or b""
                cb = C.create_string_buffer(b, len(b)) if b else None
                bufs.append(cb)
                setattr(arr[i], name, C.cast(cb, C.c_void_p) if cb is not None else None)
                setattr(arr[i], name + "_len", len(b))
        self._check(self.lib.bftkv_gpu_keyring_set(self.h, arr, len(keys)), "keyring_set")

    def message_verify(self, messages):
        """Signature half of PGPMessage.Decrypt for a batch of already-decrypted packet sequences (bftkv_gpu_message_verify).
        Returns (status[n], signer_key_id[n], peer_id[n], [plain bytes], [file name bytes])."""
        n = len(messages)
        blob = np.frombuffer(b"".join(messages), dtype=np.uint8) if n and sum(map(len, messages)) else np.zeros(1, dtype=np.uint8)
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(m) for m in messages], dtype=np.uint64)
        st = np.zeros(max(1, n), dtype=np.uint8)
        signer = np.zeros(max(1, n), dtype=np.uint64)
        peer = np.zeros(max(1, n), dtype=np.uint64)
        plain = np.zeros(max(1, int(off[n])), dtype=np.uint8)
        poff = np.zeros(n + 1, dtype=np.uint64)
        fn = np.zeros((max(1, n), 256), dtype=np.uint8)
        fl = np.zeros(max(1, n), dtype=np.uint8)
        self._check(self.lib.bftkv_gpu_message_verify(self.h, n, _ptr(np.ascontiguousarray(blob)), _ptr(off), _ptr(st), _ptr(signer), _ptr(peer),
                                                      _ptr(plain), len(plain), _ptr(poff), _ptr(fn), _ptr(fl)), "message_verify")
        plains = [plain[int(poff[i]):int(poff[i + 1])].tobytes() for i in range(n)]
        names = [fn[i, :int(fl[i])].tobytes() for i in range(n)]
        return st[:n], signer[:n], peer[:n], plains, names

    def set_early_exit(self, on: bool) -> None:
        """collective_verify: stop verifying where the reference stops reading (default) / verify every packet."""
        self._check(self.lib.bftkv_gpu_set_early_exit(self.h, 1 if on else 0), "set_early_exit")

    def set_host_pipeline(self, pieces: int, copy: str = "", tight: bool = False) -> None:
        """collective_verify over host buffers: 0 = split big batches by size (default), 1 = never, 2..8 = that many pieces;
        copy = "direct" (hipMemcpyAsync from the caller's memory, the default) or "ring" (the library's page-locked staging ring)."""
        mode = {"": 0, "ring": 0x100, "direct": 0x200}[copy]
        self._check(self.lib.bftkv_gpu_set_host_pipeline(self.h, pieces | mode | (0x400 if tight else 0)), "set_host_pipeline")

    def host_pipeline_trace(self):
        """Host-side timeline (microseconds) of the last pipelined host-buffer call: see bftkv_gpu_host_pipeline_trace."""
        buf = (C.c_float * 96)()
        n = C.c_uint32(0)
        self._check(self.lib.bftkv_gpu_host_pipeline_trace(self.h, buf, 96, C.byref(n)), "host_pipeline_trace")
        v = [float(buf[i]) for i in range(min(96, n.value))]
        if not v:
            return None
        P = int(v[0])
        names = ("ss_enqueued", "payload_enqueued", "picked_up", "payload_hook", "enqueued", "drained", "gpu_start", "gpu_modexp_start", "gpu_modexp_end", "gpu_end")
        return {"pieces": P, "ring": bool(v[1]), "copiers_joined_us": v[2], "copy_stream_drained_us": v[3], "done_us": v[4], "largest_piece_items": int(v[5]), "second_passes": int(v[6]),
                "per_piece_us": [{nm: round(v[8 + 10 * k + j], 1) for j, nm in enumerate(names)} for k in range(P)]}

    def set_dsa_window_bits(self, bits: int) -> None:
        """Pin the DSA fixed-base table width (4, 8, 16 .. 20 bits; 0 = default policy); applies at the next keyring_set."""
        self._check(self.lib.bftkv_gpu_set_dsa_window_bits(self.h, bits), "set_dsa_window_bits")

    def dsa_window_bits(self) -> int:
        """The width the current key table's DSA tables were built at (0: no DSA key)."""
        b = C.c_uint32(0)
        self._check(self.lib.bftkv_gpu_dsa_window_bits(self.h, C.byref(b)), "dsa_window_bits")
        return int(b.value)

    def set_dsa_table_budget(self, nbytes: int) -> None:
        """Bound on the HBM the DSA tables may hold (0 = the free-memory policy); applies at the next keyring_set."""
        self._check(self.lib.bftkv_gpu_set_dsa_table_budget(self.h, int(nbytes)), "set_dsa_table_budget")

    def dsa_table_bytes(self):
        """(bytes of HBM the DSA tables hold, limbs per table entry: 76, or 112 with a key whose p exceeds 2048 bits)"""
        b, e = C.c_uint64(0), C.c_uint32(0)
        self._check(self.lib.bftkv_gpu_dsa_table_bytes(self.h, C.byref(b), C.byref(e)), "dsa_table_bytes")
        return int(b.value), int(e.value)

    # ---- quorum
    def quorum_create(self, qcs) -> int:
        """qcs: iterable of (f, min, threshold, suff, [node ids])."""
        qcs = list(qcs)
        arr = (QC * max(1, len(qcs)))()
        keep = []
        for i, (f, mn, thr, suff, nodes) in enumerate(qcs):
            ids = np.ascontiguousarray(np.array(list(nodes), dtype=np.uint64))
            keep.append(ids)
            arr[i].f, arr[i].min, arr[i].threshold, arr[i].suff = f, mn, thr, suff
            arr[i].node_ids = ids.ctypes.data if len(ids) else None
            arr[i].n_nodes = len(ids)
        out = C.c_int(-1)
        self._check(self.lib.bftkv_gpu_quorum_create(self.h, arr, len(qcs), C.byref(out)), "quorum_create")
        return out.value

    def quorum_destroy(self, q: int):
        self._check(self.lib.bftkv_gpu_quorum_destroy(self.h, q), "quorum_destroy")

    # ---- verification
    def collective_verify(self, quorum: int, tbs_blob, tbs_off, ss_blob, ss_off):
        n = len(tbs_off) - 1
        tbs_blob, ss_blob = _u8(tbs_blob), _u8(ss_blob)
        tbs_off, ss_off = _u64(tbs_off), _u64(ss_off)
        err = np.zeros(n, dtype=np.uint8)
        nver = np.zeros(n, dtype=np.uint32)
        verdict = np.zeros(n, dtype=np.uint8)
        self.last_fenced = np.zeros(n, dtype=np.uint8)       # fenced_out of this call (see include/bftkv_gpu.h "fenced inputs")
        small = self.collective_verify_small(quorum, tbs_blob, tbs_off, ss_blob, ss_off) if self._cross_check(n) else None
        self._check(self.lib.bftkv_gpu_collective_verify(self.h, quorum, n, _ptr(tbs_blob), _ptr(tbs_off), _ptr(ss_blob),
                                                         _ptr(ss_off), _ptr(err), _ptr(nver), _ptr(verdict), _ptr(self.last_fenced)),
                    "collective_verify")
        self._compare_small("collective_verify", small, err, self.last_fenced)
        return err, nver, verdict

    # The parity tests set check_small: every batched verify call is then ALSO made through the staged small-call route
    # (bftkv_gpu_*_verify_small -- midstates from the host, one stream, 8-lane modexp, every packet verified) first, and the two
    # answers must be identical item by item.
    check_small = False

    def collective_verify_segments(self, quorum: int, prefix_blob, prefix_off, shared_blob, shared_off, seg_of_item, ss_blob, ss_off):
        """bftkv_gpu_collective_verify_segments: payload i = prefix i || shared segment seg_of_item[i] (0xFFFFFFFF: none)."""
        n = len(prefix_off) - 1
        prefix_blob, shared_blob, ss_blob = _u8(prefix_blob), _u8(shared_blob), _u8(ss_blob)
        prefix_off, shared_off, ss_off = _u64(prefix_off), _u64(shared_off), _u64(ss_off)
        seg = np.ascontiguousarray(seg_of_item, dtype=np.uint32)
        err = np.zeros(n, dtype=np.uint8)
        nver = np.zeros(n, dtype=np.uint32)
        verdict = np.zeros(n, dtype=np.uint8)
        self.last_fenced = np.zeros(n, dtype=np.uint8)
        self._check(self.lib.bftkv_gpu_collective_verify_segments(self.h, quorum, n, _ptr(prefix_blob), _ptr(prefix_off), _ptr(shared_blob),
                                                                  _ptr(shared_off), len(shared_off) - 1, _ptr(seg), _ptr(ss_blob), _ptr(ss_off),
                                                                  _ptr(err), _ptr(nver), _ptr(verdict), _ptr(self.last_fenced)),
                    "collective_verify_segments")
        return err, nver, verdict

    def _cross_check(self, n: int) -> bool:
        return bool(self.check_small) and 0 < n <= 4096

    @staticmethod
    def _compare_small(what, small, err, fenced):
        if small is None:
            return
        e2, f2 = small
        if not (np.array_equal(e2, err) and np.array_equal(f2, fenced)):
            bad = [int(i) for i in np.nonzero((e2 != err) | (f2 != fenced))[0][:8]]
            raise NativeError("%s: the small-call route disagrees with the batched one at items %s: err %s vs %s, fenced %s vs %s"
                              % (what, bad, e2[bad].tolist(), err[bad].tolist(), f2[bad].tolist(), fenced[bad].tolist()))

    def collective_verify_small(self, quorum: int, tbs_blob, tbs_off, ss_blob, ss_off):
        """bftkv_gpu_collective_verify_small: (err, fenced)."""
        n = len(tbs_off) - 1
        tbs_blob, ss_blob = _u8(tbs_blob), _u8(ss_blob)
        tbs_off, ss_off = _u64(tbs_off), _u64(ss_off)
        err = np.zeros(n, dtype=np.uint8)
        fenced = np.zeros(n, dtype=np.uint8)
        self._check(self.lib.bftkv_gpu_collective_verify_small(self.h, quorum, n, _ptr(tbs_blob), _ptr(tbs_off), _ptr(ss_blob), _ptr(ss_off),
                                                               _ptr(err), _ptr(fenced)), "collective_verify_small")
        return err, fenced

    def signature_verify_small(self, tbs_blob, tbs_off, sig_blob, sig_off, cert_key_id=None):
        """bftkv_gpu_signature_verify_small: (err, fenced)."""
        n = len(tbs_off) - 1
        tbs_blob, sig_blob = _u8(tbs_blob), _u8(sig_blob)
        tbs_off, sig_off = _u64(tbs_off), _u64(sig_off)
        ck = None if cert_key_id is None else _u64(cert_key_id)
        err = np.zeros(n, dtype=np.uint8)
        fenced = np.zeros(n, dtype=np.uint8)
        self._check(self.lib.bftkv_gpu_signature_verify_small(self.h, n, _ptr(tbs_blob), _ptr(tbs_off), _ptr(sig_blob), _ptr(sig_off), _ptr(ck),
                                                              _ptr(err), _ptr(fenced)), "signature_verify_small")
        return err, fenced

    def collective_verify_dev(self, quorum: int, n_items: int, tbs_ptr: int, tbs_off_ptr: int, ss_ptr: int, ss_off_ptr: int,
                              ss_len: int, err_ptr: int, nver_ptr: int, verdict_ptr: int, fenced_ptr: int = 0):
        self._check(self.lib.bftkv_gpu_collective_verify_dev(self.h, quorum, n_items, tbs_ptr, tbs_off_ptr, ss_ptr, ss_off_ptr,
                                                             ss_len, err_ptr, nver_ptr, verdict_ptr, fenced_ptr or None), "collective_verify_dev")

    def sync(self):
        self._check(self.lib.bftkv_gpu_sync(self.h), "sync")

    def signature_verify(self, tbs_blob, tbs_off, sig_blob, sig_off, cert_key_id=None):
        n = len(tbs_off) - 1
        tbs_blob, sig_blob = _u8(tbs_blob), _u8(sig_blob)
        tbs_off, sig_off = _u64(tbs_off), _u64(sig_off)
        ck = None if cert_key_id is None else _u64(cert_key_id)
        err = np.zeros(n, dtype=np.uint8)
        self.last_fenced = np.zeros(n, dtype=np.uint8)
        small = self.signature_verify_small(tbs_blob, tbs_off, sig_blob, sig_off, cert_key_id) if self._cross_check(n) else None
        self._check(self.lib.bftkv_gpu_signature_verify(self.h, n, _ptr(tbs_blob), _ptr(tbs_off), _ptr(sig_blob), _ptr(sig_off),
                                                        _ptr(ck), _ptr(err), _ptr(self.last_fenced)), "signature_verify")
        self._compare_small("signature_verify", small, err, self.last_fenced)
        return err

    def last_statuses(self):
        n = C.c_uint32(0)
        self._check(self.lib.bftkv_gpu_last_statuses(self.h, None, None, 0, C.byref(n)), "last_statuses")
        st = np.zeros(max(1, n.value), dtype=np.uint8)
        item = np.zeros(max(1, n.value), dtype=np.uint32)
        self._check(self.lib.bftkv_gpu_last_statuses(self.h, _ptr(st), _ptr(item), n.value, C.byref(n)), "last_statuses")
        return st[:n.value], item[:n.value]

    def last_counters(self):
        c = (C.c_uint64 * 4)()
        self._check(self.lib.bftkv_gpu_last_counters(self.h, c), "last_counters")
        return {"packets": c[0], "pubkey_ops": c[1], "items": c[2], "dsa_ops": c[3]}

    def last_sclk_mhz(self) -> float:
        v = C.c_float(0)
        self._check(self.lib.bftkv_gpu_last_sclk_mhz(self.h, C.byref(v)), "last_sclk_mhz")
        return float(v.value)

    def last_timing(self):
        ms = (C.c_float * 8)()
        self._check(self.lib.bftkv_gpu_last_timing(self.h, ms), "last_timing")
        return {"total": ms[0], "parse": ms[1], "hash": ms[2], "rsa": ms[3], "tally": ms[4], "compare": ms[5], "dsa": ms[6]}

    def signers(self, ss_blob, ss_off):
        n = len(ss_off) - 1
        ss_blob, ss_off = _u8(ss_blob), _u64(ss_off)
        cap = max(1, int(len(ss_blob) // 12) + 1)
        ids = np.zeros(cap, dtype=np.uint64)
        off = np.zeros(n + 1, dtype=np.uint64)
        self.last_fenced = np.zeros(n, dtype=np.uint8)
        self._check(self.lib.bftkv_gpu_signers_fenced(self.h, n, _ptr(ss_blob), _ptr(ss_off), _ptr(ids), _ptr(off), cap, _ptr(self.last_fenced)), "signers")
        return ids[:int(off[n])], off

    def quorum_tally(self, quorum: int, ids, list_off):
        n = len(list_off) - 1
        ids, list_off = _u64(ids), _u64(list_off)
        v = np.zeros(n, dtype=np.uint8)
        self._check(self.lib.bftkv_gpu_quorum_tally(self.h, quorum, n, _ptr(ids), _ptr(list_off), _ptr(v)), "quorum_tally")
        return v

    # ---- multi-GPU exchange step (RCCL)
    @staticmethod
    def comm_unique_id() -> np.ndarray:
        uid = np.zeros(128, dtype=np.uint8)
        rc = load_library().bftkv_gpu_comm_unique_id(_ptr(uid))
        if rc:
            raise NativeError("bftkv_gpu_comm_unique_id failed (%d): librccl not loadable" % rc)
        return uid

    def comm_init(self, n_ranks: int, rank: int, uid: np.ndarray):
        self._check(self.lib.bftkv_gpu_comm_init(self.h, n_ranks, rank, _ptr(np.ascontiguousarray(uid, dtype=np.uint8))), "comm_init")

    @staticmethod
    def comm_library():
        """(path of the librccl the library resolved, whether the process already held it)"""
        buf = C.create_string_buffer(1024)
        pre = C.c_int(0)
        rc = load_library().bftkv_gpu_comm_library(buf, 1024, C.byref(pre))
        if rc:
            raise NativeError("bftkv_gpu_comm_library failed (%d): librccl not loadable" % rc)
        return buf.value.decode(), bool(pre.value)

    def comm_selftest(self, nbytes: int = 4096):
        """Collective all-gather self-test on the verifier's stream; raises with the rank and RCCL's error string."""
        bad = C.c_uint32(0)
        self._check(self.lib.bftkv_gpu_comm_selftest(self.h, nbytes, C.byref(bad)), "comm_selftest")

    def allgather_verdicts(self, local_ptr: int, nbytes: int, out_ptr: int):
        self._check(self.lib.bftkv_gpu_allgather_verdicts(self.h, local_ptr, nbytes, out_ptr), "allgather_verdicts")

    def allgather_errs_dev(self, err_ptr: int, n_items: int, slots: int, out_ptr: int):
        """Asynchronous on the context's stream: pack err == 0 into a bitmap of ceil(slots/8) bytes, all-gather rank-major."""
        self._check(self.lib.bftkv_gpu_allgather_errs_dev(self.h, err_ptr, n_items, slots, out_ptr), "allgather_errs_dev")

    # ---- threshold share combine (config 5); numbers are Python ints at this level
    def modmul_product(self, factors, moduli, mod_idx, nbytes: int = 256):
        """factors: [n_ops][k] ints; returns [n_ops] ints = prod mod moduli[mod_idx[op]] (rsa.go:318-329)."""
        n, k = len(factors), len(factors[0])
        f = _ints_to_be([x for row in factors for x in row], nbytes)
        m = _ints_to_be(moduli, nbytes)
        mi = np.ascontiguousarray(mod_idx, dtype=np.uint32)
        out = np.zeros((n, nbytes), dtype=np.uint8)
        self._check(self.lib.bftkv_gpu_modmul_product(self.h, n, k, _ptr(f), nbytes, _ptr(mi), len(moduli), _ptr(m), _ptr(out)), "modmul_product")
        return [int.from_bytes(out[i].tobytes(), "big") for i in range(n)]

    def lagrange_combine(self, xs, ys, moduli, mod_idx, nbytes: int = 256):
        """xs: [n_ops][k] small ints, ys: [n_ops][k] ints -> ([n_ops] ints, status[n_ops]) (sss.go:69-107)."""
        n, k = len(xs), len(xs[0])
        x = np.ascontiguousarray(np.array(xs, dtype=np.int32))
        y = _ints_to_be([v for row in ys for v in row], nbytes)
        m = _ints_to_be(moduli, nbytes)
        mi = np.ascontiguousarray(mod_idx, dtype=np.uint32)
        out = np.zeros((n, nbytes), dtype=np.uint8)
        st = np.zeros(n + 8, dtype=np.uint8)
        self._check(self.lib.bftkv_gpu_lagrange_combine(self.h, n, k, _ptr(x), _ptr(y), nbytes, _ptr(mi), len(moduli), _ptr(m), _ptr(out),
                                                        _ptr(st)), "lagrange_combine")
        return [int.from_bytes(out[i].tobytes(), "big") for i in range(n)], st[:n]

    def dsa_calculate_r(self, xs, ri, vi, groups, group_idx, pbytes: int = 256, qbytes: int = 32):
        """CalculateR (dsa.go:33-52): xs/ri/vi [n_ops][k]; groups: list of (p, q) -> ([n_ops] ints, status)."""
        n, k = len(xs), len(xs[0])
        x = np.ascontiguousarray(np.array(xs, dtype=np.int32))
        r = _ints_to_be([v for row in ri for v in row], pbytes)
        v = _ints_to_be([v for row in vi for v in row], qbytes)
        p = _ints_to_be([g[0] for g in groups], pbytes)
        q = _ints_to_be([g[1] for g in groups], qbytes)
        gi = np.ascontiguousarray(group_idx, dtype=np.uint32)
        out = np.zeros((n, qbytes), dtype=np.uint8)
        st = np.zeros(n + 8, dtype=np.uint8)
        self._check(self.lib.bftkv_gpu_dsa_calculate_r(self.h, n, k, _ptr(x), _ptr(r), pbytes, _ptr(v), qbytes, _ptr(gi), len(groups), _ptr(p),
                                                       _ptr(q), _ptr(out), _ptr(st)), "dsa_calculate_r")
        return [int.from_bytes(out[i].tobytes(), "big") for i in range(n)], st[:n]

    def sss_distribute(self, polys, n_shares: int, moduli, mod_idx, nbytes: int = 256):
        """polys: [n_polys][k] ints (polys[p][0] = secret) -> [n_polys][n_shares] ints (sss.go:23-47)."""
        n, k = len(polys), len(polys[0])
        cf = _ints_to_be([c for row in polys for c in row], nbytes)
        m = _ints_to_be(moduli, nbytes)
        mi = np.ascontiguousarray(mod_idx, dtype=np.uint32)
        out = np.zeros((n * n_shares, nbytes), dtype=np.uint8)
        self._check(self.lib.bftkv_gpu_sss_distribute(self.h, n, n_shares, k, _ptr(cf), nbytes, _ptr(mi), len(moduli), _ptr(m), _ptr(out)),
                    "sss_distribute")
        return [[int.from_bytes(out[p * n_shares + x].tobytes(), "big") for x in range(n_shares)] for p in range(n)]

    def modinv(self, values, moduli, mod_idx, nbytes: int = 256):
        v = _ints_to_be(values, nbytes)
        m = _ints_to_be(moduli, nbytes)
        mi = np.ascontiguousarray(mod_idx, dtype=np.uint32)
        out = np.zeros((len(values), nbytes), dtype=np.uint8)
        st = np.zeros(len(values) + 8, dtype=np.uint8)
        self._check(self.lib.bftkv_gpu_modinv(self.h, len(values), _ptr(v), nbytes, _ptr(mi), len(moduli), _ptr(m), _ptr(out), _ptr(st)), "modinv")
        return [int.from_bytes(out[i].tobytes(), "big") for i in range(len(values))], st[:len(values)]

    def selftest_reduce(self, values, moduli, mod_idx, lanes: int, nbytes: int = 256):
        """values[i] - m when values[i] >= m else values[i] (values < 2m): the kernels' conditional subtraction on its own."""
        v = _ints_to_be(values, nbytes)
        m = _ints_to_be(moduli, nbytes)
        mi = np.ascontiguousarray(mod_idx, dtype=np.uint32)
        out = np.zeros((len(values), nbytes), dtype=np.uint8)
        self._check(self.lib.bftkv_gpu_selftest_reduce(self.h, len(values), _ptr(v), nbytes, _ptr(mi), len(moduli), _ptr(m), lanes, _ptr(out)),
                    "selftest_reduce")
        return [int.from_bytes(out[i].tobytes(), "big") for i in range(len(values))]

    def modexp_ops(self, base: np.ndarray, mod_idx: np.ndarray, mods: np.ndarray, exps: np.ndarray) -> np.ndarray:
        """out[i] = base[i] ^ exps[i] mod mods[mod_idx[i]] (one exponent per operation: CalculatePartialR, dsa.go:27-31)."""
        base, mods, exps = _u8(base), _u8(mods), _u8(exps)
        mod_idx = np.ascontiguousarray(mod_idx, dtype=np.uint32)
        assert exps.shape[0] == base.shape[0]
        out = np.zeros_like(base)
        self._check(self.lib.bftkv_gpu_modexp_ops(self.h, base.shape[0], _ptr(base), base.shape[1], _ptr(mod_idx), mods.shape[0],
                                                  _ptr(mods), _ptr(exps), exps.shape[1], _ptr(out)), "modexp_ops")
        return out

    def modexp(self, base: np.ndarray, mod_idx: np.ndarray, mods: np.ndarray, exps: np.ndarray) -> np.ndarray:
        """base [n, nbytes] u8 BE; mods [m, nbytes]; exps [m, exp_len] -> [n, nbytes]."""
        base, mods, exps = _u8(base), _u8(mods), _u8(exps)
        mod_idx = np.ascontiguousarray(mod_idx, dtype=np.uint32)
        out = np.zeros_like(base)
        self._check(self.lib.bftkv_gpu_modexp(self.h, base.shape[0], _ptr(base), base.shape[1], _ptr(mod_idx), mods.shape[0],
                                              _ptr(mods), _ptr(exps), exps.shape[1], _ptr(out)), "modexp")
        return out


class Batcher:
    """bftkv_gpu_batcher: blocking one-message calls from many threads, aggregated into device batches."""

    def __init__(self, ctx: Context, max_items: int = 256, max_wait_us: int = 200, n_lanes: int = 0):
        self.ctx = ctx
        self.lib = ctx.lib
        self.h = C.c_void_p(self.lib.bftkv_gpu_batcher_create_lanes(ctx.h, max_items, max_wait_us, n_lanes))
        if not self.h:
            raise NativeError("bftkv_gpu_batcher_create_lanes failed")

    def close(self):
        if self.h:
            self.lib.bftkv_gpu_batcher_destroy(self.h)
            self.h = None

    def collective_verify(self, quorum: int, tbs: bytes, ss: bytes, raw: bool = False):
        """raw=True returns (rc, err, fenced) without raising (the fail-closed tests look at err when rc != 0)."""
        err = np.zeros(1, dtype=np.uint8)
        fenced = np.zeros(1, dtype=np.uint8)
        rc = self.lib.bftkv_gpu_batcher_collective_verify(self.h, quorum, tbs, len(tbs), ss, len(ss), _ptr(err), _ptr(fenced))
        if raw:
            return rc, int(err[0]), int(fenced[0])
        if rc:
            raise NativeError("batcher collective_verify failed: %d" % rc)
        return int(err[0])

    def signature_verify(self, tbs: bytes, sig: bytes, cert_key_id: Optional[int] = None, raw: bool = False):
        err = np.zeros(1, dtype=np.uint8)
        ck = None if cert_key_id is None else np.array([cert_key_id], dtype=np.uint64)
        fenced = np.zeros(1, dtype=np.uint8)
        rc = self.lib.bftkv_gpu_batcher_signature_verify(self.h, tbs, len(tbs), sig, len(sig), _ptr(ck), _ptr(err), _ptr(fenced))
        if raw:
            return rc, int(err[0]), int(fenced[0])
        if rc:
            raise NativeError("batcher signature_verify failed: %d" % rc)
        return int(err[0])

    def cert_verify(self, cert: bytes, tbs: bytes, sig: Optional[bytes], raw: bool = False):
        """bftkv_gpu_batcher_cert_verify: Signature.Issuer(sig) + VerifyWithCertificate(tbs, sig, issuer) for a principal outside the
        node keyring; sig=None asks for the issuer alone.  Returns (err, fenced, issuer_key_id, fingerprint)."""
        err = np.zeros(1, dtype=np.uint8)
        fenced = np.zeros(1, dtype=np.uint8)
        iid = np.zeros(1, dtype=np.uint64)
        fp = np.zeros(20, dtype=np.uint8)
        rc = self.lib.bftkv_gpu_batcher_cert_verify(self.h, cert, len(cert), tbs, len(tbs), sig, 0 if sig is None else len(sig), _ptr(err), _ptr(fenced),
                                                    iid.ctypes.data, _ptr(fp))
        if raw:
            return rc, int(err[0]), int(fenced[0])
        if rc:
            raise NativeError("batcher cert_verify failed: %d" % rc)
        return int(err[0]), int(fenced[0]), int(iid[0]), fp.tobytes()

    def cert_entity(self, cert: bytes, cap: int = 256):
        """bftkv_gpu_batcher_cert_entity: Issuer(sig) without ReadEntity on the CPU.  Returns (rc, err, fenced, issuer_key_id,
        fingerprint, entity_off, entity_len, roles) with roles = [(role number, index, chosen)]."""
        err, fenced = np.zeros(1, dtype=np.uint8), np.zeros(1, dtype=np.uint8)
        iid, rng = np.zeros(1, dtype=np.uint64), np.zeros(2, dtype=np.uint64)
        fp = np.zeros(20, dtype=np.uint8)
        while True:
            roles, n = np.zeros(max(1, cap), dtype=np.uint32), np.zeros(1, dtype=np.uint32)
            rc = self.lib.bftkv_gpu_batcher_cert_entity(self.h, cert, len(cert), _ptr(err), _ptr(fenced), iid.ctypes.data, _ptr(fp), rng.ctypes.data,
                                                        rng.ctypes.data + 8, roles.ctypes.data, cap, n.ctypes.data)
            if rc == -3 and int(n[0]) > cap:
                cap = int(n[0])
                continue
            return rc, int(err[0]), int(fenced[0]), int(iid[0]), fp.tobytes(), int(rng[0]), int(rng[1]), [(int(r) & 0xFF, (int(r) >> 8) & 0xFFFF, bool(int(r) >> 24)) for r in roles[:int(n[0])]]

    def message_verify(self, msg: bytes):
        """One transport message through the batcher: (status, signer_key_id, peer_id, plain, file_name)."""
        st = np.zeros(1, dtype=np.uint8)
        ids = np.zeros(2, dtype=np.uint64)
        plain = np.zeros(max(1, len(msg)), dtype=np.uint8)
        plen = np.zeros(1, dtype=np.uint64)
        fn = np.zeros(256, dtype=np.uint8)
        fl = np.zeros(1, dtype=np.uint8)
        rc = self.lib.bftkv_gpu_batcher_message_verify(self.h, msg, len(msg), _ptr(st), ids.ctypes.data, ids.ctypes.data + 8, _ptr(plain),
                                                       len(plain), _ptr(plen), _ptr(fn), _ptr(fl))
        if rc:
            raise NativeError("batcher message_verify failed: %d" % rc)
        return int(st[0]), int(ids[0]), int(ids[1]), plain[:int(plen[0])].tobytes(), fn[:int(fl[0])].tobytes()

    # ---- one threshold share-combine operation per call (numbers are Python ints; returns (rc, status, value))
    def modmul_product(self, factors, mod: int, nbytes: int = 256):
        """prod factors mod `mod` (calculateSignature, rsa.go:318-329) as ONE micro-batched operation."""
        f, m = _ints_to_be(factors, nbytes), _ints_to_be([mod], nbytes)
        out, st = np.full(nbytes, 0xAA, dtype=np.uint8), np.zeros(1, dtype=np.uint8)
        rc = self.lib.bftkv_gpu_batcher_modmul_product(self.h, len(factors), _ptr(f), nbytes, _ptr(m), _ptr(out), _ptr(st))
        return rc, int(st[0]), int.from_bytes(out.tobytes(), "big")

    def lagrange_combine(self, xs, ys, mod: int, nbytes: int = 256):
        """sum Lagrange(x_j) y_j mod `mod` (calculateSecret sss.go:81-92, calculateS dsa_core.go:389-403)."""
        x = np.ascontiguousarray(np.array(xs, dtype=np.int32))
        y, m = _ints_to_be(ys, nbytes), _ints_to_be([mod], nbytes)
        out, st = np.full(nbytes, 0xAA, dtype=np.uint8), np.zeros(1, dtype=np.uint8)
        rc = self.lib.bftkv_gpu_batcher_lagrange_combine(self.h, len(xs), _ptr(x), _ptr(y), nbytes, _ptr(m), _ptr(out), _ptr(st))
        return rc, int(st[0]), int.from_bytes(out.tobytes(), "big")

    def dsa_calculate_r(self, xs, ri, vi, p: int, q: int, pbytes: int = 256, qbytes: int = 32):
        """CalculateR (dsa.go:33-52)."""
        x = np.ascontiguousarray(np.array(xs, dtype=np.int32))
        r, v = _ints_to_be(ri, pbytes), _ints_to_be(vi, qbytes)
        pb, qb = _ints_to_be([p], pbytes), _ints_to_be([q], qbytes)
        out, st = np.full(qbytes, 0xAA, dtype=np.uint8), np.zeros(1, dtype=np.uint8)
        rc = self.lib.bftkv_gpu_batcher_dsa_calculate_r(self.h, len(xs), _ptr(x), _ptr(r), pbytes, _ptr(v), qbytes, _ptr(pb), _ptr(qb), _ptr(out), _ptr(st))
        return rc, int(st[0]), int.from_bytes(out.tobytes(), "big")

    def modexp(self, base: int, exp: int, mod: int, nbytes: int = 256, exp_len: int = 32):
        """base^exp mod `mod` (CalculatePartialR dsa.go:27-31; the per-fragment power of rsa.go:161-171)."""
        b, e, m = _ints_to_be([base], nbytes), _ints_to_be([exp], exp_len), _ints_to_be([mod], nbytes)
        out, st = np.full(nbytes, 0xAA, dtype=np.uint8), np.zeros(1, dtype=np.uint8)
        rc = self.lib.bftkv_gpu_batcher_modexp(self.h, _ptr(b), nbytes, _ptr(e), exp_len, _ptr(m), _ptr(out), _ptr(st))
        return rc, int(st[0]), int.from_bytes(out.tobytes(), "big")

    def stats(self):
        st = (C.c_uint64 * 4)()
        self.lib.bftkv_gpu_batcher_stats(self.h, st)
        ns = (C.c_uint64 * 8)()
        self.lib.bftkv_gpu_batcher_times(self.h, ns)
        return {"calls": st[0], "batches": st[1], "max_batch": st[2], "lanes": st[3], "cert_fast": ns[7]}


def _ints_to_be(vals, nbytes: int) -> np.ndarray:
    return np.frombuffer(b"".join(int(v).to_bytes(nbytes, "big") for v in vals), dtype=np.uint8).reshape(len(vals), nbytes).copy()


def _u8(a) -> np.ndarray:
    if isinstance(a, (bytes, bytearray)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.size == 0:
        a = np.zeros(1, dtype=np.uint8)[:0].copy()
    return a


def _u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)
