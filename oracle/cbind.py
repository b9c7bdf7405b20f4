"""ctypes binding of oracle/liboracle.so (oracle/c/oracle.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, "c", f) for f in ("oracle.c", "threshold.c", "Makefile")]
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "c")])
    return LIB


def _nb(x: int) -> bytes:
    return x.to_bytes(max(1, (x.bit_length() + 7) // 8), "big")


class COracle:
    def __init__(self):
        if not os.path.exists(LIB):
            build()
        self.lib = C.CDLL(LIB)
        self.lib.oracle_new.restype = C.c_void_p
        self.lib.oracle_free.argtypes = [C.c_void_p]
        self.lib.oracle_add_key.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int] + [C.c_char_p, C.c_int] * 4
        self.lib.oracle_set_quorum.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
        self.lib.oracle_collective_verify.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 6 + [C.c_int]
        self.lib.oracle_collective_verify.restype = C.c_uint64
        self.lib.oracle_trace_item.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_void_p, C.c_int,
                                               C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        self.lib.oracle_set_weak_hashes.argtypes = [C.c_int]
        self.h = C.c_void_p(self.lib.oracle_new())

    def __del__(self):
        try:
            self.lib.oracle_free(self.h)
        except Exception:
            pass

    def set_weak_hashes(self, md5: bool, ripemd160: bool):
        """process-wide: are MD5 / RIPEMD-160 "available" in the modelled reference binary (oracle/openpgp.py HASH_POLICY)"""
        self.lib.oracle_set_weak_hashes((1 if md5 else 0) | (2 if ripemd160 else 0))

    def set_keyring(self, keyring):
        """keyring: oracle.collective.Keyring"""
        for e in keyring.get_keyring():
            cands = [(e.primary, e.flags_valid, e.flag_sign, e.self_sig_revoked)] + [(k, fv, fs, rr) for k, fv, fs, rr in e.subkeys]
            for k, fv, fs, rr in cands:
                usable = int(not (e.revoked or rr) and not (fv and not fs))
                if k.pk_algo == 17:
                    a, b, g, y = _nb(k.p), _nb(k.q), _nb(k.g), _nb(k.y)
                else:
                    a, b, g, y = _nb(k.n), _nb(k.e), b"", b""
                self.lib.oracle_add_key(self.h, k.key_id, e.id, k.pk_algo, usable, a, len(a), b, len(b), g, len(g), y, len(y))

    def set_quorum(self, q):
        """q: oracle.wotqs.WotQ"""
        n = len(q.qcs)
        arr = lambda xs: np.ascontiguousarray(np.array(xs, dtype=np.int32))
        f, mn, thr, su = arr([c.f for c in q.qcs]), arr([c.min for c in q.qcs]), arr([c.threshold for c in q.qcs]), arr([c.suff for c in q.qcs])
        ids = np.ascontiguousarray(np.array([i for c in q.qcs for i in c.nodes], dtype=np.uint64))
        cnt = arr([len(c.nodes) for c in q.qcs])
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self.lib.oracle_set_quorum(self.h, n, p(f), p(mn), p(thr), p(su), p(ids), p(cnt))

    def collective_verify(self, tbs_blob, tbs_off, ss_blob, ss_off, n_threads: int = 1):
        n = len(tbs_off) - 1
        tbs_blob = np.ascontiguousarray(tbs_blob, dtype=np.uint8)
        ss_blob = np.ascontiguousarray(ss_blob, dtype=np.uint8)
        tbs_off = np.ascontiguousarray(tbs_off, dtype=np.uint64)
        ss_off = np.ascontiguousarray(ss_off, dtype=np.uint64)
        err = np.zeros(n, dtype=np.uint8)
        nver = np.zeros(n, dtype=np.uint32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        ops = self.lib.oracle_collective_verify(self.h, n, p(tbs_blob), p(tbs_off), p(ss_blob), p(ss_off), p(err), p(nver), n_threads)
        return err, nver, int(ops)

    def trace_item(self, tbs: bytes, ss: bytes):
        tr = np.zeros(8192, dtype=np.uint8)
        nver = C.c_uint32(0)
        err = C.c_int(0)
        nt = self.lib.oracle_trace_item(self.h, tbs, len(tbs), ss, len(ss), tr.ctypes.data_as(C.c_void_p), len(tr),
                                        C.byref(nver), C.byref(err))
        return list(tr[:nt]), nver.value, err.value


class CThreshold:
    """oracle/c/threshold.c: the reference's share-combine arithmetic on OpenSSL bignums, threads over operations."""

    def __init__(self):
        build()
        self.lib = C.CDLL(LIB)
        V = C.c_void_p
        self.lib.oracle_rsa_combine.argtypes = [C.c_uint32, C.c_uint32, V, C.c_uint32, V, V, C.c_int]
        self.lib.oracle_lagrange_combine.argtypes = [C.c_uint32, C.c_uint32, V, V, C.c_uint32, V, V, V, C.c_int]
        self.lib.oracle_dsa_calculate_r.argtypes = [C.c_uint32, C.c_uint32, V, V, C.c_uint32, V, C.c_uint32, V, V, V, V, C.c_int]

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p)

    @staticmethod
    def _be(vals, nbytes):
        return np.frombuffer(b"".join(int(v).to_bytes(nbytes, "big") for v in vals), dtype=np.uint8).copy()

    def rsa_combine(self, factors_be, k, nbytes, mod, n_threads=1):
        """factors_be: uint8 [n][k][nbytes] -> uint8 [n][nbytes]"""
        f = np.ascontiguousarray(factors_be, dtype=np.uint8)
        n = f.size // (k * nbytes)
        out = np.zeros((n, nbytes), dtype=np.uint8)
        m = self._be([mod], nbytes)
        self.lib.oracle_rsa_combine(n, k, self._p(f), nbytes, self._p(m), self._p(out), n_threads)
        return out

    def lagrange_combine(self, xs, ys_be, nbytes, mod, n_threads=1):
        xs = np.ascontiguousarray(xs, dtype=np.int32)
        n, k = xs.shape
        y = np.ascontiguousarray(ys_be, dtype=np.uint8)
        out = np.zeros((n, nbytes), dtype=np.uint8)
        st = np.zeros(n, dtype=np.uint8)
        m = self._be([mod], nbytes)
        self.lib.oracle_lagrange_combine(n, k, self._p(xs), self._p(y), nbytes, self._p(m), self._p(out), self._p(st), n_threads)
        return out, st

    def dsa_calculate_r(self, xs, ri_be, pbytes, vi_be, qbytes, p, q, n_threads=1):
        xs = np.ascontiguousarray(xs, dtype=np.int32)
        n, k = xs.shape
        ri = np.ascontiguousarray(ri_be, dtype=np.uint8)
        vi = np.ascontiguousarray(vi_be, dtype=np.uint8)
        out = np.zeros((n, qbytes), dtype=np.uint8)
        st = np.zeros(n, dtype=np.uint8)
        pb, qb = self._be([p], pbytes), self._be([q], qbytes)
        self.lib.oracle_dsa_calculate_r(n, k, self._p(xs), self._p(ri), pbytes, self._p(vi), qbytes, self._p(pb), self._p(qb),
                                        self._p(out), self._p(st), n_threads)
        return out, st
