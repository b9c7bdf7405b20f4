"""CPU oracle for the bftkv quorum-verification hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` may be imported, linked or executed by the
product path (``bftkv_amd/``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

PARITY UNPINNED: the reference (yahoo/bftkv) ships no golden vectors for this path, its arithmetic
lives in the un-vendored module golang.org/x/crypto v0.0.0-20191227163750-53104e6ec876 (go.mod:8)
plus the Go 1.12/1.13 standard library, and no Go toolchain exists in this environment.  The
restatement here follows the reference's own call sites (cited per function) and the published
behaviour of x/crypto's ``openpgp`` package; it is pinned instead against GnuPG 2.2.27 (an independent
OpenPGP implementation) and OpenSSL via ``tests/golden`` fixtures, and against the config-5
known-answer relations the reference's tests do hold (crypto/threshold/rsa/rsa_test.go:165-206).
"""
