"""The modular inverse k_modinv runs (bftkv_amd/csrc/safegcd.inc), compiled for the CPU and checked against Python's own inverse: the
arithmetic is the same text on both sides, so what the GPU suite adds is the kernel around it (test_gpu_threshold.py)."""
import ctypes
import math
import random
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
N28 = 76


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = tmp_path_factory.mktemp("safegcd") / "safegcd_host.so"
    subprocess.run(["gcc", "-O2", "-std=gnu99", "-shared", "-fPIC", "-Wall", "-Werror", str(ROOT / "tests/c/safegcd_host.c"), "-o", str(so)], check=True)
    L = ctypes.CDLL(str(so))
    L.sg_host_modinv.restype = ctypes.c_int
    L.sg_host_rounds.restype = ctypes.c_int
    return L


def limbs(v):
    return np.array([(v >> (28 * j)) & 0xFFFFFFF for j in range(N28)], dtype=np.uint32)


def value(a):
    return sum(int(x) << (28 * j) for j, x in enumerate(a))


def inverse(lib, x, m):
    out = np.zeros(N28, dtype=np.uint32)
    xs, ms = limbs(x), limbs(m)
    ok = lib.sg_host_modinv(xs.ctypes.data_as(ctypes.c_void_p), ms.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return ok, value(out)


def check(lib, x, m):
    ok, got = inverse(lib, x, m)
    if math.gcd(x, m) != 1:
        assert (ok, got) == (0, 0), (x, m)
    else:
        assert ok == 1 and got == pow(x, -1, m), (x, m)      # pow(x, -1, 1) is 0, and so is big.Int.ModInverse's answer


def test_every_pair_under_small_moduli(lib):
    for m in range(1, 200, 2):
        for x in range(0, 2 * m + 3):
            check(lib, x, m)


def test_random_operands_of_every_width(lib):
    rng = random.Random(5)
    for bits in (2, 9, 29, 30, 31, 59, 60, 61, 255, 256, 257, 1023, 1024, 2047, 2048):
        for _ in range(60):
            m = rng.getrandbits(bits) | 1 | (1 << (bits - 1))
            for x in (rng.randrange(m), rng.getrandbits(2048), 0, 1, m - 1, m, m + 1):   # unreduced arguments as well
                check(lib, x, m)


def test_operands_that_share_a_factor_and_extreme_values(lib):
    rng = random.Random(6)
    top = (1 << 2048) - 1
    for _ in range(40):
        p = rng.getrandbits(rng.randrange(2, 600)) | 1
        m = p * (rng.getrandbits(rng.randrange(1, 2048 - p.bit_length())) | 1)
        if p > 1:
            check(lib, p * rng.getrandbits(500), m)
        check(lib, rng.getrandbits(2048), m)
    for m in (top, top - 2, (1 << 2047) + 1, 3, 1):
        for x in (top, top - 1, 1 << 2047, 2, (1 << 2048) - 3, 0):
            check(lib, x, m)
    # powers of two against a modulus of all ones: long runs of halvings, sign flips of f
    for k in range(0, 2048, 97):
        check(lib, 1 << k, top)
        check(lib, top - (1 << k), (1 << 2047) - 1)


def test_round_count_stays_under_the_bound_the_kernel_allows(lib):
    rng = random.Random(7)
    worst = 0
    for _ in range(300):
        m = rng.getrandbits(2048) | 1 | (1 << 2047)
        x = rng.getrandbits(2048)
        r = lib.sg_host_rounds(limbs(x).ctypes.data_as(ctypes.c_void_p), limbs(m).ctypes.data_as(ctypes.c_void_p))
        assert 0 <= r
        worst = max(worst, r)
    for x, m in (((1 << 2048) - 1, (1 << 2047) + 1), (1 << 2047, (1 << 2048) - 1), (3, (1 << 2048) - 1), ((1 << 2048) - 2, (1 << 2048) - 1)):
        r = lib.sg_host_rounds(limbs(x).ctypes.data_as(ctypes.c_void_p), limbs(m).ctypes.data_as(ctypes.c_void_p))
        worst = max(worst, r)
    assert worst <= 197, worst            # (49*2048+57)/17 = 5906 division steps; SG_MAX_ROUNDS = 208
