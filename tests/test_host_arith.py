"""CPU: the mod-q arithmetic of the DSA kernels (bftkv_amd/csrc/u256.h, __host__ __device__) compiled for the host and
checked against textbook references -- the exact inverse / Montgomery code the GPU runs (tools/hostcheck/check_u256.hip)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_u256_modinv_and_montmul_on_host(tmp_path):
    exe = str(tmp_path / "check_u256")
    subprocess.run(["hipcc", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "hostcheck", "check_u256.hip"), "-o", exe],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert b" 0 mismatches; montmul mismatches 0" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not on PATH")
def test_mont_setup_equals_its_definition_word_for_word(tmp_path):
    """hostbn::mont_setup (csrc/host_bignum.h: what every key-table row and every threshold modulus goes through; Montgomery
    exponentiation of 2 on 64-bit words) gives the numbers of mont_setup_by_doubling -- R^2 mod n, n as limbs, -n^-1 mod 2^28 -- for
    random and edge-case moduli of every width and limb count the callers use, and beyond the contract (tools/hostcheck)."""
    exe = str(tmp_path / "check_mont_setup")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "hostcheck", "check_mont_setup.cpp"), "-o", exe],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    r = subprocess.run([exe, "1500"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b"6000 cases, 0 mismatches" in r.stdout, r.stdout.decode() + r.stderr.decode()
