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
