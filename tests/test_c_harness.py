"""CPU: a plain C99 program linked against libbftkv_gpu.so through include/*.h only (the position of a cgo preamble):
the verifier context refuses to exist without a GPU, the host-side entry points work anywhere."""
import os
import shutil
import subprocess

import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not on PATH")
def test_c_caller_links_and_runs(tmp_path):
    ge.build()
    exe = str(tmp_path / "harness")
    libdir = os.path.join(ROOT, "bftkv_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_harness", "harness.c"),
                    "-L", libdir, "-lbftkv_gpu", "-Wl,-rpath," + libdir, "-o", exe], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    out = dict(line.split("=", 1) for line in r.stdout.decode().splitlines() if "=" in line)
    import torch
    if not torch.cuda.is_available():
        assert int(out["init_rc"]) < 0                      # BFTKV_E_DEVICE: no GPU, and no CPU fallback
    assert out["err_invalid"] == "crypto: invalid signature"
    assert out["err_insufficient"] == "crypto: insufficient number of signatures"
    assert out["packet"].startswith("0,0,0 len=32 x_len=3 v_len=5 t=42 tbs=32 has_sig=0")
    assert out["quorum n_qcs"] == "1 f=1 min=4 threshold=3 suff=3 n_nodes=4 suff3=1 suff_dup=0 thr3=1 reject2=1"
