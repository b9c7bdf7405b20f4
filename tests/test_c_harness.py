"""CPU: a plain C99 program linked against libbftkv_gpu.so through include/*.h only (the position of a cgo preamble):
the verifier context refuses to exist without a GPU, the host-side entry points work anywhere."""
import os
import shutil
import subprocess

import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(tmp_path, *args):
    ge.build()
    exe = str(tmp_path / "harness")
    libdir = os.path.join(ROOT, "bftkv_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_harness", "harness.c"),
                    "-L", libdir, "-lbftkv_gpu", "-Wl,-rpath," + libdir, "-o", exe], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    r = subprocess.run([exe] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    return dict(line.split("=", 1) for line in r.stdout.decode().splitlines() if "=" in line)


def test_fixture_header_is_current():
    """fixture.h is generated: the committed text must be what the committed generator produces."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_fixture", os.path.join(ROOT, "tests", "c_harness", "make_fixture.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert open(os.path.join(ROOT, "tests", "c_harness", "fixture.h")).read() == m.render()


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not on PATH")
def test_c_caller_verifies_on_the_gpu(tmp_path):
    """A plain C99 caller -- the position of the cgo shim -- uploads a keyring, creates a quorum and verifies signed writes
    on the MI355X, batched and through the micro-batcher; its verdicts are the oracle's."""
    import importlib.util
    from tests import helpers as H
    spec = importlib.util.spec_from_file_location("make_fixture", os.path.join(ROOT, "tests", "c_harness", "make_fixture.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    cl, c = m.corpus()
    kr, q = H.oracle_keyring(cl), H.clique_quorum(cl)
    want = [H.oracle_collective(kr, q, c, i) for i in range(c.n_items)]
    out = _build_and_run(tmp_path, "gpu")
    assert out["init_rc"] == "0" and out["gpu_keyring_set_rc"] == "0" and out["gpu_quorum_create_rc"] == "0" and out["gpu_verify_rc"] == "0"
    err = [int(v) for v in out["gpu_verify_err"].split(",")]
    assert err == [0 if r.err is None else 2 for r in want] and 0 in err and 2 in err
    assert [int(v) for v in out["gpu_verify_nver"].split(",")] == [len(r.verified) for r in want]
    assert out["gpu_verify_fenced"] == ",".join(["0"] * c.n_items)
    assert out["gpu_first_error_string"] == "crypto: insufficient number of signatures"
    assert out["gpu_batcher_rc"] == "0" and [int(v) for v in out["gpu_batcher_err"].split(",")] == err
    assert out["gpu_signature_verify"] == "0,0,0"
    assert out["gpu_fail_closed"] == "1,2"
    assert out["gpu_pipelined"] == "0,1"                      # three pieces on worker contexts: byte for byte the unsplit answers
    assert out["gpu_cert_verify"] == "0,0,0,1"                # a stranger's certificate out of the request: issuer found, signature good
    assert out["gpu_cert_verify_forged"] == "0,3,0"           # BFTKV_ERR_CERTIFICATE_NOT_FOUND: ReadEntity would refuse it
    assert out["gpu_cert_verify_other_bytes"] == "0,1,0"      # crypto.ErrInvalidSignature
    # config 5, one share-combine operation per call (bftkv_gpu_batcher_modmul_product / _lagrange_combine / _modexp)
    assert out["gpu_th_product"] == "0,0,1"                   # (N-1)^3 = N-1 mod N
    assert out["gpu_th_lagrange"] == "0,0,1,7"                # f(0) of f(x) = 7 + 3x + 2x^2 out of f(1), f(2), f(3)
    assert out["gpu_th_modexp"] == "0,0,1,243"
    assert out["gpu_th_fail_closed"] == "-4,255,1"            # even modulus: BFTKV_E_UNSUPPORTED, status BFTKV_TH_FAILED, zeroes


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not on PATH")
def test_c_caller_links_and_runs(tmp_path):
    out = _build_and_run(tmp_path)
    import torch
    if not torch.cuda.is_available():
        assert int(out["init_rc"]) < 0                      # BFTKV_E_DEVICE: no GPU, and no CPU fallback
    assert out["err_invalid"] == "crypto: invalid signature"
    assert out["err_insufficient"] == "crypto: insufficient number of signatures"
    assert out["packet"].startswith("0,0,0 len=32 x_len=3 v_len=5 t=42 tbs=32 has_sig=0")
    assert out["quorum n_qcs"] == "1 f=1 min=4 threshold=3 suff=3 n_nodes=4 suff3=1 suff_dup=0 thr3=1 reject2=1"
    # PGPSignature.Signers' walk on the host, from plain C (bftkv_host_signers_walk), against the oracle's walk of the same stream
    import importlib.util
    from oracle import collective as col
    from oracle.packet import SignaturePacket
    from tests import helpers as H
    spec = importlib.util.spec_from_file_location("make_fixture", os.path.join(ROOT, "tests", "c_harness", "make_fixture.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    cl, c = m.corpus()

    class Everyone:
        def get_cert_by_id(self, i):
            return type("E", (), {"id": i})()
    s0 = c.ss_data(0)
    every = col.signers(Everyone(), SignaturePacket(1, 0, False, s0, None))
    ring = col.signers(H.oracle_keyring(cl), SignaturePacket(1, 0, False, s0, None))
    cut = col.signers(Everyone(), SignaturePacket(1, 0, False, s0[:-5], None))
    assert out["host_signers_walk"] == "0,%d,%d,0 cut=0,%d,0" % (len(every), len(ring), len(cut)) and len(cut) == len(every) - 1
