"""The host side of the library entered by many threads at once, without a GPU: tools/tsan_host.sh builds csrc/capi.hip for the host
with ThreadSanitizer over tools/fakehip (a stand-in HIP runtime whose kernels do not run) and drives every micro-batcher entry,
pipelined host-buffer calls on forked contexts and a keyring / quorum writer concurrently (tools/fakehip/stress.c) -- the shape of
protocol.Server's goroutine per request (transport/http/http.go:85,143 -> protocol/server.go:562-620).  Checked: every call returns,
a call that fails leaves a failing status byte, ThreadSanitizer prints nothing.  Answers are not checked here (no kernels): the
-m gpu tests do that through the same entries."""
import glob
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None or not glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so"),
                    reason="needs hipcc and clang's ThreadSanitizer runtime")
def test_every_batcher_entry_under_thread_sanitizer(tmp_path):
    env = dict(os.environ, BFTKV_TSAN_QUICK="1")
    env.pop("LD_PRELOAD", None)        # (tools/sanitize_host.sh runs this suite under another sanitizer's runtime)
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "tsan_host.sh"), str(tmp_path), "6"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ThreadSanitizer: no report" in r.stdout
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][0]
    got = json.loads(line)
    assert got["failed_open"] == 0
    # the register of accepted request certificates (host logic, no kernel needed): the second request with a certificate comes
    # from the register, a keyring change forgets it, the two requests after it repeat that
    assert got["register"] == [0, 1, 0, 1], got["register"]
    kinds = ("collective", "signature", "certificate", "modmul_product", "lagrange_combine", "dsa_calculate_r", "modexp", "host_buffer_call",
             "keyring_set", "quorum_create_destroy")
    for kind in kinds:
        assert got[kind]["rc_nonzero"] == 0, (kind, got[kind])
    # how far each kind gets in a few seconds under the sanitizer depends on the machine: every kind normally runs, the mix must
    assert got["collective"]["calls"] > 0 and sum(got[k]["calls"] for k in kinds) + got["message"]["calls"] > 10, got
    # (the message walk kernel does not run here: those calls fail, closed -- "failed_open" above)
