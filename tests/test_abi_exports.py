"""CPU: the C-ABI library loads and exports every symbol include/bftkv_gpu.h declares (no compute calls:
there is no GPU here), and the product path fails loudly instead of falling back to the CPU."""
import os
import re

import pytest

import __graft_entry__ as ge
from bftkv_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    ge.build()
    return _native.load_library()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "bftkv_gpu.h")).read()
    declared = set(re.findall(r"\b(bftkv_gpu_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_native.EXPORTS)


def test_error_strings_are_the_reference_singletons(lib):
    # crypto/crypto.go:19-20; these strings travel in the X-error header (transport/http/http.go:145)
    assert lib.bftkv_gpu_error_string(1) == b"crypto: invalid signature"
    assert lib.bftkv_gpu_error_string(2) == b"crypto: insufficient number of signatures"
    assert lib.bftkv_gpu_error_string(0) == b""


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_native.NativeError):
        _native.Context(0)


def test_product_package_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bftkv_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_headers_are_plain_c():
    """The drop-in boundary is a C ABI: both headers must compile as C99 on their own (what cgo's preamble does)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not on PATH")
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    for h in ("bftkv_gpu.h", "bftkv_host.h"):
        r = subprocess.run(["gcc", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", os.path.join(inc, h)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()


@pytest.mark.parametrize("src", ["tools/serving/batcher_load.c", "tools/serving/threshold_load.c", "tools/serving/cert_load.c", "tools/fakehip/stress.c"])
def test_plain_c_callers_of_the_serving_entries_compile_and_link(src, tmp_path):
    """The load generators bench.py and the profiles run on the GPU box (and the sanitizer stress driver) are plain C against
    include/bftkv_gpu.h alone: they must build without warnings and resolve every symbol in the library."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not on PATH")
    exe = str(tmp_path / "a.out")
    r = subprocess.run(["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-std=gnu11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, src),
                        "-L", os.path.join(ROOT, "bftkv_amd"), "-lbftkv_gpu", "-lpthread", "-Wl,-rpath," + os.path.join(ROOT, "bftkv_amd"), "-o", exe],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_the_stand_in_runtime_is_test_infrastructure_only():
    """tools/fakehip (a HIP runtime whose kernels do not run, for the sanitizer builds) is never part of the product: nothing under
    bftkv_amd/, the build entry or the bench names it, and the library the package loads depends on the real runtime."""
    import subprocess
    for top in ("bftkv_amd", "__graft_entry__.py", "bench.py", "include"):
        p = os.path.join(ROOT, top)
        files = [p] if os.path.isfile(p) else [os.path.join(d, f) for d, _, fs in os.walk(p) for f in fs if not f.endswith((".so", ".pyc"))]
        for f in files:
            with open(f, "rb") as fh:
                assert b"fakehip" not in fh.read(), f
    ge.build()
    needed = subprocess.run(["readelf", "-d", os.path.join(ROOT, "bftkv_amd", "libbftkv_gpu.so")], capture_output=True, text=True).stdout
    assert "libamdhip64.so" in needed and "fake" not in needed, needed
