"""the host half of the C-ABI library under AddressSanitizer + UndefinedBehaviorSanitizer.
`make san` builds libmolann_hip_san.so (host code instrumented, device code not); a child process with the ASan
runtime preloaded loads it and runs tests/san_driver.py.  CPU only - never on the GPU box (GPU ASan is not available
there, and this test needs no device)."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "molann_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang"


def test_host_half_under_asan_ubsan():
    if not os.path.exists(CLANG):
        pytest.skip("no ROCm clang")
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        pytest.skip("no ASan runtime")
    subprocess.check_call(["make", "-C", CSRC, "san"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=rt, MOLANN_SAN_LIB="1", PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "san_driver.py")], capture_output=True, text=True,
                       env=env, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "san_driver ok" in p.stdout
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr
