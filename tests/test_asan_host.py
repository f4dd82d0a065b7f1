"""Sanitizer build of the host side (memory / UB checkers; asked for since round 1): libstego_corr.so compiled with
-fsanitize=address,undefined for its HOST code (stego_amd._build.build_asan; the device code is the product's), and the tests of the
C ABI's host logic - descriptor validation, workspace geometry, every error path that returns before a launch, the symbol table of
include/*.h, the ViT / head / KNN / dense-correlation argument checks - run through it in a python that preloads the sanitizer runtime.
A heap overflow in a geometry computation, a use of an uninitialised descriptor field or signed overflow in a size would fail here."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_host_logic_under_address_and_ub_sanitizers():
    sys.path.insert(0, ROOT)
    from stego_amd import _build
    rt = _build.asan_runtime()
    if rt is None:
        pytest.skip("this toolchain ships no AddressSanitizer runtime")
    lib = _build.build_asan()
    # the build really is instrumented
    syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
    assert "__asan_init" in syms and "stego_corr_fwd" in syms
    env = dict(os.environ, LD_PRELOAD=rt, STEGO_LIB_PATH=lib,
               ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    files = [os.path.join(ROOT, "tests", f) for f in ("test_host_logic.py", "test_vit_native.py", "test_head_native.py", "test_knn.py",
                                                       "test_dense_corr.py")]
    # (test_missing_library_fails_loudly swaps the library path itself: not meaningful with the override)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", "-k", "not missing_library"] + files
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    text = out.stdout + out.stderr
    assert "AddressSanitizer" not in text and "runtime error:" not in text, text[-4000:]
    assert out.returncode == 0, text[-4000:]
    assert " passed" in out.stdout
