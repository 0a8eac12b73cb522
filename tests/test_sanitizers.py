"""The threaded host code under ThreadSanitizer and AddressSanitizer + UBSan ("run host tests under
TSan/ASan").  tests/sanitize/host_threads.cpp is built from the product's host SOURCES with -fsanitize=... and exercises the
forking kd / BVH builders, ParallelSlices, the flattener and re::HipRaytracer's render-thread / condvar / promise lifetime
(without a GPU the device's refusal travels through the promise; with one the frames render)."""
import os
import subprocess

import pytest

from conftest import ROOT

DIR = os.path.join(ROOT, "tests", "sanitize")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", DIR], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize("binary,marker", [("host_threads_tsan", "ThreadSanitizer"), ("host_threads_asan", "Sanitizer")])
def test_threaded_host_code_is_clean_under(built, binary, marker):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66", ASAN_OPTIONS="detect_leaks=0 exitcode=67", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(DIR, binary)], capture_output=True, text=True, timeout=600, env=env, cwd=DIR)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "host_threads ok" in out, out[-4000:]
    assert marker not in out and "runtime error" not in out, out[-4000:]
