"""CPU tier: the sanitizer runs of SURVEY.md section 5 ("race detection").  libapk's host side is threaded - proving slots taken by
concurrent callers (slot_gate.h), parked host threads for the [lin] combination (host_msm.h HostPool), a communicator with its own
sockets (comm.cpp) - and none of that needs a GPU, so it runs here under ThreadSanitizer and AddressSanitizer:

  * tools/san/host_hammer.cpp : SlotGate and HostPool + host_lincomb - the library's own headers - hammered from 32 threads;
  * tests/test_comm_world2.py : the multi-process communicator tests once more, against a libapk whose comm.cpp is compiled with
    -fsanitize=thread (algoplonk_amd/libapk_thread.so, `make SAN=thread san`), the interpreter started with libtsan preloaded.
A report from either sanitizer fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "algoplonk_amd", "csrc")


def _make(san, target="san-hammer"):
    r = subprocess.run(["make", "-C", CSRC, "SAN=%s" % san, target], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.parametrize("san", ["thread", "address"])
def test_host_threading_under_the_sanitizers(san):
    _make(san)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1:exitcode=67")
    r = subprocess.run([os.path.join(ROOT, "tools", "san", "host_hammer_%s" % san)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "SAN HAMMER OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr, r.stderr[-3000:]


def test_communicator_under_thread_sanitizer():
    # libapk_thread.so links the GPU backends' objects as they are: they exist wherever libapk.so was built (build()), and this test
    # must not start a ten-minute kernel compile of its own when they do not
    if not all(os.path.exists(os.path.join(CSRC, o)) for o in ("backend_bn254.o", "backend_bls12381.o", "apk_api.o", "verify_api.o")):
        pytest.skip("GPU backend objects not built here (run __graft_entry__.build() first)")
    _make("thread", "san")
    tsan = subprocess.check_output(["gcc", "-print-file-name=libtsan.so"], text=True).strip()
    if not os.path.isabs(tsan) or not os.path.exists(tsan):
        pytest.skip("no libtsan.so with this gcc")
    env = dict(os.environ, APK_LIB=os.path.join(ROOT, "algoplonk_amd", "libapk_thread.so"), LD_PRELOAD=tsan,
               TSAN_OPTIONS="halt_on_error=0 exitcode=66 report_signal_unsafe=0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_comm_world2.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "ThreadSanitizer" not in r.stdout + r.stderr, (r.stdout + r.stderr)[-4000:]
    assert " passed" in r.stdout
