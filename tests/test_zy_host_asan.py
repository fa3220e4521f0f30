"""Host-side memory safety of the C ABI (needs an MI355X and the prebuilt checker).

tests/native/sedmi_host_asan = the library's sources + tests/native/host_asan.cpp built by tools/build_host_asan.sh
with AddressSanitizer on the host code only.  It drives every create / pack / launch / destroy path and the error
paths of include/sedmi.h with exact-size heap buffers; a host-side access past a weight tensor, a table or a
workspace aborts it."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "native", "sedmi_host_asan")


@pytest.mark.timeout(300)
def test_every_entry_point_under_host_address_sanitizer():
    if not os.path.exists(BIN):
        pytest.skip("tests/native/sedmi_host_asan not built (tools/build_host_asan.sh)")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([BIN], env=env, capture_output=True, text=True, timeout=280)
    out = r.stdout + r.stderr
    started = "mel ok" in r.stdout or "FAILED" in out or "ERROR: AddressSanitizer" in out
    if r.returncode != 0 and not started:       # e.g. the sanitizer's shadow memory cannot be mapped on this host
        pytest.skip(f"the sanitizer build could not run on this host (exit {r.returncode}): {out[-300:]}")
    assert "ERROR: AddressSanitizer" not in out, out[-3000:]
    assert r.returncode == 0 and "host asan check: all ok" in r.stdout, out[-3000:]
