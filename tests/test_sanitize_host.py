"""Sanitizer + corruption leg of the host C-ABI.  CPU only: the three HIP-free translation units of
libcovahip.so (hostlib.cpp, h264_front.cpp, h264_cabac.cpp -- everything that parses untrusted bytes: bincode, MP4 boxes, avcC,
NAL units, CABAC slice data) are built with -fsanitize=address,undefined (make -C cova_amd/csrc host-san) and loaded, through
the test hook COVAHIP_HOST_SAN_LIB of cova_amd/_lib.py, into a python that has the ASan runtime preloaded.

* the test_host_* suites run against that build: no sanitizer report on any input the functional tests use;
* tests/helpers/san_fuzz.py: >= 10,000 seeded corruptions; every call returns a status, nothing crashes, reads or writes out
  of bounds, overflows a signed integer or shifts out of range.

(GPU AddressSanitizer is not available on this pool; the kernels' operand shapes are validated on the host,
tests/test_gpu_errors.py.)"""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_LIB = os.path.join(ROOT, "cova_amd", "libcovahost_san.so")


def _asan_runtime():
    try:
        p = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True, timeout=30).stdout.strip()
        return p if os.path.isabs(p) and os.path.exists(p) else None
    except (OSError, subprocess.SubprocessError):
        return None


ASAN = _asan_runtime()
pytestmark = [pytest.mark.sanitize,
              pytest.mark.skipif(ASAN is None, reason="no ASan runtime for g++ in this environment")]


@pytest.fixture(scope="module")
def san_env():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "cova_amd", "csrc"), "host-san"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and os.path.exists(SAN_LIB), r.stderr[-2000:]
    env = dict(os.environ)
    env.update({
        "LD_PRELOAD": ASAN,                          # the ASan runtime has to be the first library of the process
        "COVAHIP_HOST_SAN_LIB": SAN_LIB,
        # leaks: the interpreter and numpy "leak" by design; everything else aborts the process on the first report
        "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1:allocator_may_return_null=1",
        "UBSAN_OPTIONS": "halt_on_error=1:abort_on_error=1:print_stacktrace=1",
        "OMP_NUM_THREADS": "1",
    })
    return env


def test_host_suites_under_asan_ubsan(san_env):
    """Every CPU test of the host objects (bbox / frame wire format, metapreprocess ring, SORT, cova GoP filter, sink formats,
    aggregator rules, MP4 / H.264 headers / CABAC on the demo stream) against the sanitizer build."""
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "test_host_*.py")))
    assert len(files) >= 8
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", *files],
                       env=san_env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = (r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail


@pytest.mark.parametrize("seed", [1])
def test_seeded_corruption_returns_statuses(san_env, seed):
    """bincode vectors / frames / track exports / detector text (8,000 cases), MP4 boxes (1,200), avcC and access units (1,200),
    corrupted by flips, extreme 32-bit fields, truncation and random runs: 10,400 cases, all answered with a status."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "san_fuzz.py"), str(seed)],
                       env=san_env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-4000:])
    counts = json.loads(r.stdout.strip().splitlines()[-1])
    assert counts["bincode_vec_frame_tracks_text"] >= 8000
    if os.path.exists("/root/reference/demo/1m.mp4"):
        assert counts["total_cases"] >= 10000 and counts["mp4_boxes"] >= 1000 and counts["avcc_and_access_units"] >= 1000
        # the corruptions are not all trivially rejected at the first byte: some files still open, some access units still parse
        assert counts["mp4_opened_despite_corruption"] > 0 and counts["access_units_accepted"] > 0
