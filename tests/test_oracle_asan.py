"""CPU sanitizer leg: the oracle rebuilt with -fsanitize=address,undefined and the oracle-facing CPU tests
re-run against it in a child process (the sanitizer runtime has to be preloaded before python starts).  CPU only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_suite_under_asan_and_ubsan():
    if os.environ.get("F1P_ORACLE_LIB"):
        pytest.skip("already inside the sanitizer leg")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan.so in this image")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, F1P_ORACLE_LIB=os.path.join(ROOT, "oracle", "liborc_asan.so"), LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2")
    files = ["tests/test_oracle_golden.py", "tests/test_oracle_clothoid.py", "tests/test_oracle_grid.py", "tests/test_host_logic.py"]
    files = [f for f in files if os.path.exists(os.path.join(ROOT, f))]
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + files, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-4000:]
    assert "passed" in p.stdout and "AddressSanitizer" not in p.stdout and "runtime error" not in p.stdout, p.stdout[-4000:]
