"""ASan + UBSan runs of the CPU-side code (the reference has no race/sanitizer tooling; GPU
sanitizers are not available on the pool, so these cover the oracle and the host half of the library)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan-test"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 mismatches" in out.stdout and "ERROR" not in out.stderr


def test_host_library_under_asan_ubsan():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "hare_amd", "csrc"), "-s", "asan-test"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failures" in out.stdout and "ERROR" not in out.stderr
