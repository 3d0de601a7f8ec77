"""AddressSanitizer + UBSan run of the CPU oracle (sanitizers exist for the CPU build only)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan():
    d = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", d, "selftest_asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               OMP_NUM_THREADS="3")
    r = subprocess.run([os.path.join(d, "selftest_asan")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "oracle selftest ok" in r.stdout
