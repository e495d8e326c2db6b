"""The oracle's whole path under AddressSanitizer + UndefinedBehaviorSanitizer ("run our CPU restatement
under -fsanitize=address,undefined"): any out-of-bounds access, signed overflow, misaligned or invalid conversion in the
checker aborts the run (-fno-sanitize-recover=all)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_path_is_clean_under_asan_and_ubsan():
    d = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", d, "-s", "sanitize_main"])
    r = subprocess.run([os.path.join(d, "sanitize_main")], capture_output=True, text=True,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and r.stdout.startswith("ok "), (r.stdout + r.stderr)[-1500:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
