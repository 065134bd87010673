"""CPU sanitizer builds: AddressSanitizer + UndefinedBehaviorSanitizer over the oracle and
over the product's host-only code (client, key transforms, the whole DAG layer through a planner context).  GPU
sanitizers are not available on this pool, so nothing here touches a device."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def _run(make_dir, target, exe, ok):
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, make_dir), target],
                          stderr=subprocess.DEVNULL)
    p = subprocess.run([os.path.join(ROOT, exe)], capture_output=True, text=True, timeout=600, env=ENV)
    assert p.returncode == 0 and ok in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-4000:]


def test_oracle_under_asan_ubsan():
    _run("oracle", "asan", "oracle/_asan/oracle_asan_test", "oracle sanitizer run ok")


def test_host_code_under_asan_ubsan():
    _run("fhestring_amd/csrc", "asan", "fhestring_amd/asan_host_test", "host sanitizer run ok")
