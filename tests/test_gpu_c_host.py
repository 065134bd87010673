"""The drop-in boundary from a compiled host: examples/c_host.c (plain C99, no Python or C++ on its side of the C ABI)
runs the reference's flow -- client key, encrypt with padding, MyServerKey methods, decrypt -- on the MI355X and checks
every result against the clear computation, printing `Test Passed: OK, Result: ...` lines like src/utils.rs:114-120."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("text,pat", [("the quick brown fox jumps over the lazy dog", "brown"),
                                      ("abcabcabcabd", "cabd"), ("no match in here", "zama"), ("aaaa", "aa")])
def test_c_host_runs_the_reference_flow(text, pat):
    exe = os.path.join(ROOT, "examples", "c_host")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    out = subprocess.run([exe, text, pat], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    assert [l.split(":")[0] for l in lines[:6]] == ["contains_clear", "contains", "find", "len", "to_upper", "replace"]
    assert all("Test Passed: OK" in l for l in lines[:6]) and "FAILED" not in out.stdout
    want_find = text.find(pat) if pat in text else 255
    assert "Result: %d, Expected: %d" % (want_find, want_find) in lines[2]
    # replace with |from| >= |to| follows the REFERENCE (mod.rs:828-882: every matching window of the original text is
    # overwritten, later over earlier), which is str.replace only where occurrences do not overlap
    want_replace = "<<<>" if (text, pat) == ("aaaa", "aa") else text.replace(pat, "<>")
    assert 'Result: "%s"' % want_replace in lines[5]
    assert lines[6].startswith("PBS executed:")
