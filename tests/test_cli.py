"""CLI diff harness (counterpart of src/main.rs:34-116 + src/utils.rs:114-718)."""
import io

import pytest


def test_output_format_matches_reference_helpers():
    from fhestring_amd import cli
    out = io.StringIO()
    assert cli.compare_and_print(1, 1, out)
    assert not cli.compare_and_print("a\"b", "x\ny", out)
    assert out.getvalue() == 'Test Passed: OK, Result: 1, Test Failed: Expected: "a\\"b", Got: "x\\ny", '
    assert cli.rust_duration(1.5) == "1.5s" and cli.rust_duration(0.0123) == "12.3ms"
    assert len(cli.METHODS) == 52 and not cli.SKIPPED         # enum StringMethod, src/string_method.rs:2-55
    assert cli.rust_debug(["a", "b c"]) == '["a", "b c"]'


@pytest.mark.gpu
def test_readme_invocation_all_methods_pass(capsys):
    """README.md:50: --string "hello" --pattern "ello" --n 1 --from "ello" --to "_llo"."""
    from fhestring_amd import cli
    rc = cli.main(["--string", "hello", "--pattern", "ello", "--n", "1", "--from", "ello", "--to", "_llo"])
    text = capsys.readouterr().out
    assert rc == 0, text
    assert "Test Failed" not in text
    assert text.count("Test Passed: OK") >= len(cli.METHODS)
    assert 'Test Passed: OK, Result: "h_llo", Replace ' in text
    assert "Test Passed: OK, Result: 1, Find " in text
