"""Pins oracle/strings.py (the restatement of the reference's string algorithms)
to the reference's own test literals, on the clear-u8 char model."""
import pytest

from oracle import strings as ostr
from oracle.radix import ClearChar
from golden_util import load_vectors, python_expected, run_vector, check_vector

VECTORS = load_vectors()


def clear_env():
    ops = ostr.SplitOps(ClearChar)
    ops.trivial_char = lambda v: ClearChar(v)
    enc_s = lambda t, pad: [ClearChar(b) for b in ostr.pad_plain(t, pad)]
    enc_p = lambda t: [ClearChar(b) for b in ostr.pad_plain(t, 0)]
    enc_c = lambda v: ClearChar(v)
    dec_s = lambda s: ostr.truncate_plain([c.v for c in s])
    dec_c = lambda c: c.v
    return ops, enc_s, enc_p, enc_c, dec_s, dec_c


@pytest.mark.parametrize("v", VECTORS, ids=[v["name"] for v in VECTORS])
def test_fixture_transcription_matches_python_str(v):
    exp = python_expected(v)
    if exp is not None:
        assert exp == v["expected"]


@pytest.mark.parametrize("v", VECTORS, ids=[v["name"] for v in VECTORS])
def test_clear_model_golden(v):
    env = clear_env()
    if "expected_panic" in v:
        with pytest.raises(OverflowError, match=v["expected_panic"]):
            run_vector(v, *env)
        return
    check_vector(v, run_vector(v, *env))


def test_find_size_limit_panics_like_reference():
    # src/server_key/mod.rs:1025-1027
    ops, enc_s, enc_p, *_ = clear_env()
    with pytest.raises(OverflowError):
        ops.find(enc_s("a" * 257, 1), enc_p("abc"))


@pytest.mark.parametrize("s,p", [("", ""), ("abc", ""), ("", "a"), ("abc", "abcd"), ("aaa", "aa")])
def test_clear_model_edge_cases(s, p):
    ops, enc_s, enc_p, _, _, dec_c = clear_env()
    assert dec_c(ops.contains(enc_s(s, 1), enc_p(p))) == int(p in s)
    assert dec_c(ops.starts_with(enc_s(s, 1), enc_p(p))) == int(s.startswith(p))
    if p:
        assert dec_c(ops.find(enc_s(s, 1), enc_p(p))) == (s.find(p) if p in s else 255)
