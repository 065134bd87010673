"""The product's string layer evaluated on the CPU by constant folding (no GPU, no oracle in the product path).

Trivially encrypted inputs make every bootstrap of a DAG fold to its look-up value (Engine::pbs on a trivial block,
luts.h: lut_eval with the padding-bit rule) and every linear combination to a constant mod 32, so a planner context
computes what the DAG means.  `fhs_trivial_value` reads the result.  That checks the LOGIC of both formulations --
weights, look-up tables, trees, the negacyclic tricks -- against

* the reference's own test literals (tests/golden/ref_tests.json), both modes, and
* the loop-for-loop restatement of the reference on the clear-u8 char model (oracle/strings.py with ClearChar) on
  random strings, patterns and paddings, and on every byte pair for the character-level formulations.

Noise bookkeeping and the kernels are the GPU tests' business."""
import random

import pytest

from oracle import strings as ostr
from oracle.radix import ClearChar
from golden_util import load_vectors, run_vector, check_vector

VECTORS = load_vectors()
WS = " \t\n\x0b\x0c\r"


@pytest.fixture(scope="module")
def folded():
    from fhestring_amd.api import MyServerKey
    sk = MyServerKey.planner()
    sk.trivial_char = sk.trivial
    yield sk
    sk.close()


def product_env(sk):
    from fhestring_amd.api import FheString

    def dec_c(c):
        v = c.trivial_value()
        assert v is not None, "the result did not fold to a constant"
        return v
    enc_s = lambda t, pad: FheString([sk.trivial(b) for b in ostr.pad_plain(t, pad)])
    enc_p = lambda t: FheString([sk.trivial(b) for b in ostr.pad_plain(t, 0)])
    dec_s = lambda s: ostr.truncate_plain([dec_c(c) for c in s.chars])
    return sk, enc_s, enc_p, sk.trivial, dec_s, dec_c


def clear_env():
    ops = ostr.SplitOps(ClearChar)
    ops.trivial_char = lambda v: ClearChar(v)
    enc_s = lambda t, pad: [ClearChar(b) for b in ostr.pad_plain(t, pad)]
    enc_p = lambda t: [ClearChar(b) for b in ostr.pad_plain(t, 0)]
    return ops, enc_s, enc_p, ClearChar, lambda s: ostr.truncate_plain([c.v for c in s]), lambda c: c.v


@pytest.mark.parametrize("mode", [0, 1], ids=["as_written", "fused"])
@pytest.mark.parametrize("v", VECTORS, ids=[v["name"] for v in VECTORS])
def test_golden_vectors_by_constant_folding(folded, v, mode):
    folded.set_mode(mode)
    env = product_env(folded)
    if "expected_panic" in v:
        with pytest.raises(OverflowError, match=v["expected_panic"]):
            run_vector(v, *env)
        return
    check_vector(v, run_vector(v, *env))
    assert folded.stats(reset=True)["pbs_executed"] == 0         # everything folded: nothing was left pending


def _rand_text(rnd, n, alphabet):
    return "".join(rnd.choice(alphabet) for _ in range(n))


def _cases(seed, count):
    """Random vectors in the golden format: short strings over a small alphabet (so that patterns occur, repeat and
    overlap), whitespace, upper / lower case, paddings 0-3."""
    rnd = random.Random(seed)
    abc = "abAB zZ\t_a"
    out = []
    for k in range(count):
        n = rnd.randint(0, 9)
        s = _rand_text(rnd, n, abc)
        pad = rnd.randint(1, 3)
        if s and rnd.random() < 0.6:
            i = rnd.randrange(len(s))
            p = s[i:i + rnd.randint(1, 3)]
        else:
            p = _rand_text(rnd, rnd.randint(1, 3), abc)
        other = s if rnd.random() < 0.2 else (s.swapcase() if rnd.random() < 0.25 else _rand_text(rnd, rnd.randint(0, 9), abc))
        if rnd.random() < 0.3 and s:
            other = s[:rnd.randint(0, len(s))] + _rand_text(rnd, rnd.randint(0, 2), abc)
        to = _rand_text(rnd, rnd.randint(0, 4), abc)
        base = {"string": s, "pad": pad, "name": "rand%d" % k}
        for op in ("contains", "starts_with", "ends_with", "find", "rfind"):
            out.append(dict(base, op=op, pattern=p))
        for op in ("to_upper", "to_lower", "trim_end", "trim_start", "trim", "is_empty", "len"):
            out.append(dict(base, op=op))
        for op in ("eq", "eq_ignore_case", "lt", "le", "gt", "ge", "concatenate"):
            out.append(dict(base, op=op, other=other, other_pad=rnd.randint(1, 3)))
        out.append(dict(base, op="replace", **{"from": p, "to": to}))
        out.append(dict(base, op="replacen", n=rnd.randint(0, 3), **{"from": p, "to": to}))
        out.append(dict(base, op="repeat", n=rnd.randint(0, 3)))
        out.append(dict(base, op="strip_prefix", pattern=p))
        out.append(dict(base, op="strip_suffix", pattern=p))
    return out


def _same(v, got, want):
    assert got == want, (v["op"], {k: v[k] for k in v if k not in ("name",)}, "product", got, "reference model", want)


@pytest.mark.parametrize("mode", [0, 1], ids=["as_written", "fused"])
def test_random_strings_against_the_clear_model(folded, mode):
    """Both formulations mean what the reference's loops mean (oracle/strings.py on plain bytes), op by op."""
    folded.set_mode(mode)
    penv, cenv = product_env(folded), clear_env()
    for v in _cases(20261004 + mode, 60 if mode == 1 else 12):   # as written is O(n^2) host work per op: fewer, same shapes
        try:
            want = run_vector(v, *cenv)
        except OverflowError:
            with pytest.raises(OverflowError):
                run_vector(v, *penv)
            continue
        _same(v, run_vector(v, *penv), want)


def test_split_family_against_the_clear_model(folded):
    folded.set_mode(1)
    penv, cenv = product_env(folded), clear_env()
    rnd = random.Random(77)
    for k in range(25):
        s = _rand_text(rnd, rnd.randint(1, 9), "ab, a")
        p = rnd.choice([",", " ", "a", "ab", ", "])
        base = {"string": s, "pad": rnd.randint(1, 2), "name": "split%d" % k, "pattern": p}
        for op in ("split", "split_inclusive", "split_terminator", "rsplit", "rsplit_terminator", "rsplit_once"):
            if op == "rsplit_once" and p not in s:
                continue
            v = dict(base, op=op)
            _same(v, run_vector(v, *penv), run_vector(v, *cenv))
        for op in ("splitn", "rsplitn"):
            v = dict(base, op=op, n=rnd.randint(1, 4))
            _same(v, run_vector(v, *penv), run_vector(v, *cenv))
        v = {"string": s.replace(",", " "), "pad": 1, "name": "ws%d" % k, "op": "split_ascii_whitespace"}
        _same(v, run_vector(v, *penv), run_vector(v, *cenv))


def test_every_byte_pair_of_the_character_level_formulations(folded):
    """eq_ignore_case, the comparisons and eq on one-character buffers for all 65 536 byte pairs (NUL included: a
    buffer may hold it); case folds, whitespace and NUL tests on all 256 bytes."""
    from fhestring_amd.api import FheString
    sk = folded
    sk.set_mode(1)
    tr = [sk.trivial(v) for v in range(256)]
    low = lambda x: x + 32 if 0x41 <= x <= 0x5A else x
    for a in range(256):
        sa = FheString([tr[a]])
        for b in range(256):
            sb = FheString([tr[b]])
            got = (sk.eq_ignore_case(sa, sb).trivial_value(), sk.le(sa, sb).trivial_value(), sk.gt(sa, sb).trivial_value(),
                   sk.eq(sa, sb).trivial_value())
            assert got == (int(low(a) == low(b)), int(a <= b), int(a > b), int(a == b)), (a, b, got)
    for a in range(256):
        s = FheString([tr[0x41], tr[a], tr[0x42]])
        up, lo = sk.to_upper(s)[1].trivial_value(), sk.to_lower(s)[1].trivial_value()
        assert up == (a - 32 if 0x61 <= a <= 0x7A else a) and lo == low(a), (a, up, lo)
        # x?y: a NUL or whitespace byte in the middle is kept by the trims; at the end it is trimmed (NUL: it is padding)
        t = FheString([tr[0x41], tr[a]])
        got = [c.trivial_value() for c in sk.trim_end(t).chars]
        assert got == ([0x41, 0] if (a == 0 or chr(a) in WS) else [0x41, a]), (a, got)
        assert sk.len(t).trivial_value() == (1 if a == 0 else 2) and sk.is_empty(FheString([tr[a]])).trivial_value() == int(a == 0)


def test_comparisons_follow_the_first_difference_on_long_buffers(folded):
    """The three-state tree over many positions: the first differing nibble decides, buffers of different length compare
    through the longer one's tail."""
    from fhestring_amd.api import FheString
    sk = folded
    sk.set_mode(1)
    rnd = random.Random(3)
    for _ in range(40):
        n, m = rnd.randint(1, 60), rnd.randint(1, 60)
        a = [rnd.choice([0x61, 0x62, 0x7A]) for _ in range(n)]
        b = list(a[:m]) + [rnd.choice([0x61, 0x62]) for _ in range(max(0, m - n))]
        if rnd.random() < 0.7 and b:
            b[rnd.randrange(len(b))] = rnd.choice([0x60, 0x63])
        pa, pb = rnd.randint(0, 3), rnd.randint(0, 3)
        sa = FheString([sk.trivial(v) for v in a + [0] * pa])
        sb = FheString([sk.trivial(v) for v in b + [0] * pb])
        ba, bb = bytes(a), bytes(b)
        got = tuple(f(sa, sb).trivial_value() for f in (sk.lt, sk.le, sk.gt, sk.ge, sk.eq))
        assert got == (int(ba < bb), int(ba <= bb), int(ba > bb), int(ba >= bb), int(ba == bb)), (a, b, pa, pb, got)


def test_find_at_every_position_of_a_256_char_string(folded):
    """BASELINE config 3's shape: the thermometer index over 254 windows (17 chunks of the prefix-OR tree: the 16th chunk
    takes its predecessor's prefix and total, the last positions sit there), a match planted at every position in turn,
    a second match behind it, and the miss."""
    from fhestring_amd.api import FheString
    sk = folded
    sk.set_mode(1)
    pat = FheString([sk.trivial(ord(c)) for c in "wxyz"])
    base = [ord("a") + (i % 3) for i in range(256)]
    tr = {v: sk.trivial(v) for v in set(base) | set(b"wxyz") | {0}}
    for pos in list(range(0, 253)) + [None]:
        t = list(base)
        if pos is not None:
            t[pos:pos + 4] = b"wxyz"
            if pos + 9 <= 252:
                t[pos + 5:pos + 9] = b"wxyz"             # a later match must not disturb the first
        s = FheString([tr[v] for v in t] + [tr[0]])
        assert sk.find(s, pat).trivial_value() == (255 if pos is None else pos), pos
    for pos in (0, 100, 239, 240, 252):                   # rfind: the last match
        t = list(base)
        t[pos:pos + 4] = b"wxyz"
        if pos >= 8:
            t[pos - 8:pos - 4] = b"wxyz"
        s = FheString([tr[v] for v in t] + [tr[0]])
        assert sk.rfind(s, pat).trivial_value() == pos, pos


def test_compaction_against_the_bubble_on_strings_with_holes(folded):
    """bubble_zeroes_right: the routing network (prefix counts, LSB-first conditional moves) against the reference's
    bubble (utils.rs:28-46) on buffers with NULs anywhere, lengths across several powers of two."""
    from fhestring_amd.api import FheString
    sk = folded
    sk.set_mode(1)
    rnd = random.Random(11)
    ops = clear_env()[0]
    for n in (1, 2, 3, 7, 8, 9, 31, 33, 64, 100):
        for density in (0.1, 0.5, 0.9):
            vals = [0 if rnd.random() < density else rnd.randint(1, 255) for _ in range(n)]
            got = [c.trivial_value() for c in sk.bubble_zeroes_right(FheString([sk.trivial(v) for v in vals])).chars]
            want = [c.v for c in ops.bubble_zeroes_right([ClearChar(v) for v in vals])] if n <= 33 else \
                [v for v in vals if v] + [0] * vals.count(0)
            assert got == want, (n, vals, got)


def test_long_equalities_and_case_insensitive_equalities(folded):
    from fhestring_amd.api import FheString
    sk = folded
    sk.set_mode(1)
    rnd = random.Random(19)
    for _ in range(30):
        n = rnd.randint(1, 80)
        a = [rnd.choice(b"abcXYZ_ 09") for _ in range(n)]
        b = list(a)
        kind = rnd.random()
        if kind < 0.3:
            b = [v ^ 0x20 if chr(v).isalpha() and rnd.random() < 0.5 else v for v in b]    # case only
        elif kind < 0.6:
            b[rnd.randrange(n)] ^= 0x20                       # one byte: a letter's case, or '_' / ' ' / digit changed
        elif kind < 0.8:
            b = b[:rnd.randint(0, n)]                         # a prefix
        sa = FheString([sk.trivial(v) for v in a + [0] * rnd.randint(0, 3)])
        sb = FheString([sk.trivial(v) for v in b + [0] * rnd.randint(0, 3)])
        ba, bb = bytes(a), bytes(b)
        got = (sk.eq(sa, sb).trivial_value(), sk.ne(sa, sb).trivial_value(), sk.eq_ignore_case(sa, sb).trivial_value())
        assert got == (int(ba == bb), int(ba != bb), int(ba.lower() == bb.lower())), (a, b, got)


def test_window_counts_around_the_chunk_boundaries_of_the_prefix_tree(folded):
    """find / rfind / contains with W windows for W around multiples of 15 and 225 (chunking of the prefix-OR tree, the
    16th-chunk shortcut, the switch to the one-hot route above 256 windows): first, last and no match."""
    from fhestring_amd.api import FheString
    sk = folded
    sk.set_mode(1)
    pat = FheString([sk.trivial(ord(c)) for c in "xy"])
    tr = {v: sk.trivial(v) for v in (0, ord("a"), ord("x"), ord("y"))}
    cops = clear_env()[0]
    for W in (1, 2, 14, 15, 16, 17, 29, 30, 31, 45, 224, 225, 226, 239, 240, 241, 254, 255):
        n = W + 1                                             # text length: W windows of 2
        for pos in (None, 0, W // 2, W - 1):
            t = [ord("a")] * n
            if pos is not None:
                t[pos:pos + 2] = b"xy"
            s = FheString([tr[v] for v in t] + [tr[0]])
            want = 255 if pos is None else pos
            cs, cp = [ClearChar(v) for v in t + [0]], [ClearChar(ord(c)) for c in "xy"]
            for name in ("find", "rfind"):                    # the reference panics at len >= 255 + m (mod.rs:1025-1027;
                try:                                          # rfind appends a NUL first, :737): follow the clear model
                    ref = getattr(cops, name)(cs, cp).v
                except OverflowError:
                    with pytest.raises(OverflowError):
                        getattr(sk, name)(s, pat)
                    continue
                assert ref == want and getattr(sk, name)(s, pat).trivial_value() == want, (name, W, pos)
            assert sk.contains(s, pat).trivial_value() == int(pos is not None), (W, pos)


def test_longer_random_strings_against_the_clear_model(folded):
    """The same differential on longer strings (routing network over several stages, greedy non-overlapping matches of
    longer patterns, `to` longer and shorter than `from`, counters)."""
    folded.set_mode(1)
    penv, cenv = product_env(folded), clear_env()
    rnd = random.Random(4242)
    for k in range(12):
        s = _rand_text(rnd, rnd.randint(20, 45), "ab ab_A")
        i = rnd.randrange(len(s) - 4)
        p = s[i:i + rnd.randint(1, 4)]
        base = {"string": s, "pad": rnd.randint(1, 2), "name": "long%d" % k}
        vs = [dict(base, op="replace", **{"from": p, "to": _rand_text(rnd, rnd.randint(0, 6), "xyz")}),
              dict(base, op="replacen", n=rnd.randint(0, 4), **{"from": p, "to": _rand_text(rnd, rnd.randint(0, 6), "xyz")}),
              dict(base, op="find", pattern=p), dict(base, op="rfind", pattern=p), dict(base, op="trim"),
              dict(base, op="len"), dict(base, op="strip_suffix", pattern=s[-3:]), dict(base, op="strip_prefix", pattern=s[:3]),
              dict(base, op="split", pattern=p), dict(base, op="rsplitn", pattern=p, n=rnd.randint(1, 3)),
              dict(base, op="le", other=s[:i] + "b" + s[i + 1:], other_pad=1),
              dict(base, op="eq_ignore_case", other=s.swapcase(), other_pad=2)]
        for v in vs:
            _same(v, run_vector(v, *penv), run_vector(v, *cenv))


def test_position_sharded_comparison_partials_combine_to_the_whole(folded):
    """Multi-GPU position sharding of lt / le / gt / ge (fhs_str_compare_partial per range, fhs_flags_first_decides over
    the ranges): any split of two equally long buffers gives the verdict of the whole comparison."""
    from fhestring_amd.api import FheString
    sk = folded
    sk.set_mode(1)
    rnd = random.Random(8)
    for _ in range(25):
        n = rnd.randint(2, 40)
        a = [rnd.choice(b"abc") for _ in range(n)]
        b = list(a)
        for _k in range(rnd.randint(0, 2)):
            b[rnd.randrange(n)] = rnd.choice(b"abcd")
        ta, tb = [sk.trivial(v) for v in a], [sk.trivial(v) for v in b]
        cuts = sorted(rnd.sample(range(1, n), min(n - 1, rnd.randint(1, 4))))
        ranges = list(zip([0] + cuts, cuts + [n]))
        for cmp, py in ((0, bytes(a) < bytes(b)), (1, bytes(a) <= bytes(b)), (2, bytes(a) > bytes(b)), (3, bytes(a) >= bytes(b))):
            parts = [sk.compare_partial(FheString(ta[lo:hi]), FheString(tb[lo:hi]), cmp) for lo, hi in ranges]
            got = sk.flags_first_decides([p[0] for p in parts], [p[1] for p in parts], tie=cmp in (1, 3)).trivial_value()
            whole = sk._compare(FheString(ta), FheString(tb), cmp).trivial_value()
            assert got == whole == int(py), (a, b, ranges, cmp, got, whole)


def test_baseline_configs_at_full_size_by_folding(folded):
    """BASELINE configs 3-5 at their sizes (256 / 1024 / 4096 characters), as plaintext through the re-associated DAGs:
    the same DAGs the GPU tests run on ciphertexts (tests/test_gpu_fullsize.py), checked here against python."""
    from fhestring_amd.api import FheString
    sk = folded
    sk.set_mode(1)
    rnd = random.Random(2026)
    enc = lambda t, pad=1: FheString([sk.trivial(ord(c)) for c in t] + [sk.trivial(0)] * pad)
    dec = lambda s: ostr.truncate_plain([c.trivial_value() for c in s.chars])
    # config 4: replace on 1024 characters, from / to of 5, overlapping candidates, `to` with the pattern's prefix inside
    text = list(_rand_text(rnd, 1024, "abc"))
    for pos in (0, 7, 12, 500, 503, 1019):                 # 500 / 503 overlap: the earlier match wins, like str.replace
        text[pos:pos + 5] = "xyxyz" if pos != 503 else "yzxyz"
    text = "".join(text)
    for frm, to in (("xyxyz", "xyQRS"), ("xyxyz", "q"), ("yz", "yzyz")):
        assert dec(sk.replace(enc(text), enc(frm, 0), enc(to, 0))) == text.replace(frm, to), (frm, to)
    # config 5: eq_ignore_case and <= on 4096 characters, equal up to case / differing at one late position / a prefix
    a = _rand_text(rnd, 4096, "abcXYZ019 _")
    for b in (a.swapcase(), a[:4000] + ("b" if a[4000] != "b" else "c") + a[4001:], a[:4095], a):
        got = (sk.eq_ignore_case(enc(a), enc(b)).trivial_value(), sk.le(enc(a), enc(b)).trivial_value(),
               sk.ge(enc(a), enc(b, 3)).trivial_value(), sk.eq(enc(a), enc(b, 2)).trivial_value())
        assert got == (int(a.lower() == b.lower()), int(a <= b), int(a >= b), int(a == b)), (b[:8], got)
    # config 2 at several sizes: contains with clear and with encrypted patterns, hit in the last window and miss
    for n in (64, 65, 300):
        t = _rand_text(rnd, n, "ab")
        for p in (t[-4:], "abba" * 2, "c"):
            assert sk.contains(enc(t), enc(p, 0)).trivial_value() == int(p in t), (n, p)
            assert sk.contains_clear(enc(t), p).trivial_value() == int(p in t), (n, p)
    assert sk.len(enc("x" * 300)).trivial_value() == 300 % 256


@pytest.mark.parametrize("mode", [0, 1], ids=["as_written", "fused"])
def test_empty_strings_empty_patterns_and_zero_padding(folded, mode):
    """The grid of small edge cases (empty string, empty pattern, pattern longer than the string, no padding) through
    every method: same answers -- and the same panics -- as the clear-model restatement of the reference."""
    import itertools
    folded.set_mode(mode)
    penv, cenv = product_env(folded), clear_env()
    strings, pats = ["", "a", "ab", " a ", "aaa", "abab"], ["", "a", "ab", "aa", "abc", "abababab"]
    for s, p, pad in itertools.product(strings, pats, (0, 1, 2)):
        base = {"string": s, "pad": pad, "name": "edge"}
        vs = [dict(base, op=op, pattern=p) for op in ("contains", "starts_with", "ends_with", "find", "rfind", "strip_prefix", "strip_suffix")]
        vs += [dict(base, op=op) for op in ("to_upper", "trim", "trim_start", "trim_end", "is_empty", "len")]
        vs += [dict(base, op=op, other=p, other_pad=pad) for op in ("eq", "eq_ignore_case", "lt", "le", "gt", "ge", "concatenate")]
        vs += [dict(base, op="replace", **{"from": p, "to": t}) for t in ("", "x", "xyz")]
        vs += [dict(base, op="replacen", n=k, **{"from": p, "to": "x"}) for k in (0, 1, 2)]
        vs += [dict(base, op="repeat", n=k) for k in (0, 1, 2)]
        if p:
            vs += [dict(base, op=op, pattern=p) for op in ("split", "rsplit", "split_terminator", "split_inclusive", "rsplit_terminator")]
            vs += [dict(base, op=op, pattern=p, n=2) for op in ("splitn", "rsplitn")]
        for v in vs:
            def outcome(env):
                try:
                    return ("ok", run_vector(v, *env))
                except Exception as exc:                      # the reference panics here: the product raises the same kind
                    return ("raises", type(exc).__name__)
            _same(v, outcome(penv), outcome(cenv))


@pytest.mark.parametrize("mode", [0, 1], ids=["as_written", "fused"])
def test_character_operators_on_byte_pairs(folded, mode):
    """FheAsciiChar's operators (fheasciichar.rs:35-168) by constant folding: eq / ne / le / lt / ge / gt / bitand / bitor
    / add / sub / if_then_else on a grid of byte pairs (every b for every 8th a, every 5th b otherwise), the three class
    tests and flip on all bytes."""
    sk = folded
    sk.set_mode(mode)
    tr = [sk.trivial(v) for v in range(256)]
    for a in range(256):
        ca = tr[a]
        for b in range(0, 256, 1 if a % 8 == 0 else 5):
            cb = tr[b]
            got = tuple(x.trivial_value() for x in (ca.eq(cb), ca.ne(cb), ca.le(cb), ca.lt(cb), ca.ge(cb), ca.gt(cb), ca.bitand(cb),
                                                      ca.bitor(cb), ca.add(cb), ca.sub(cb), ca.if_then_else(cb, ca)))
            assert got == (int(a == b), int(a != b), int(a <= b), int(a < b), int(a >= b), int(a > b), a & b, a | b,
                           (a + b) & 255, (a - b) & 255, b if a else a), (a, b, got)
        got = (ca.is_whitespace().trivial_value(), ca.is_uppercase().trivial_value(), ca.is_lowercase().trivial_value())
        assert got == (int(a in (32, 9, 10, 11, 12, 13)), int(0x41 <= a <= 0x5A), int(0x61 <= a <= 0x7A)), (a, got)
    assert (tr[0].flip().trivial_value(), tr[1].flip().trivial_value()) == (1, 0)
