"""CLI diff harness: counterpart of the reference binary's `main` + `run_fhe_str_method`
(src/main.rs:34-116, src/utils.rs:114-718) for all 52 `StringMethod` variants.

    python -m fhestring_amd.cli --string "hello" --pattern "ello" --n 1 --from "ello" --to "_llo"

Encrypts the inputs with the client key, runs each MyServerKey method on the MI355X, decrypts,
compares with the plaintext semantics (python `str` here, Rust `std::str` there) and prints the
reference's lines: `Test Passed: OK, Result: ..., ` / `Test Failed: Expected: ..., Got: ..., `
(utils.rs:114-120) followed by `<Method> <duration>` (main.rs:114).  As in the reference the timed
region includes encryption and decryption (utils.rs:135-145).
"""
import argparse
import sys
import time

from .api import MAX_FIND_LENGTH, MAX_REPETITIONS, STRING_PADDING, FheSplit, MyClientKey, trim_vector

WS = " \t\n\x0b\x0c\r"


def rust_debug(v):
    """`{:?}` of a u8 / String / &str / Vec<String>."""
    if isinstance(v, (list, tuple)):
        return "[" + ", ".join(rust_debug(x) for x in v) + "]"
    if isinstance(v, str):
        out = '"'
        for ch in v:
            if ch == '"':
                out += '\\"'
            elif ch == "\\":
                out += "\\\\"
            elif ch == "\n":
                out += "\\n"
            elif ch == "\t":
                out += "\\t"
            elif ch == "\r":
                out += "\\r"
            elif ord(ch) < 0x20 or ord(ch) == 0x7F:
                out += "\\u{%x}" % ord(ch)
            else:
                out += ch
        return out + '"'
    return str(int(v))


def compare_and_print(expected, actual, out=None):                # utils.rs:114-120
    out = sys.stdout if out is None else out          # looked up at call time: a default bound at import misses redirection
    if expected == actual:
        out.write("Test Passed: OK, Result: %s, " % rust_debug(actual))
        return True
    out.write("Test Failed: Expected: %s, Got: %s, " % (rust_debug(expected), rust_debug(actual)))
    return False


def rust_duration(sec):
    """`{:?}` of a std::time::Duration."""
    if sec >= 1:
        return ("%.9f" % sec).rstrip("0").rstrip(".") + "s"
    if sec >= 1e-3:
        return ("%.6f" % (sec * 1e3)).rstrip("0").rstrip(".") + "ms"
    return ("%.3f" % (sec * 1e6)).rstrip("0").rstrip(".") + "µs"


METHODS = [  # order of src/main.rs:47-100
    "Contains", "ContainsClear", "EndsWith", "EndsWithClear", "EqIgnoreCase", "Find", "FindClear", "IsEmpty",
    "Len", "Repeat", "RepeatClear", "Replace", "ReplaceClear", "ReplaceN", "ReplaceNClear", "Rfind",
    "RfindClear", "Rsplit", "RsplitClear", "RsplitOnce", "RsplitOnceClear", "RsplitN", "RsplitNClear",
    "RsplitTerminator", "RsplitTerminatorClear", "Split", "SplitClear", "SplitAsciiWhitespace", "SplitInclusive",
    "SplitInclusiveClear", "SplitTerminator", "SplitTerminatorClear", "SplitN", "SplitNClear", "StartsWith",
    "StartsWithClear", "StripPrefix", "StripPrefixClear", "StripSuffix", "StripSuffixClear", "ToLower",
    "ToUpper", "Trim", "TrimEnd", "TrimStart", "Concatenate", "Lt", "Le", "Gt", "Ge", "Eq", "Ne",
]
SKIPPED = []


def _py_split(method, s, p, n):
    """Rust std semantics of the split family (the `expected` side of src/utils.rs:321-521)."""
    base = method[:-5] if method.endswith("Clear") else method
    if base == "Split": return s.split(p)
    if base == "SplitInclusive":
        parts = s.split(p)
        return [x + p for x in parts[:-1]] + ([parts[-1]] if parts[-1] != "" else [])
    if base == "SplitTerminator":
        parts = s.split(p)
        return parts[:-1] if parts and parts[-1] == "" else parts
    if base == "SplitAsciiWhitespace": return s.split()
    if base == "SplitN": return s.split(p, n - 1) if n > 0 else []
    if base == "Rsplit": return s.split(p)[::-1]
    if base == "RsplitN": return s.rsplit(p, n - 1)[::-1] if n > 0 else []
    if base == "RsplitTerminator":
        parts = s.split(p)
        return (parts[:-1] if parts and parts[-1] == "" else parts)[::-1]
    raise KeyError(method)


def run_fhe_str_method(sk, ck, a, method, out=None):             # utils.rs:122-718
    out = sys.stdout if out is None else out
    s_plain, p_plain, f_plain, t_plain, n_plain = a.string, a.pattern, a.frm, a.to, a.n
    s = ck.encrypt(s_plain, STRING_PADDING, None, sk)             # utils.rs:135-145
    pat = ck.encrypt_no_padding(p_plain, sk)
    frm = ck.encrypt_no_padding(f_plain, sk)
    to = ck.encrypt_no_padding(t_plain, sk)
    n = ck.encrypt_char(n_plain & 255, sk)
    dch, dst = ck.decrypt_char, ck.decrypt
    other = lambda: ck.encrypt(p_plain, STRING_PADDING, None, sk)
    clear = lambda text: [sk.trivial(b) for b in text.encode("ascii")]
    pos = lambda v: v if v >= 0 else MAX_FIND_LENGTH
    ok = True
    if method in ("Contains", "ContainsClear"):
        r = sk.contains(s, pat) if method == "Contains" else sk.contains_clear(s, p_plain)
        ok = compare_and_print(int(p_plain in s_plain), dch(r), out)
    elif method in ("EndsWith", "EndsWithClear"):
        r = sk.ends_with(s, pat if method == "EndsWith" else clear(p_plain))
        ok = compare_and_print(int(s_plain.endswith(p_plain)), dch(r), out)
    elif method in ("StartsWith", "StartsWithClear"):
        r = sk.starts_with(s, pat if method == "StartsWith" else clear(p_plain))
        ok = compare_and_print(int(s_plain.startswith(p_plain)), dch(r), out)
    elif method == "EqIgnoreCase":
        ok = compare_and_print(int(s_plain.lower() == p_plain.lower()), dch(sk.eq_ignore_case(s, other())), out)
    elif method in ("Find", "FindClear"):
        r = sk.find(s, pat) if method == "Find" else sk.find_clear(s, p_plain)
        ok = compare_and_print(pos(s_plain.find(p_plain)) & 255, dch(r), out)
    elif method in ("Rfind", "RfindClear"):
        r = sk.rfind(s, pat if method == "Rfind" else clear(p_plain))
        ok = compare_and_print(pos(s_plain.rfind(p_plain)) & 255, dch(r), out)
    elif method == "IsEmpty":
        ok = compare_and_print(int(s_plain == ""), dch(sk.is_empty(s)), out)
    elif method == "Len":
        ok = compare_and_print(len(s_plain) & 255, dch(sk.len(s)), out)
    elif method == "Repeat":
        ok = compare_and_print(s_plain * n_plain, dst(sk.repeat(s, n)), out)
    elif method == "RepeatClear":
        ok = compare_and_print(s_plain * n_plain, dst(sk.repeat_clear(s, n_plain)), out)
    elif method in ("Replace", "ReplaceClear"):
        r = sk.replace(s, frm, to) if method == "Replace" else sk.replace_clear(s, f_plain, t_plain)
        ok = compare_and_print(s_plain.replace(f_plain, t_plain), dst(r), out)
    elif method in ("ReplaceN", "ReplaceNClear"):
        r = sk.replacen(s, frm, to, n) if method == "ReplaceN" else \
            sk.replacen(s, clear(f_plain), clear(t_plain), sk.trivial(n_plain & 255))
        ok = compare_and_print(s_plain.replace(f_plain, t_plain, n_plain), dst(r), out)
    elif method in ("StripPrefix", "StripPrefixClear", "StripSuffix", "StripSuffixClear"):
        prefix = method.startswith("StripPrefix")
        p = pat if not method.endswith("Clear") else clear(p_plain)
        res, found = (sk.strip_prefix if prefix else sk.strip_suffix)(s, p)
        hit = s_plain.startswith(p_plain) if prefix else s_plain.endswith(p_plain)
        if hit:                                                   # utils.rs:542-550: two comparisons
            exp = s_plain[len(p_plain):] if prefix else s_plain[:len(s_plain) - len(p_plain)]
            ok = compare_and_print(exp, dst(res), out)
        ok = compare_and_print(int(hit), dch(found), out) and ok
    elif method == "ToLower":
        ok = compare_and_print(s_plain.lower(), dst(sk.to_lower(s)), out)
    elif method == "ToUpper":
        ok = compare_and_print(s_plain.upper(), dst(sk.to_upper(s)), out)
    elif method == "Trim":
        ok = compare_and_print(s_plain.strip(WS), dst(sk.trim(s)), out)
    elif method == "TrimEnd":
        ok = compare_and_print(s_plain.rstrip(WS), dst(sk.trim_end(s)), out)
    elif method == "TrimStart":
        ok = compare_and_print(s_plain.lstrip(WS), dst(sk.trim_start(s)), out)
    elif method == "Concatenate":
        ok = compare_and_print(s_plain + p_plain, dst(sk.concatenate(s, other())), out)
    elif "plit" in method:                                         # split family, utils.rs:321-521
        clear = method.endswith("Clear")
        base = method[:-5] if clear else method
        pp = p_plain if clear else pat
        if base == "SplitAsciiWhitespace":
            r = sk.split_ascii_whitespace(s)
        elif base in ("SplitN", "RsplitN"):
            f = {"SplitN": (sk.splitn, sk.splitn_clear), "RsplitN": (sk.rsplitn, sk.rsplitn_clear)}[base][clear]
            r = f(s, pp, n_plain if clear else n)
        else:
            name = {"Split": "split", "SplitInclusive": "split_inclusive", "SplitTerminator": "split_terminator",
                    "Rsplit": "rsplit", "RsplitTerminator": "rsplit_terminator", "RsplitOnce": "rsplit_once"}[base]
            r = getattr(sk, name + ("_clear" if clear else ""))(s, pp)
        bufs, found = FheSplit.decrypt(r, ck)
        if base == "RsplitOnce":                                   # utils.rs:345-385
            if p_plain in s_plain:
                a, b = s_plain.rsplit(p_plain, 1)
                ok = compare_and_print(trim_vector([b, a]), trim_vector(bufs), out)
            else:
                ok = compare_and_print(0, found, out)
        else:
            ok = compare_and_print(trim_vector(_py_split(method, s_plain, p_plain, n_plain)), trim_vector(bufs), out)
    elif method in ("Lt", "Le", "Gt", "Ge", "Eq", "Ne"):
        f = {"Lt": sk.lt, "Le": sk.le, "Gt": sk.gt, "Ge": sk.ge, "Eq": sk.eq, "Ne": sk.ne}[method]
        e = {"Lt": s_plain < p_plain, "Le": s_plain <= p_plain, "Gt": s_plain > p_plain,
             "Ge": s_plain >= p_plain, "Eq": s_plain == p_plain, "Ne": s_plain != p_plain}[method]
        ok = compare_and_print(int(e), dch(f(s, other())), out)
    else:
        raise KeyError(method)
    return ok


def main(argv=None):
    ap = argparse.ArgumentParser(prog="fhestring_amd.cli", description=__doc__.split("\n")[0])
    ap.add_argument("--string", required=True)                    # src/args.rs:4-26: all required
    ap.add_argument("--pattern", required=True)
    ap.add_argument("--n", type=int, required=True)
    ap.add_argument("--from", dest="frm", required=True)
    ap.add_argument("--to", required=True)
    ap.add_argument("--mode", choices=["fused", "as_written"], default="fused")
    ap.add_argument("--arith", choices=["exact", "fft", "fft_mb2", "exact_mb2"], default="exact",
                    help="arithmetic of blind rotation (fhs_set_arithmetic): exact two-prime NTT, f64 FFT, or either with "
                         "two key bits per external product (needs the pair key, generated by the client key)")
    ap.add_argument("--methods", default="", help="comma-separated subset (default: all non-split methods)")
    ap.add_argument("--seed", type=int, default=None,
                    help="reproducible (insecure) keys for tests; default: OS entropy, like the reference")
    a = ap.parse_args(argv)
    assert a.n <= MAX_REPETITIONS, "n must be <= MAX_REPETITIONS"  # src/main.rs:37-40
    ck = MyClientKey.from_params(seed=a.seed)                     # src/main.rs:43
    sk = ck.get_server_key(0, arith={"exact": 0, "fft": 1, "fft_mb2": 2, "exact_mb2": 3}[a.arith])
    sk.set_mode(1 if a.mode == "fused" else 0)
    methods = [m for m in a.methods.split(",") if m] or METHODS
    failed = 0
    for m in methods:
        t0 = time.perf_counter()
        try:
            if not run_fhe_str_method(sk, ck, a, m):
                failed += 1
        except OverflowError as e:                                # the reference panics here
            sys.stdout.write("panicked: %s, " % e)
        print("%s %s" % (m, rust_duration(time.perf_counter() - t0)))
    sk.close()
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
