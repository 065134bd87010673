"""bindings/fhestring_hip.rs is the complete Rust `extern "C"` binding of include/fhestring_hip.h.
Rust cannot be compiled in this image, so the file is derived mechanically (tools/gen_rust_bindings.py) and checked
here from three independent sides:

  1. the committed file is exactly what the generator makes of today's header (no drift);
  2. a second, token-based parser of the header written HERE (not the generator's regexes) and a parser of the Rust text
     agree on every function name, arity, argument / return type, constant value and struct field;
  3. the ABI class of every argument (pointer, 32-bit int, 64-bit int, size_t, double, byte) agrees with the ctypes
     declarations of fhestring_amd/_lib.py -- the binding the GPU tests actually call the library through.
"""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fhestring_hip.h")
RS = os.path.join(ROOT, "bindings", "fhestring_hip.rs")

C2R = {"int": "c_int", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "int64_t": "i64",
       "int32_t": "i32", "double": "f64", "char": "c_char", "void": "c_void", "fhs_char_t": "fhs_char_t",
       "fhs_ctx": "fhs_ctx", "fhs_client": "fhs_client", "fhs_stats": "fhs_stats", "fhs_capture_rec": "fhs_capture_rec",
       "fhs_allgather_fn": "fhs_allgather_fn"}


def _tokens(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    return re.findall(r"[A-Za-z_]\w*|\d+|[(){}\[\];,*]|\"C\"", text)


def _c_type(tok):
    """tokens of one declarator without its name -> Rust spelling"""
    const = "const" in tok
    stars = tok.count("*")
    base = [t for t in tok if t not in ("const", "*", "struct")]
    assert len(base) == 1, tok
    r = C2R[base[0]]
    if stars == 0:
        return r
    return "*mut " * (stars - 1) + ("*const " if const else "*mut ") + r


def header_functions():
    """{name: (ret, [types])} by walking the token stream: `<type tokens> fhs_name ( params ) ;` at brace depth <= 1"""
    toks = _tokens(open(HEADER).read())
    out, i, start = {}, 0, 0
    depth = 0
    while i < len(toks):
        t = toks[i]
        if t == "typedef":                       # skip to the end of the typedef (struct bodies included)
            d = 0
            while not (toks[i] == ";" and d == 0):
                d += toks[i] == "{"
                d -= toks[i] == "}"
                i += 1
            start = i + 1
        elif t in ("{", "}"):
            start = i + 1
        elif t == ";":
            start = i + 1
        elif t.startswith("fhs_") and i + 1 < len(toks) and toks[i + 1] == "(" and t not in C2R:
            ret = toks[start:i]
            j, params, cur = i + 2, [], []
            while toks[j] != ")":
                if toks[j] == ",":
                    params.append(cur); cur = []
                else:
                    cur.append(toks[j])
                j += 1
            params.append(cur)
            types = []
            if params != [["void"]]:
                for p in params:
                    if "[" in p:                 # `uint32_t key[8]`: decays to a pointer
                        p = p[:p.index("[")] + ["*"]
                        name_at = max(k for k, x in enumerate(p) if re.match(r"[A-Za-z_]", x) and x not in C2R and x != "const")
                        p = p[:name_at] + p[name_at + 1:]
                    else:
                        p = p[:-1]               # drop the parameter name
                    types.append(_c_type(p))
            r = None if ret == ["void"] else _c_type(ret)
            out[t] = (r, types)
            i = j
        i += 1
    return out


def rust_functions():
    text = open(RS).read()
    block = text[text.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)(?: -> ([^;]+))?;", block):
        args = [a.split(":", 1)[1].strip() for a in m.group(2).split(", ")] if m.group(2).strip() else []
        out[m.group(1)] = (m.group(3).strip() if m.group(3) else None, args)
    return out


def test_committed_binding_is_what_the_generator_makes_of_the_header():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_bindings.py"), "--check"],
                       capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr


def test_every_function_matches_the_header_in_name_arity_and_types():
    h, r = header_functions(), rust_functions()
    assert len(h) >= 130 and sorted(h) == sorted(r)
    bad = {n: (h[n], r[n]) for n in h if h[n] != r[n]}
    assert not bad, bad
    # spot checks of what a shim for src/ciphertext/fheasciichar.rs:35-104 and src/client_key.rs:81-87 calls
    assert r["fhs_eq"] == ("fhs_char_t", ["*mut fhs_ctx", "fhs_char_t", "fhs_char_t"])
    assert r["fhs_if_then_else"] == ("fhs_char_t", ["*mut fhs_ctx"] + ["fhs_char_t"] * 3)
    assert r["fhs_ctx_create"] == ("c_int", ["c_int", "*mut *mut fhs_ctx"])
    assert r["fhs_client_decrypt_char"] == ("c_int", ["*const fhs_client", "*const u64", "*mut u8"])
    assert r["fhs_last_error"] == ("*const c_char", ["*const fhs_ctx"])
    assert r["fhs_ctx_destroy"] == (None, ["*mut fhs_ctx"])


def test_constants_and_structs_match_the_header():
    text = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    rs = open(RS).read()
    consts = dict(re.findall(r"pub const (FHS_\w+): \w+ = (-?\d+);", rs))
    defines = re.findall(r"^#define\s+(FHS_\w+)\s+(.+)$", text, flags=re.M)
    assert len(defines) == len(consts) >= 25
    env = {}
    for name, expr in defines:                   # evaluate with C's own arithmetic through the preprocessor-free subset
        val = eval(re.sub(r"\(\s*size_t\s*\)", "", expr), {"__builtins__": {}}, dict(env))
        env[name] = val
        assert int(consts[name]) == val, name
    assert int(consts["FHS_CHAR_WORDS"]) == 4 * 2049 and int(consts["FHS_ERR_LIMIT"]) == -4
    for sname in ("fhs_stats", "fhs_capture_rec"):
        body = re.search(r"typedef\s+struct\s*\{([^{}]*)\}\s*%s\s*;" % sname, text, flags=re.S).group(1)
        c_fields = []
        for decl in body.split(";"):
            if decl.strip():
                ty, names = decl.split(None, 1)
                c_fields += [(n.strip(), C2R[ty]) for n in names.split(",")]
        r_body = re.search(r"pub struct %s \{(.*?)\}" % sname, rs, flags=re.S).group(1)
        r_fields = re.findall(r"pub (\w+): (\w+),", r_body)
        assert r_fields == c_fields, sname
        assert re.search(r"#\[repr\(C\)\]\n#\[derive\([^)]*\)\]\npub struct %s \{" % sname, rs)
    for opaque in ("fhs_ctx", "fhs_client"):
        assert re.search(r"#\[repr\(C\)\]\npub struct %s \{\n    _private: \[u8; 0\],\n\}" % opaque, rs)
    assert "pub type fhs_char_t = u64;" in rs


def _abi_class_rust(t):
    if t.startswith("*") or t == "fhs_allgather_fn":
        return "ptr"
    return {"c_int": "i32", "i32": "i32", "u32": "i32", "usize": "i64", "u64": "i64", "i64": "i64", "fhs_char_t": "i64",
            "u8": "i8", "f64": "f64"}[t]


def _abi_class_ctypes(t):
    if t is None:
        return None
    if t in (ctypes.c_void_p, ctypes.c_char_p) or issubclass(t, (ctypes._Pointer, ctypes._CFuncPtr)):
        return "ptr"
    size = ctypes.sizeof(t)
    if t is ctypes.c_double:
        return "f64"
    return {1: "i8", 4: "i32", 8: "i64"}[size]


def test_abi_classes_agree_with_the_ctypes_binding_the_gpu_tests_use():
    import fhestring_amd
    L = fhestring_amd.lib()
    r = rust_functions()
    checked = 0
    for name, (ret, args) in r.items():
        f = getattr(L, name)
        if f.argtypes is None:
            continue                              # not declared in _lib.py (diagnostics called with explicit casts)
        got = [_abi_class_ctypes(t) for t in f.argtypes]
        assert got == [_abi_class_rust(a) for a in args], (name, got, args)
        if ret is None:
            assert f.restype is None, name
        else:
            assert _abi_class_ctypes(f.restype) == _abi_class_rust(ret), (name, f.restype, ret)
        checked += 1
    assert checked >= 100, checked
