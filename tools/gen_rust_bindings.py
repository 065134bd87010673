#!/usr/bin/env python3
"""include/fhestring_hip.h -> bindings/fhestring_hip.rs: the COMPLETE Rust `extern "C"` binding of the drop-in boundary.

The reference is a Rust crate (src/ciphertext/fheasciichar.rs:17-168, src/client_key.rs:30-106 are the seam,
SURVEY.md section 8(b)); this image has no Rust toolchain, so the binding cannot be compiled here.  What can be done is
to DERIVE it mechanically from the header that `build()` compiles as C99 and that tests/test_cabi.py checks against
the exported symbols of libfhestring_hip.so, and to CHECK the derived file with an independent parser
(tests/test_rust_bindings.py): every function, its arity, every argument and return type, every constant, every struct
field.  bindgen conventions: `#[repr(C)]` opaque types, `pub type` aliases, raw pointers, std::os::raw integer types.

    python tools/gen_rust_bindings.py            # rewrites bindings/fhestring_hip.rs
    python tools/gen_rust_bindings.py --check    # exit 1 if the committed file is not what the header gives
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fhestring_hip.h")
OUT = os.path.join(ROOT, "bindings", "fhestring_hip.rs")

SCALARS = {"int": "c_int", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "int64_t": "i64",
           "int32_t": "i32", "double": "f64", "char": "c_char", "void": "c_void"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def parse_type(words, stars, known):
    """('const', 'uint64_t'), 1 -> '*const u64'; the pointee's constness decides *const / *mut (single level), `T **` is
    `*mut *mut T` (an out-parameter that receives a pointer)."""
    const = "const" in words
    base = [w for w in words if w not in ("const", "struct")]
    assert len(base) == 1, words
    name = base[0]
    rust = SCALARS.get(name) or (name if name in known else None)
    assert rust, "unknown C type %r" % name
    if stars == 0:
        assert rust != "c_void"
        return rust
    t = ("*const " if const else "*mut ") + rust
    for _ in range(stars - 1):
        t = "*mut " + t
    return t


def parse_param(p, known):
    p = p.strip()
    m = re.match(r"^(.*?)(\**)\s*([A-Za-z_]\w*)\s*((?:\[[^\]]*\])*)$", p.replace(" *", "*").replace("* ", "*"))
    assert m, p
    words = m.group(1).split()
    stars = len(m.group(2)) + (1 if m.group(4) else 0)       # `uint32_t key[8]` decays to a pointer
    return m.group(3), parse_type(words, stars, known)


def parse_header(path=HEADER):
    """-> dict(consts=[(name, int)], opaque=[name], aliases=[(name, rust)], fnptrs=[(name, ret, [(arg, type)])],
    structs=[(name, [(field, type)])], funcs=[(name, ret or None, [(arg, type)])]) in header order"""
    text = strip_comments(open(path).read())
    out = {"consts": [], "opaque": [], "aliases": [], "fnptrs": [], "structs": [], "funcs": []}
    env = {}
    for m in re.finditer(r"^#define\s+(FHS_\w+)\s+(.+)$", text, flags=re.M):
        expr = re.sub(r"\(\s*size_t\s*\)", "", m.group(2)).strip()
        assert re.fullmatch(r"[\w\s()*+\-]+", expr), expr
        val = eval(expr, {"__builtins__": {}}, dict(env))     # integer arithmetic over earlier FHS_ constants only
        env[m.group(1)] = val
        out["consts"].append((m.group(1), val))
    body = re.sub(r"^#.*$", "", text, flags=re.M)
    body = body.replace('extern "C" {', "")
    known = set()
    # struct typedefs with fields first (their names are types of later declarations)
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", body, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ty, names = decl.split(None, 1)
            for n in names.split(","):
                fields.append((n.strip(), SCALARS[ty]))
        out["structs"].append((m.group(2), fields))
        known.add(m.group(2))
    body = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", body, flags=re.S)
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", body):
        out["opaque"].append(m.group(2))
        known.add(m.group(2))
    for m in re.finditer(r"typedef\s+(\w+)\s+(\w+)\s*;", body):
        out["aliases"].append((m.group(2), SCALARS[m.group(1)]))
        known.add(m.group(2))
    for m in re.finditer(r"typedef\s+(\w+)\s*\(\s*\*\s*(\w+)\s*\)\s*\((.*?)\)\s*;", body, flags=re.S):
        args = [parse_param(p, known) for p in m.group(3).split(",")]
        out["fnptrs"].append((m.group(2), SCALARS[m.group(1)], args))
        known.add(m.group(2))
    body = re.sub(r"typedef[^;]*;", "", body)
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt or stmt == "}":
            continue
        m = re.match(r"^(.*?)(\**)\s*(fhs_\w+)\s*\((.*)\)$", stmt.replace(" *", "*"))
        assert m, stmt
        ret_words, ret_stars = m.group(1).split(), len(m.group(2))
        ret = None if (ret_words == ["void"] and ret_stars == 0) else parse_type(ret_words, ret_stars, known)
        params = m.group(4).strip()
        args = [] if params == "void" else [parse_param(p, known) for p in params.split(",")]
        out["funcs"].append((m.group(3), ret, args))
    return out


RUST_KEYWORDS = {"type", "fn", "in", "ref", "mod", "move", "loop", "match", "box", "use", "where", "impl", "self", "super"}


def ident(n):
    return "r#" + n if n in RUST_KEYWORDS else n


def render(h):
    L = ["// GENERATED by tools/gen_rust_bindings.py from include/fhestring_hip.h -- do not edit.",
         "// The complete C ABI of libfhestring_hip.so (%d functions): what a Rust host of MakisChristou/fhestring binds in"
         % len(h["funcs"]),
         "// place of the tfhe::integer calls of src/ciphertext/fheasciichar.rs:17-168 and src/client_key.rs:30-106.",
         "// Argument meaning, ownership and the reference line each entry point replaces: see the header's comments.",
         "// Checked against the header by tests/test_rust_bindings.py (names, arity, types, constants, struct fields).",
         "#![allow(non_camel_case_types, dead_code)]",
         "use std::os::raw::{c_char, c_int, c_void};", ""]
    for name, val in h["consts"]:
        ty = "usize" if name.endswith("_WORDS") or name.endswith("_BYTES") else "c_int"
        L.append("pub const %s: %s = %d;" % (name, ty, val))
    L.append("")
    for name in h["opaque"]:
        L += ["#[repr(C)]", "pub struct %s {" % name, "    _private: [u8; 0],", "}"]
    for name, rust in h["aliases"]:
        L.append("pub type %s = %s;" % (name, rust))
    for name, ret, args in h["fnptrs"]:
        L.append("pub type %s = Option<unsafe extern \"C\" fn(%s) -> %s>;"
                 % (name, ", ".join("%s: %s" % (ident(a), t) for a, t in args), ret))
    for name, fields in h["structs"]:
        L += ["#[repr(C)]", "#[derive(Debug, Default, Clone, Copy)]", "pub struct %s {" % name]
        L += ["    pub %s: %s," % (ident(f), t) for f, t in fields]
        L.append("}")
    L += ["", "#[link(name = \"fhestring_hip\")]", "extern \"C\" {"]
    for name, ret, args in h["funcs"]:
        sig = ", ".join("%s: %s" % (ident(a), t) for a, t in args)
        L.append("    pub fn %s(%s)%s;" % (name, sig, " -> " + ret if ret else ""))
    L += ["}", ""]
    return "\n".join(L)


def main():
    text = render(parse_header())
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            sys.stderr.write("bindings/fhestring_hip.rs is stale: run python tools/gen_rust_bindings.py\n")
            return 1
        return 0
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(text)
    h = parse_header()
    print("wrote %s: %d functions, %d constants, %d structs" % (OUT, len(h["funcs"]), len(h["consts"]), len(h["structs"])))
    return 0


if __name__ == "__main__":
    sys.exit(main())
