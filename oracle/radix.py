"""Oracle radix layer -- TEST INFRASTRUCTURE ONLY.

Restates the 13 `tfhe::integer::ServerKey` ops the reference calls from
src/ciphertext/fheasciichar.rs:23-102 on 4 blocks x PARAM_MESSAGE_2_CARRY_2
(SURVEY.md Appendix B).  tfhe 0.5.2 is not on disk, so the block decompositions
are this project's own (decrypt-level results are what the reference pins); the
product's C++ radix layer (fhestring_amd/csrc/radix.cpp) uses the SAME
decompositions and the SAME trivial-folding rules so that its ciphertexts can be
compared bit-for-bit with the ones produced here.

Two char models implement one interface:
  ClearChar  -- u8 arithmetic, pins oracle/strings.py to the golden vectors fast
  CipherChar -- 4 lazily evaluated LWE blocks, PBS through oracle/tfhe_oracle.c
"""
import numpy as np

from . import core

# --------------------------------------------------------------------------
# LUT catalogue: name -> f(v), v in [0,16).  Must match fhestring_amd/csrc/luts.h
# --------------------------------------------------------------------------
LUTS = {
    "eq_biv": lambda v: int((v >> 2) == (v & 3)),
    "ne_biv": lambda v: int((v >> 2) != (v & 3)),
    "and_biv": lambda v: (v >> 2) & (v & 3),
    "or_biv": lambda v: (v >> 2) | (v & 3),
    "is4": lambda v: int(v == 4),          # all four block-equalities hold
    "nz": lambda v: int(v != 0),
    "msg": lambda v: v & 3,
    "carry": lambda v: (v >> 2) & 3,
    "sign": lambda v: int(v != 0),          # negacyclic: x<0 -> -1, 0 -> 0, x>0 -> +1
    "cmp_lt": lambda v: int(_sel_sign(v) == 0),
    "cmp_le": lambda v: int(_sel_sign(v) in (0, 1)),
    "cmp_gt": lambda v: int(_sel_sign(v) == 2),
    "cmp_ge": lambda v: int(_sel_sign(v) in (1, 2)),
    "sel_t": lambda v: (v & 3) if (v >> 2) else 0,   # 4*cond + x -> cond ? x : 0
    "sel_f": lambda v: 0 if (v >> 2) else (v & 3),   # 4*cond + x -> cond ? 0 : x
}
for _c in range(4):  # comparisons against a clear block value
    LUTS["eq_c%d" % _c] = (lambda c: (lambda v: int(v == c)))(_c)
    LUTS["ne_c%d" % _c] = (lambda c: (lambda v: int(v != c)))(_c)


def _sel_sign(v):
    """v = 4*s_hi + s_lo with s in {0:lt,1:eq,2:gt}; most significant decides."""
    hi, lo = v >> 2, v & 3
    return hi if hi != 1 else lo


LUT_NAMES = sorted(LUTS)
_lut_cache = {}


def lut_poly(name):
    if name not in _lut_cache:
        _lut_cache[name] = core.make_lut(LUTS[name])
    return _lut_cache[name]


def lut_eval(name, v):
    """Plaintext semantics of a PBS incl. the negacyclic padding-bit rule."""
    v &= 31
    return LUTS[name](v) & 31 if v < 16 else (-LUTS[name](v - 16)) & 31


# --------------------------------------------------------------------------
# lazy blocks
# --------------------------------------------------------------------------
class Blk:
    """One shortint block: trivial constant, materialised ciphertext, or a
    pending PBS / linear combination."""
    __slots__ = ("kind", "val", "ct", "terms", "const", "src", "lut", "level")

    def __init__(self, kind):
        self.kind = kind      # 'triv' | 'ct' | 'lin' | 'pbs'
        self.val = None
        self.ct = None
        self.terms = None
        self.const = 0
        self.src = None
        self.lut = None
        self.level = 0


def triv(v):
    b = Blk("triv")
    b.val = int(v) & 31
    return b


def from_ct(ct):
    b = Blk("ct")
    b.ct = np.ascontiguousarray(ct, np.uint64)
    return b


class Engine:
    """Counts PBS and evaluates pending nodes level by level in thread batches."""

    def __init__(self, server_key, nthreads=None, mode=0):
        self.sk = server_key
        self.nthreads = nthreads
        self.mode = mode          # 0 exact NTT (parity oracle), 2 f64 FFT (the reference's algorithm class)
        self.pbs_count = 0
        self.levels = 0

    # -- node constructors --------------------------------------------------
    def lin(self, terms, const=0):
        """sum(c*b) + const*Delta.  Trivial terms fold into the constant."""
        k = int(const)
        flat = []
        for c, b in terms:
            c = int(c)
            if c == 0:
                continue
            if b.kind == "triv":
                k += c * b.val
            elif b.kind == "lin":
                k += c * b.const
                flat.extend((c * c2, b2) for c2, b2 in b.terms)
            else:
                flat.append((c, b))
        if not flat:
            return triv(k)
        n = Blk("lin")
        n.terms = flat
        n.const = k & 31
        n.level = max(b.level for _, b in flat)
        return n

    def pbs(self, x, lut):
        if x.kind == "triv":              # constant folding: no PBS executed
            return triv(lut_eval(lut, x.val))
        n = Blk("pbs")
        n.src = x
        n.lut = lut
        n.level = x.level + 1
        self.pbs_count += 1
        return n

    # -- evaluation ---------------------------------------------------------
    def _lin_value(self, b):
        acc = np.zeros(core.BIG_CT, np.uint64)
        for c, t in b.terms:
            acc += np.uint64(c & (2**64 - 1)) * t.ct
        acc[core.BIG_N:] += np.array([((b.const & 31) << core.DELTA_LOG) & (2**64 - 1)], np.uint64)
        return acc

    def materialize(self, blocks):
        """Make every block in `blocks` a 'ct' or 'triv'."""
        pend = {}

        def visit(b):
            if b.kind == "pbs" and id(b) not in pend:
                pend[id(b)] = b
                visit(b.src)
            elif b.kind == "lin":
                for _, t in b.terms:
                    visit(t)
        for b in blocks:
            visit(b)
        by_level = {}
        for b in pend.values():
            by_level.setdefault(b.level, []).append(b)
        for lv in sorted(by_level):
            batch = by_level[lv]
            ins = []
            for b in batch:
                s = b.src
                ins.append(self._lin_value(s) if s.kind == "lin" else s.ct)
            names = sorted({b.lut for b in batch})
            luts = np.stack([lut_poly(n) for n in names])
            idx = np.array([names.index(b.lut) for b in batch], np.uint32)
            outs = self.sk.pbs_batch(np.stack(ins), idx, luts, self.nthreads, self.mode)
            for b, o in zip(batch, outs):
                b.kind, b.ct, b.src, b.level = "ct", o, None, 0
            self.levels += 1
        for b in blocks:
            if b.kind == "lin":
                b.ct = self._lin_value(b)
                b.kind, b.terms, b.level = "ct", None, 0

    def block_ct(self, b):
        self.materialize([b])
        return core.trivial_block(b.val) if b.kind == "triv" else b.ct


# --------------------------------------------------------------------------
# char models
# --------------------------------------------------------------------------
class ClearChar:
    """u8 model of FheAsciiChar (fheasciichar.rs:8-168)."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = int(v) & 255

    @staticmethod
    def trivial(v, ctx=None):
        return ClearChar(v)

    def eq(self, o): return ClearChar(self.v == o.v)
    def ne(self, o): return ClearChar(self.v != o.v)
    def le(self, o): return ClearChar(self.v <= o.v)
    def lt(self, o): return ClearChar(self.v < o.v)
    def ge(self, o): return ClearChar(self.v >= o.v)
    def gt(self, o): return ClearChar(self.v > o.v)
    def bitand(self, o): return ClearChar(self.v & o.v)
    def bitor(self, o): return ClearChar(self.v | o.v)
    def add(self, o): return ClearChar(self.v + o.v)
    def sub(self, o): return ClearChar(self.v - o.v)
    def if_then_else(self, t, f): return ClearChar(t.v if self.v != 0 else f.v)
    def flip(self): return ClearChar(1 - self.v)
    def value(self): return self.v


class CipherChar:
    """4 lazily evaluated blocks, little endian 2-bit digits (fheasciichar.rs:8-10)."""
    __slots__ = ("b", "e")

    def __init__(self, blocks, engine):
        self.b = list(blocks)
        self.e = engine

    @staticmethod
    def trivial(v, engine):
        """create_trivial_radix (fheasciichar.rs:17-25)."""
        return CipherChar([triv((int(v) >> (2 * i)) & 3) for i in range(4)], engine)

    @staticmethod
    def from_cts(ct4, engine):
        return CipherChar([from_ct(ct4[i]) for i in range(4)], engine)

    def _flag(self, blk):
        """BooleanBlock::into_radix(4) (fheasciichar.rs:37): 3 trivial zero blocks."""
        return CipherChar([blk, triv(0), triv(0), triv(0)], self.e)

    # eq_parallelized / ne_parallelized (fheasciichar.rs:35-43): 4 bivariate + 1 reduce
    def _eq_blocks(self, o, kind):
        e = self.e
        outs = []
        for x, y in zip(self.b, o.b):
            if y.kind == "triv":
                outs.append(e.pbs(x, "%s_c%d" % (kind, y.val & 3)))
            elif x.kind == "triv":
                outs.append(e.pbs(y, "%s_c%d" % (kind, x.val & 3)))
            else:
                outs.append(e.pbs(e.lin([(4, x), (1, y)]), kind + "_biv"))
        return outs

    def eq(self, o):
        e = self.e
        return self._flag(e.pbs(e.lin([(1, b) for b in self._eq_blocks(o, "eq")]), "is4"))

    def ne(self, o):
        e = self.e
        return self._flag(e.pbs(e.lin([(1, b) for b in self._eq_blocks(o, "ne")]), "nz"))

    # lt/le/gt/ge_parallelized (fheasciichar.rs:45-63): pack pairs, 2 sign PBS, 1 combine
    def _cmp(self, o, lut):
        e = self.e
        signs = []
        for p in range(2):
            d = e.lin([(1, self.b[2 * p]), (4, self.b[2 * p + 1]),
                       (-1, o.b[2 * p]), (-4, o.b[2 * p + 1])])
            signs.append(e.lin([(1, e.pbs(d, "sign"))], 1))   # {0:lt, 1:eq, 2:gt}
        return self._flag(e.pbs(e.lin([(4, signs[1]), (1, signs[0])]), lut))

    def lt(self, o): return self._cmp(o, "cmp_lt")
    def le(self, o): return self._cmp(o, "cmp_le")
    def gt(self, o): return self._cmp(o, "cmp_gt")
    def ge(self, o): return self._cmp(o, "cmp_ge")

    # bitand/bitor_parallelized (fheasciichar.rs:65-81): 4 bivariate PBS
    def _bitop(self, o, lut):
        e = self.e
        return CipherChar([e.pbs(e.lin([(4, x), (1, y)]), lut) for x, y in zip(self.b, o.b)], e)

    def bitand(self, o): return self._bitop(o, "and_biv")
    def bitor(self, o): return self._bitop(o, "or_biv")

    # add/sub_parallelized (fheasciichar.rs:83-91): block sums + sequential carry chain
    def _addc(self, terms_per_block, carry_in):
        e = self.e
        out, carry = [], carry_in
        for i in range(4):
            s = e.lin(terms_per_block[i] + ([(1, carry)] if carry is not None else []))
            out.append(e.pbs(s, "msg"))
            carry = e.pbs(s, "carry") if i < 3 else None
        return CipherChar(out, e)

    def add(self, o):
        return self._addc([[(1, x), (1, y)] for x, y in zip(self.b, o.b)], None)

    def sub(self, o):
        # a - b = a + (3 - b_i per block) + 1  (mod 256)
        return self._addc([[(1, x), (-1, y), (1, triv(3))] for x, y in zip(self.b, o.b)], triv(1))

    def flip(self):
        """1 - x (fheasciichar.rs:161-168)."""
        return CipherChar.trivial(1, self.e).sub(self)

    # scalar_ne_parallelized(self,0) + if_then_else_parallelized (fheasciichar.rs:93-104)
    def if_then_else(self, t, f):
        e = self.e
        p0 = e.pbs(e.lin([(1, self.b[0]), (4, self.b[1])]), "nz")
        p1 = e.pbs(e.lin([(1, self.b[2]), (4, self.b[3])]), "nz")
        cond = e.pbs(e.lin([(1, p0), (1, p1)]), "nz")
        out = []
        for x, y in zip(t.b, f.b):
            tx = e.pbs(e.lin([(4, cond), (1, x)]), "sel_t")
            fy = e.pbs(e.lin([(4, cond), (1, y)]), "sel_f")
            out.append(e.lin([(1, tx), (1, fy)]))
        return CipherChar(out, e)

    def cts(self):
        """[4, 2049] ciphertext of this char (trivial blocks included)."""
        self.e.materialize(self.b)
        return np.stack([core.trivial_block(b.val) if b.kind == "triv" else b.ct for b in self.b])


def decrypt_char(keys, ch):
    if isinstance(ch, ClearChar):
        return ch.v
    return keys.decrypt_char(ch.cts())
