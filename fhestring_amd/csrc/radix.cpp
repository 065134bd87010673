#include "radix.h"

namespace fhs {

Ref lin(Engine *e, std::initializer_list<std::pair<int64_t, const Ref *>> terms, int konst) {
    Term t[16];
    size_t n = 0;
    for (const auto &p : terms) t[n++] = {p.first, p.second->id()};
    return Ref(e, e->lin(t, n, konst));
}
Ref pbs(const Ref &x, int lut) { return Ref(x.engine(), x.engine()->pbs(x.id(), lut)); }
Ref trivial_block(Engine *e, int v) { return Ref(e, e->triv(v)); }

FChar ch_trivial(Engine *e, uint8_t v) {
    FChar c;
    for (int i = 0; i < 4; i++) c.b[i] = trivial_block(e, (v >> (2 * i)) & 3);
    return c;
}
FChar ch_flag(Engine *e, const Ref &blk) {
    FChar c;
    c.b[0] = blk;
    for (int i = 1; i < 4; i++) c.b[i] = trivial_block(e, 0);
    return c;
}

// eq_parallelized / ne_parallelized: 4 block comparisons (bivariate, or univariate against a
// clear block), sum, 1 reduce
static Ref eq_blocks_sum(const FChar &a, const FChar &b, bool ne) {
    Engine *e = a.b[0].engine();
    Ref f[4];
    for (int i = 0; i < 4; i++) {
        const Ref &x = a.b[i], &y = b.b[i];
        if (e->is_triv(y.id())) f[i] = pbs(x, (ne ? LUT_NE_C0 : LUT_EQ_C0) + (e->triv_val(y.id()) & 3));
        else if (e->is_triv(x.id())) f[i] = pbs(y, (ne ? LUT_NE_C0 : LUT_EQ_C0) + (e->triv_val(x.id()) & 3));
        else f[i] = pbs(lin(e, {{4, &x}, {1, &y}}), ne ? LUT_NE_BIV : LUT_EQ_BIV);
    }
    return lin(e, {{1, &f[0]}, {1, &f[1]}, {1, &f[2]}, {1, &f[3]}});
}
Ref blk_eq_flag(const FChar &a, const FChar &b) { return pbs(eq_blocks_sum(a, b, false), LUT_IS4); }
Ref blk_ne_flag(const FChar &a, const FChar &b) { return pbs(eq_blocks_sum(a, b, true), LUT_NZ); }
FChar ch_eq(const FChar &a, const FChar &b) { return ch_flag(a.b[0].engine(), blk_eq_flag(a, b)); }
FChar ch_ne(const FChar &a, const FChar &b) { return ch_flag(a.b[0].engine(), blk_ne_flag(a, b)); }

// lt/le/gt/ge_parallelized: pack block pairs, subtract, sign PBS through the padding bit
// (negacyclic LUT: x<0 -> -1, 0 -> 0, x>0 -> +1), then one combining PBS
Ref blk_cmp_flag(const FChar &a, const FChar &b, int lut) {
    Engine *e = a.b[0].engine();
    Ref s[2];
    for (int p = 0; p < 2; p++) {
        Ref d = lin(e, {{1, &a.b[2 * p]}, {4, &a.b[2 * p + 1]}, {-1, &b.b[2 * p]}, {-4, &b.b[2 * p + 1]}});
        Ref sg = pbs(d, LUT_SIGN);
        s[p] = lin(e, {{1, &sg}}, 1);   // {0: lt, 1: eq, 2: gt}
    }
    return pbs(lin(e, {{4, &s[1]}, {1, &s[0]}}), lut);
}
FChar ch_lt(const FChar &a, const FChar &b) { return ch_flag(a.b[0].engine(), blk_cmp_flag(a, b, LUT_CMP_LT)); }
FChar ch_le(const FChar &a, const FChar &b) { return ch_flag(a.b[0].engine(), blk_cmp_flag(a, b, LUT_CMP_LE)); }
FChar ch_gt(const FChar &a, const FChar &b) { return ch_flag(a.b[0].engine(), blk_cmp_flag(a, b, LUT_CMP_GT)); }
FChar ch_ge(const FChar &a, const FChar &b) { return ch_flag(a.b[0].engine(), blk_cmp_flag(a, b, LUT_CMP_GE)); }

static FChar bitop(const FChar &a, const FChar &b, int lut) {
    Engine *e = a.b[0].engine();
    FChar r;
    for (int i = 0; i < 4; i++) r.b[i] = pbs(lin(e, {{4, &a.b[i]}, {1, &b.b[i]}}), lut);
    return r;
}
FChar ch_bitand(const FChar &a, const FChar &b) { return bitop(a, b, LUT_AND_BIV); }
FChar ch_bitor(const FChar &a, const FChar &b) { return bitop(a, b, LUT_OR_BIV); }

// add/sub_parallelized: block sums, then a sequential carry chain (message + carry PBS per block)
static FChar add_with_carry(const FChar &a, const FChar &b, bool subtract) {
    Engine *e = a.b[0].engine();
    FChar r;
    Ref carry = subtract ? trivial_block(e, 1) : Ref();
    for (int i = 0; i < 4; i++) {
        Ref s;
        if (subtract) s = lin(e, {{1, &a.b[i]}, {-1, &b.b[i]}, {1, &carry}}, 3);   // a + (3 - b) + c
        else if (carry) s = lin(e, {{1, &a.b[i]}, {1, &b.b[i]}, {1, &carry}});
        else s = lin(e, {{1, &a.b[i]}, {1, &b.b[i]}});
        r.b[i] = pbs(s, LUT_MSG);
        if (i < 3) carry = pbs(s, LUT_CARRY);
    }
    return r;
}
FChar ch_add(const FChar &a, const FChar &b) { return add_with_carry(a, b, false); }
FChar ch_sub(const FChar &a, const FChar &b) { return add_with_carry(a, b, true); }
FChar ch_flip(const FChar &a) { return ch_sub(ch_trivial(a.b[0].engine(), 1), a); }

// scalar_ne_parallelized(x, 0): pack pairs, 2 non-zero tests, 1 reduce
Ref blk_nonzero_flag(const FChar &a) {
    Engine *e = a.b[0].engine();
    Ref p0 = pbs(lin(e, {{1, &a.b[0]}, {4, &a.b[1]}}), LUT_NZ);
    Ref p1 = pbs(lin(e, {{1, &a.b[2]}, {4, &a.b[3]}}), LUT_NZ);
    return pbs(lin(e, {{1, &p0}, {1, &p1}}), LUT_NZ);
}
// if_then_else_parallelized: zero out each branch by cond / !cond, add
FChar ch_ite(const FChar &cond, const FChar &t, const FChar &f) {
    Engine *e = cond.b[0].engine();
    Ref c = blk_nonzero_flag(cond);
    FChar r;
    for (int i = 0; i < 4; i++) {
        Ref tx = pbs(lin(e, {{4, &c}, {1, &t.b[i]}}), LUT_SEL_T);
        Ref fy = pbs(lin(e, {{4, &c}, {1, &f.b[i]}}), LUT_SEL_F);
        r.b[i] = lin(e, {{1, &tx}, {1, &fy}});
    }
    return r;
}

FChar ch_is_whitespace(const FChar &a) {
    Engine *e = a.b[0].engine();
    static const uint8_t ws[6] = {0x20, 0x09, 0x0A, 0x0B, 0x0C, 0x0D};
    FChar r = ch_eq(a, ch_trivial(e, ws[0]));
    for (int i = 1; i < 6; i++) r = ch_bitor(r, ch_eq(a, ch_trivial(e, ws[i])));
    return r;
}
FChar ch_is_uppercase(const FChar &a) {
    Engine *e = a.b[0].engine();
    return ch_bitand(ch_ge(a, ch_trivial(e, 0x41)), ch_le(a, ch_trivial(e, 0x5A)));
}
FChar ch_is_lowercase(const FChar &a) {
    Engine *e = a.b[0].engine();
    return ch_bitand(ch_ge(a, ch_trivial(e, 0x61)), ch_le(a, ch_trivial(e, 0x7A)));
}

}  // namespace fhs
