// LUT catalogue of the radix layer: name -> f(v), v in [0,16).  Same functions as the oracle's
// table (oracle/radix.py LUTS) so that both sides bootstrap with identical polynomials.
#pragma once
#include <cstdint>
#include <vector>

namespace fhs {

enum LutId : uint16_t {
    LUT_EQ_BIV = 0, LUT_NE_BIV, LUT_AND_BIV, LUT_OR_BIV, LUT_IS4, LUT_NZ, LUT_MSG, LUT_CARRY, LUT_SIGN,
    LUT_CMP_LT, LUT_CMP_LE, LUT_CMP_GT, LUT_CMP_GE, LUT_SEL_T, LUT_SEL_F,
    LUT_EQ_C0, LUT_EQ_C1, LUT_EQ_C2, LUT_EQ_C3, LUT_NE_C0, LUT_NE_C1, LUT_NE_C2, LUT_NE_C3,
    // fused-mode helpers (single-block flags, wide fan-in sums)
    LUT_IS1, LUT_IS2, LUT_IS3, LUT_IS5, LUT_IS6, LUT_IS7, LUT_IS8, LUT_IS9, LUT_IS10, LUT_IS11, LUT_IS12,
    LUT_IS13, LUT_IS14, LUT_IS15,
    LUT_IS0,           // v == 0
    LUT_HI2,           // v >> 2 (same as carry, kept for readability)
    LUT_SEL_T_FLAG,    // 4*cond + x -> cond ? x : 0 with x a flag (same table as SEL_T)
    LUT_MUX2,          // 2*cond + x (x in {0,1}) helper: v -> (v>>1) ? (v&1) : 0
    LUT_LE10,          // v <= 10 (low nibble of 'P'..'Z' / 'p'..'z')
    LUT_CASEFLAG,      // v = (h_a + lo_nz) + 4*(h_b + lo_le10) -> letter-of-that-case flag
    LUT_BIT0_UNLESS,   // v = digit + 4*mask -> mask ? 0 : digit & 1
    LUT_BIT1_UNLESS,   // v = digit + 4*mask -> mask ? 0 : (digit >> 1) & 1
    LUT_LO_WS0,        // low nibble of NUL or of an ASCII whitespace 0x09..0x0D: v in {0, 9..13}
    // greedy non-overlapping match selection, v = 2 * blocked_countdown + match_flag (countdown < 8): the weights
    // (2, 1) instead of (1, 8) keep the sum of squared coefficients at 5 (noise budget, fhestring_hip.h)
    LUT_GREEDY_SEL,    // 1 iff countdown == 0 and match
    LUT_GREEDY_NEXT0,  // next countdown for a pattern of length k + 1: selected ? k : max(countdown - 1, 0)
    LUT_GREEDY_NEXT1, LUT_GREEDY_NEXT2, LUT_GREEDY_NEXT3, LUT_GREEDY_NEXT4, LUT_GREEDY_NEXT5, LUT_GREEDY_NEXT6,
    LUT_GREEDY_NEXT7,
    // case-insensitive equality of two characters without folding either (Strings::f_eq_ignore_case)
    LUT_EQIC_B3,       // v = a3 + 4 b3 (top digits): 2 both are 1 (0x40..0x7F), 1 equal otherwise, 0 different
    LUT_EQIC_Z,        // v = a2 + 4 b2 (bits 4, 5): 1 equal, 2 + row bit if they differ in bit 5 (the case bit) only, 0 otherwise
    LUT_EQIC_LO,       // low nibble v: (v >= 1) + 2 (v <= 10): letter rows 0x41..0x4F / 0x50..0x5A (and 0x6_, 0x7_)
    LUT_EQIC_S1,       // v = Z + 4 LO: 1 same digit, 2 case bit differs and the low nibble is a letter's on that row, 0 otherwise
    LUT_EQIC_FIN,      // v = B3 + 3 S1 + 7 E_lo: low nibbles equal and (S1 == 1 with equal top digits, or S1 == 2 on a letter row)
    // character classes in three bootstraps (is_upper_flag, char_significant): a row class of the high nibble (1 / 2 / 0), a
    // two-flag class of the low nibble, and the pick of the flag that belongs to the row
    LUT_HI_ROW45,      // high nibble 4 -> 1, 5 -> 2, else 0 ('A'..'O' / 'P'..'Z')
    LUT_HI_ROW67,      // high nibble 6 -> 1, 7 -> 2, else 0 ('a'..'o' / 'p'..'z')
    LUT_HI_ROW02,      // high nibble 0 -> 1, 2 -> 2, else 0 (NUL, 0x09..0x0D / space)
    LUT_LO_WSCLS,      // low nibble: (v == 0 or 9 <= v <= 13) + 2 (v == 0)
    LUT_CLS_PICK,      // v = row + 4 flags: row 1 -> flags & 1, row 2 -> flags >> 1, row 0 -> 0
    // root of the three-state comparison tree (Strings::cmp_verdict): v = 8 + 4 s1 + 2 s2 + s3, s in {-1, 0, 1}
    LUT_LT8, LUT_LE8, LUT_GT8, LUT_GE8,
    LUT_COUNT
};

inline int sel_sign(int v) {   // v = 4*s_hi + s_lo, s in {0:lt, 1:eq, 2:gt}; most significant decides
    const int hi = v >> 2, lo = v & 3;
    return hi != 1 ? hi : lo;
}

// plaintext function of a LUT on v in [0,16); result is a block value (mod 32)
inline int lut_function(int id, int v) {
    switch (id) {
        case LUT_EQ_BIV: return (v >> 2) == (v & 3);
        case LUT_NE_BIV: return (v >> 2) != (v & 3);
        case LUT_AND_BIV: return (v >> 2) & (v & 3);
        case LUT_OR_BIV: return (v >> 2) | (v & 3);
        case LUT_IS4: return v == 4;
        case LUT_NZ: return v != 0;
        case LUT_MSG: return v & 3;
        case LUT_CARRY: return (v >> 2) & 3;
        case LUT_HI2: return (v >> 2) & 3;
        case LUT_SIGN: return v != 0;
        case LUT_CMP_LT: return sel_sign(v) == 0;
        case LUT_CMP_LE: return sel_sign(v) == 0 || sel_sign(v) == 1;
        case LUT_CMP_GT: return sel_sign(v) == 2;
        case LUT_CMP_GE: return sel_sign(v) == 1 || sel_sign(v) == 2;
        case LUT_SEL_T: case LUT_SEL_T_FLAG: return (v >> 2) ? (v & 3) : 0;
        case LUT_SEL_F: return (v >> 2) ? 0 : (v & 3);
        case LUT_IS0: return v == 0;
        case LUT_MUX2: return (v >> 1) ? (v & 1) : 0;
        case LUT_LE10: return v <= 10;
        case LUT_CASEFLAG: return ((v & 3) == 2) || ((v >> 2) == 2);
        case LUT_BIT0_UNLESS: return (v >> 2) ? 0 : (v & 1);
        case LUT_BIT1_UNLESS: return (v >> 2) ? 0 : ((v >> 1) & 1);
        case LUT_LO_WS0: return v == 0 || (v >= 9 && v <= 13);
        case LUT_GREEDY_SEL: return v == 1;
        case LUT_EQIC_B3: return (v & 3) != (v >> 2) ? 0 : ((v & 3) == 1 ? 2 : 1);
        case LUT_EQIC_Z: return (v & 3) == (v >> 2) ? 1 : ((((v & 3) ^ (v >> 2)) == 2) ? 2 + (v & 1) : 0);
        case LUT_EQIC_LO: return (v >= 1) + 2 * (v <= 10);
        case LUT_EQIC_S1: {
            const int z = v & 3, lo = v >> 2;
            return z <= 1 ? z : (((z == 2 ? lo : lo >> 1) & 1) ? 2 : 0);
        }
        case LUT_EQIC_FIN: return v == 11 || v == 12 || v == 15;    // (B3, S1, E_lo) = (1, 1, 1), (2, 1, 1), (2, 2, 1)
        case LUT_HI_ROW45: return v == 4 ? 1 : (v == 5 ? 2 : 0);
        case LUT_HI_ROW67: return v == 6 ? 1 : (v == 7 ? 2 : 0);
        case LUT_HI_ROW02: return v == 0 ? 1 : (v == 2 ? 2 : 0);
        case LUT_LO_WSCLS: return (v == 0 || (v >= 9 && v <= 13)) + 2 * (v == 0);
        case LUT_CLS_PICK: return (v & 3) == 1 ? ((v >> 2) & 1) : ((v & 3) == 2 ? ((v >> 3) & 1) : 0);
        case LUT_LT8: return v < 8;
        case LUT_LE8: return v <= 8;
        case LUT_GT8: return v > 8;
        case LUT_GE8: return v >= 8;
        default: break;
    }
    if (id >= LUT_GREEDY_NEXT0 && id <= LUT_GREEDY_NEXT7)
        return v == 1 ? id - LUT_GREEDY_NEXT0 : ((v >> 1) ? (v >> 1) - 1 : 0);
    if (id >= LUT_EQ_C0 && id <= LUT_EQ_C3) return v == id - LUT_EQ_C0;
    if (id >= LUT_NE_C0 && id <= LUT_NE_C3) return v != id - LUT_NE_C0;
    if (id >= LUT_IS1 && id <= LUT_IS3) return v == 1 + (id - LUT_IS1);
    if (id >= LUT_IS5 && id <= LUT_IS15) return v == 5 + (id - LUT_IS5);
    return 0;
}
inline int lut_is_k(int k) {   // LUT id of v -> (v == k), k in [0,15]
    if (k == 0) return LUT_IS0;
    if (k == 4) return LUT_IS4;
    if (k <= 3) return LUT_IS1 + (k - 1);
    return LUT_IS5 + (k - 5);
}

// value of a PBS on a plaintext v in [0,32), padding-bit (negacyclic) rule included
inline int lut_eval(int id, int v) {
    v &= 31;
    return v < 16 ? (lut_function(id, v) & 31) : ((-lut_function(id, v - 16)) & 31);
}

// LUT body polynomial (SURVEY.md Appendix A "LUT generation"): 16 boxes of 128, rotated by half a box
inline void make_lut_poly(int id, uint64_t *lut /*[2048]*/) {
    const int N = 2048, box = N / 16, half = box / 2;
    std::vector<uint64_t> tmp(N);
    for (int x = 0; x < 16; x++)
        for (int t = 0; t < box; t++) tmp[x * box + t] = (uint64_t)(lut_function(id, x) & 31) << 59;
    for (int i = 0; i < N - half; i++) lut[i] = tmp[i + half];
    for (int t = 0; t < half; t++) lut[N - half + t] = (uint64_t)0 - tmp[t];
}

}  // namespace fhs
