// Radix layer: the 13 tfhe::integer::ServerKey ops the reference calls from
// src/ciphertext/fheasciichar.rs:23-102, on 4 blocks of PARAM_MESSAGE_2_CARRY_2 as lazy DAG nodes.
// Block decompositions and trivial-folding rules are identical to oracle/radix.py (CipherChar) so
// that ciphertexts can be compared bit for bit.
#pragma once
#include <initializer_list>

#include "engine.h"

namespace fhs {

struct FChar {          // FheAsciiChar (fheasciichar.rs:8-10): little-endian 2-bit digits
    Ref b[4];
};

Ref lin(Engine *e, std::initializer_list<std::pair<int64_t, const Ref *>> terms, int konst = 0);
Ref pbs(const Ref &x, int lut);
Ref trivial_block(Engine *e, int v);

FChar ch_trivial(Engine *e, uint8_t v);                       // encrypt_trivial :17-25
FChar ch_flag(Engine *e, const Ref &blk);                     // BooleanBlock::into_radix(4) :37
FChar ch_eq(const FChar &a, const FChar &b);                  // :35-38
FChar ch_ne(const FChar &a, const FChar &b);                  // :40-43
FChar ch_le(const FChar &a, const FChar &b);                  // :45-48
FChar ch_lt(const FChar &a, const FChar &b);                  // :50-53
FChar ch_ge(const FChar &a, const FChar &b);                  // :55-58
FChar ch_gt(const FChar &a, const FChar &b);                  // :60-63
FChar ch_bitand(const FChar &a, const FChar &b);              // :65-72
FChar ch_bitor(const FChar &a, const FChar &b);               // :74-81
FChar ch_sub(const FChar &a, const FChar &b);                 // :83-86
FChar ch_add(const FChar &a, const FChar &b);                 // :88-91
FChar ch_ite(const FChar &cond, const FChar &t, const FChar &f);  // :93-104
FChar ch_flip(const FChar &a);                                // :161-168
FChar ch_is_whitespace(const FChar &a);                       // :106-130
FChar ch_is_uppercase(const FChar &a);                        // :132-144
FChar ch_is_lowercase(const FChar &a);                        // :146-158

// single-block building blocks shared with the fused string layer
Ref blk_eq_flag(const FChar &a, const FChar &b);              // 1 block: a == b
Ref blk_ne_flag(const FChar &a, const FChar &b);
Ref blk_cmp_flag(const FChar &a, const FChar &b, int lut);    // lut in {CMP_LT, CMP_LE, CMP_GT, CMP_GE}
Ref blk_nonzero_flag(const FChar &a);                         // scalar_ne(a, 0) :99

}  // namespace fhs
