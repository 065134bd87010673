// String layer: host-side mirror of the reference's MyServerKey methods
// (src/server_key/mod.rs, src/server_key/trim.rs, src/utils.rs:28-112) as DAG constructors.
#pragma once
#include <vector>

#include "radix.h"

namespace fhs {

using FStr = std::vector<FChar>;   // FheString.bytes (fhestring.rs:6-9); cst = trivial 32 (:24)

struct StrError {
    int code = 0;   // FHS_ERR_LIMIT for the reference's panic!()
    const char *msg = "";
};

class Strings {
  public:
    explicit Strings(Engine *e) : e_(e) {}
    StrError err;

    FStr clear(const char *s, size_t n) const;   // the `_clear` twins trivially encrypt the pattern
    FChar contains(const FStr &s, const FStr &needle);
    FChar starts_with(const FStr &s, const FStr &pat);
    FChar ends_with(const FStr &s, const FStr &needle);
    FChar find(const FStr &s, const FStr &pat);
    FChar rfind(const FStr &s, const FStr &pat);
    FChar is_empty(const FStr &s);
    FChar len(const FStr &s);
    FChar eq(const FStr &a, const FStr &b);
    FChar ne(const FStr &a, const FStr &b);
    FChar eq_ignore_case(const FStr &a, const FStr &b);
    FChar comparison(const FStr &a, const FStr &b, int cmp);   // 0 lt, 1 le, 2 gt, 3 ge
    // positional half on two equally long slices: (any position differs, verdict at the first differing position)
    void f_cmp_partial(const FStr &a, const FStr &b, int cmp, FChar *any_diff_out, FChar *verdict_out);
    // combine such partials over consecutive ranges: the first range that differs decides, `tie` if none does
    FChar flags_first_decides(const FStr &any_diff, const FStr &verdict, int tie);
    // window-sharded find (multi-GPU): the match flags of a slice's windows, and the rest of find on ALL flags in string
    // order (index of the first one set, 255 if none) -- fhs_dist_str_find exchanges the flags in between
    std::vector<Ref> f_find_window_flags(const FStr &s, const FStr &pat);
    FChar f_find_from_flags(const std::vector<Ref> &flags);
    FStr to_upper(const FStr &s);
    FStr to_lower(const FStr &s);
    FStr replace(const FStr &s, const FStr &from, const FStr &to);
    FStr replacen(const FStr &s, const FStr &from, const FStr &to, const FChar &n);
    FStr repeat(const FStr &s, const FChar &n);
    FStr repeat_clear(const FStr &s, size_t n);
    FStr concatenate(const FStr &a, const FStr &b);
    FStr strip_prefix(const FStr &s, const FStr &pat, FChar *found);
    FStr strip_suffix(const FStr &s, const FStr &pat, FChar *found);
    FStr trim_end(const FStr &s);
    FStr trim_start(const FStr &s);
    FStr trim(const FStr &s);
    FStr bubble_zeroes_right(const FStr &s);
    FChar flags_or(const FStr &flags);
    FChar flags_and(const FStr &flags);
    // split family (src/server_key/split.rs): result[buffer][position] + pattern_found, like FheSplit
    enum SplitKind { SPLIT = 0, SPLIT_INCLUSIVE, SPLIT_TERMINATOR, SPLITN, RSPLIT, RSPLIT_TERMINATOR, RSPLITN,
                     RSPLIT_ONCE, SPLIT_ASCII_WHITESPACE };
    std::vector<FStr> split_family(int kind, const FStr &s, const FStr &pat, const FChar *n, FChar *found);

  private:
    // char-level ops of the split family: as written, or single-block-flag versions in fused mode
    FChar s_eq(const FChar &a, const FChar &b);
    FChar s_ite(const FChar &flag, const FChar &tv, const FChar &fv);
    FChar s_not(const FChar &flag);
    FChar split_match(bool reverse, size_t i, const FStr &s, const FStr &pat, FStr &mask);
    std::vector<FStr> xsplit(const FStr &s, const FStr &pat, bool inclusive, bool terminator, const FChar *n,
                             bool reverse, FChar *found);
    // fused formulation of the distribution phase (buffer ids as prefix counts, multi-digit, membership selects)
    bool f_split_distribute(const FStr &s, const FStr &pat, const FChar *n, bool reverse, std::vector<FStr> &result,
                            FChar *found);
    void split_cleanup(std::vector<FStr> &result, const FStr &pat, bool inclusive, bool terminator, const FChar *n);
    FStr on_support(const FStr &row, FStr (Strings::*op)(const FStr &, const FStr &, const FStr &), const FStr &a,
                    const FStr &b);
    std::vector<FStr> split_ws(const FStr &s, FChar *found);
    Engine *e_;
    bool fused() const { return e_->mode == 1; }
    FChar t(uint8_t v) const { return ch_trivial(e_, v); }
    FStr longer_from(const FStr &s, const FStr &from, FStr to, const FChar &n, bool use_counter);
    FStr shorter_from(const FStr &s, const FStr &from, const FStr &to, const FChar &n, bool use_counter);

    // fused-mode helpers on single-block flags
    Ref and_tree(std::vector<Ref> flags);
    Ref or_tree(std::vector<Ref> flags);
    Ref onehot_or(std::vector<Ref> flags);   // OR of flags of which at most one is set: noise-budget-wide groups
    std::vector<Ref> block_eq_flags(const FChar &a, const FChar &b);
    Ref window_match(const FStr &s, size_t at, const FStr &pat);
    FChar count_flags(std::vector<Ref> flags);   // sum of 0/1 flags mod 256 as a 4-block char
    Ref is_upper_flag(const FChar &c, bool lower);
    std::vector<Ref> prefix_or(const std::vector<Ref> &f);
    FChar f_find(const FStr &s, const FStr &pat);
    FChar f_comparison(const FStr &a, const FStr &b, int cmp);
    // oblivious compaction: replaces the O(n^2) bubble of utils.rs:28-46 in fused mode
    typedef std::vector<Ref> Num;   // little-endian base-4 digits, clean (<= 3)
    std::vector<Num> flag_prefix_counts(const std::vector<Ref> &flags, size_t digits);   // exclusive
    Num num_add(const std::vector<const Num *> &ops, size_t digits);
    Num count_digits(const Ref *flags, size_t k, size_t digits);   // sum of <= 15 flags as a base-4 number, within the noise budget
    std::vector<Num> num_exclusive_scan(const std::vector<Num> &x, size_t digits);
    FStr f_compact(const FStr &s);
    FChar ite_flag(const Ref &flag, const FChar &t, const FChar &f);
    std::vector<Ref> suffix_or(const std::vector<Ref> &f);            // exclusive: OR_{k>i}
    Ref char_nonzero(const FChar &c);                                 // 1 block: c != 0
    Ref char_zero_test(const FChar &c, bool want_zero);               // 1 block, 1 bootstrap: c == 0 / c != 0
    Ref char_significant(const FChar &c);                             // 1 block: c is neither NUL nor whitespace
    FChar position_of(const std::vector<Ref> &pick, size_t index_offset, const Ref *absent_flag, int absent_value);
    FChar first_index(const std::vector<Ref> &before, const Ref &found);
    FChar f_eq_ignore_case(const FStr &a, const FStr &b);
    // lexicographic order as a tree of three-state values s = sign(a - b) in {-1, 0, 1}, most significant first
    std::vector<Ref> cmp_leaves(const FStr &a, const FStr &b);          // one per nibble pair, missing characters are 0
    Ref cmp_root_sum(std::vector<Ref> states);                          // 8 + 4 s1 + 2 s2 + s3 of the last <= 3 states   // index of the first set flag, 255 if none
    FChar f_rfind(const FStr &s, const FStr &pat);
    FChar f_ends_with(const FStr &s, const FStr &needle, std::vector<Ref> *pick_out);
    FStr f_trim(const FStr &s, bool from_end);
    FStr f_replace_expand(const FStr &s, const FStr &from, const FStr &to);
    FChar f_contains(const FStr &s, const FStr &needle);
    FChar f_len(const FStr &s);
    FChar f_eq(const FStr &a, const FStr &b);
    FStr f_case(const FStr &s, bool to_lower);
};

}  // namespace fhs
