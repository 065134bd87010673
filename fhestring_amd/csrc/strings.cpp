#include "strings.h"

#include <algorithm>
#include <set>

#include "../../include/fhestring_hip.h"

namespace fhs {

static size_t adjust_end_of_pattern(size_t e) { return e == 0 ? 1 : e; }   // utils.rs:106-112
static Ref sum_refs(Engine *e, const Ref *r, size_t n);

FStr Strings::clear(const char *s, size_t n) const {
    FStr r;
    for (size_t i = 0; i < n; i++) r.push_back(t((uint8_t)s[i]));
    return r;
}

// ---------------------------------------------------------------------------------------------
// as written (same op sequence as the reference; cf. oracle/strings.py)
// ---------------------------------------------------------------------------------------------
FStr Strings::bubble_zeroes_right(const FStr &in) {          // utils.rs:28-46
    if (fused()) return f_compact(in);
    FStr s = in;
    const FChar zero = t(0);
    for (size_t pass = 0; pass < s.size(); pass++)
        for (size_t i = 0; i + 1 < s.size(); i++) {
            FChar swap = ch_eq(s[i], zero);
            FChar a = ch_ite(swap, s[i + 1], s[i]);
            FChar b = ch_ite(swap, zero, s[i + 1]);
            s[i] = a;
            s[i + 1] = b;
        }
    return s;
}

FStr Strings::to_upper(const FStr &s) {                      // mod.rs:65-84
    if (fused()) return f_case(s, false);
    const FChar zero = t(0), cst = t(32);
    FStr r;
    for (const FChar &b : s) r.push_back(ch_sub(b, ch_ite(ch_flip(ch_is_lowercase(b)), zero, cst)));
    return r;
}
FStr Strings::to_lower(const FStr &s) {                      // mod.rs:110-128
    if (fused()) return f_case(s, true);
    const FChar zero = t(0), cst = t(32);
    FStr r;
    for (const FChar &b : s) r.push_back(ch_add(b, ch_ite(ch_flip(ch_is_uppercase(b)), zero, cst)));
    return r;
}

FChar Strings::contains(const FStr &s, const FStr &needle) { // mod.rs:151-182
    if (s.empty() && needle.empty()) return t(1);
    if (needle.size() > s.size()) return t(0);
    if (fused()) return f_contains(s, needle);
    FChar result = t(0);
    const FChar one = t(1);
    for (size_t i = 0; i + needle.size() <= s.size(); i++) {
        FChar cur = one;
        for (size_t j = 0; j < needle.size(); j++) cur = ch_bitand(cur, ch_eq(s[i + j], needle[j]));
        result = ch_bitor(result, cur);
    }
    return result;
}

FChar Strings::ends_with(const FStr &s, const FStr &needle) {   // mod.rs:241-288
    if (s.empty() && needle.empty()) return t(1);
    if (needle.size() > s.size()) return t(0);
    if (fused()) return f_ends_with(s, needle, nullptr);
    FChar result = t(0);
    const FChar one = t(1), zero = t(0);
    for (size_t i = 0; i + needle.size() <= s.size(); i++) {
        FChar cur = one, nonzero = one;
        for (size_t j = 0; j < needle.size(); j++) {
            cur = ch_bitand(cur, ch_eq(s[i + j], needle[j]));
            nonzero = ch_bitand(nonzero, ch_ne(s[i + j], zero));
        }
        result = ch_ite(nonzero, cur, result);
    }
    return result;
}

FChar Strings::starts_with(const FStr &s, const FStr &pat) { // mod.rs:344-371
    if (pat.size() > s.size()) return t(0);
    if (s.empty() && pat.empty()) return t(1);
    if (fused()) {
        if (pat.empty()) return t(1);
        return ch_flag(e_, window_match(s, 0, pat));
    }
    FChar result = t(1);
    for (size_t i = 0; i < std::min(pat.size(), s.size()); i++) result = ch_bitand(result, ch_eq(s[i], pat[i]));
    return result;
}

FChar Strings::is_empty(const FStr &s) {                     // mod.rs:431-451
    if (s.empty()) return t(1);
    const FChar zero = t(0);
    if (fused()) {
        std::vector<Ref> f;
        for (const FChar &c : s) f.push_back(char_zero_test(c, true));
        return ch_flag(e_, and_tree(f));
    }
    FChar result = t(1);
    for (const FChar &c : s) result = ch_bitand(result, ch_eq(c, zero));
    return result;
}

FChar Strings::len(const FStr &s) {                          // mod.rs:478-493
    const FChar zero = t(0);
    if (s.empty()) return zero;
    if (fused()) return f_len(s);
    FChar result = t(0);
    for (const FChar &c : s) result = ch_add(result, ch_ne(c, zero));
    return result;
}

FStr Strings::repeat_clear(const FStr &s, size_t n) {        // mod.rs:517-538
    if (n == 0) return {};
    FStr r;
    for (size_t k = 0; k < n; k++) r.insert(r.end(), s.begin(), s.end());
    return bubble_zeroes_right(r);
}

FStr Strings::repeat(const FStr &s, const FChar &n) {        // mod.rs:567-591
    const FChar zero = t(0);
    FStr r(FHS_MAX_REPETITIONS * s.size(), zero);
    for (size_t i = 0; i < FHS_MAX_REPETITIONS; i++) {
        FChar flag = ch_lt(t((uint8_t)i), n);
        for (size_t j = 0; j < s.size(); j++) r[i * s.size() + j] = ch_ite(flag, s[j], zero);
    }
    return bubble_zeroes_right(r);
}

FStr Strings::replace(const FStr &s, const FStr &from, const FStr &to) {   // mod.rs:624-653
    const FChar n = t(0);
    if (from.size() >= to.size()) return longer_from(s, from, to, n, false);
    return shorter_from(s, from, to, n, false);
}
FStr Strings::replacen(const FStr &s, const FStr &from, const FStr &to, const FChar &n) {   // mod.rs:1729-1757
    if (from.size() >= to.size()) return longer_from(s, from, to, n, true);
    return shorter_from(s, from, to, n, true);
}

FStr Strings::longer_from(const FStr &s, const FStr &from, FStr to, const FChar &n, bool use_counter) {   // mod.rs:828-882
    const FChar zero = t(0), one = t(1);
    FStr data = s;
    data.push_back(zero);                                   // :841
    while (to.size() < from.size()) to.push_back(zero);     // :847-849
    FChar counter = t(0);
    FStr result = data;
    if (fused() && !use_counter && !from.empty() && from.size() <= 10 && from.size() <= result.size()) {
        // The loop below writes `to` over every matching window in ascending order, flags taken on the ORIGINAL data: the
        // LAST matching window that covers a position wins.  Re-associated per position p: sel_k = window p-k matches and
        // none of p-k+1 .. p does (one bootstrap each on 5 * match[p-k] + the later flags: equals 5 exactly then), none =
        // no window covers p; result[p] = none ? data[p] : to[the k that is set] -- (m + 1) selects per block instead of
        // 2 m, two levels deep instead of m chained ones.
        const size_t m = from.size(), end = adjust_end_of_pattern(result.size() - from.size());
        std::vector<Ref> match(end);
        for (size_t i = 0; i < end; i++) match[i] = window_match(data, i, from);
        for (size_t p = 0; p < result.size(); p++) {
            std::vector<Ref> later;                          // match flags of windows p-j, j < k, that exist
            std::vector<std::pair<size_t, Ref>> sel;         // (k, flag)
            for (size_t k = 0; k < m && k <= p; k++) {
                const size_t i = p - k;
                if (i >= end) continue;
                if (later.empty()) sel.push_back({k, match[i]});
                else {
                    Term tt[12];
                    size_t c = 0;
                    tt[c++] = {5, match[i].id()};
                    for (const Ref &l : later) tt[c++] = {1, l.id()};
                    sel.push_back({k, pbs(Ref(e_, e_->lin(tt, c, 0)), lut_is_k(5))});
                }
                later.push_back(match[i]);
            }
            if (sel.empty()) continue;                       // no window reaches this position
            Ref none = pbs(sum_refs(e_, later.data(), later.size()), lut_is_k(0));
            for (int b = 0; b < 4; b++) {
                Term tt[12];
                size_t c = 0;
                Ref keepv = pbs(lin(e_, {{4, &none}, {1, &data[p].b[b]}}), LUT_SEL_T);
                std::vector<Ref> parts{keepv};
                for (auto &kv : sel) parts.push_back(pbs(lin(e_, {{4, &kv.second}, {1, &to[kv.first].b[b]}}), LUT_SEL_T));
                for (const Ref &x : parts) tt[c++] = {1, x.id()};
                result[p].b[b] = Ref(e_, e_->lin(tt, c, 0));
            }
        }
        return bubble_zeroes_right(result);                 // :881
    }
    if (from.size() <= result.size()) {
        const size_t end = adjust_end_of_pattern(result.size() - from.size());
        for (size_t i = 0; i < end; i++) {
            FChar flag = one;
            if (fused() && !from.empty()) flag = ch_flag(e_, window_match(data, i, from));
            else for (size_t j = 0; j < from.size(); j++) flag = ch_bitand(flag, ch_eq(from[j], data[i + j]));
            if (use_counter) {                              // :868-872
                counter = ch_add(counter, flag);
                flag = ch_bitand(flag, ch_ge(n, counter));
            }
            for (size_t k = 0; k < to.size(); k++)
                result[i + k] = fused() ? ite_flag(flag.b[0], to[k], result[i + k]) : ch_ite(flag, to[k], result[i + k]);
        }
    }
    return bubble_zeroes_right(result);                     // :881
}

FStr Strings::shorter_from(const FStr &s, const FStr &from, const FStr &to, const FChar &n, bool use_counter) {   // mod.rs:885-980
    if (fused() && !use_counter && !from.empty() && from.size() <= 8) return f_replace_expand(s, from, to);
    const FChar zero = t(0), one = t(1);
    FStr data = s;
    data.push_back(zero);
    const size_t size_diff = to.size() - from.size();
    FChar counter = t(0);
    size_t max_len = data.empty() ? to.size() : to.size() * data.size() + data.size();
    if (from.empty()) max_len = (data.size() + (data.size() + 1) * to.size()) + 1;
    FStr result = data;
    result.resize(max_len, zero);
    FStr copy_buffer(max_len, zero);
    FStr ignore_mask(max_len, one);
    for (size_t i = 0; i + to.size() < result.size(); i++) {
        FChar flag = one;
        for (size_t j = 0; j < from.size(); j++) {
            flag = ch_bitand(flag, ch_eq(from[j], result[i + j]));
            flag = ch_bitand(flag, ignore_mask[i + j]);
        }
        if (from.empty()) flag = (i % (to.size() + 1) == 0) ? one : zero;
        if (use_counter) {
            counter = ch_add(counter, flag);
            flag = ch_bitand(flag, ch_ge(n, counter));
        }
        for (size_t k = 0; k < max_len; k++) copy_buffer[k] = ch_ite(flag, result[k], zero);
        for (size_t k = 0; k < to.size(); k++) {
            result[i + k] = ch_ite(flag, to[k], result[i + k]);
            ignore_mask[i + k] = ch_bitand(ignore_mask[i + k], ch_ite(flag, zero, one));
        }
        for (size_t k = i + to.size(); k < max_len; k++)
            result[k] = ch_ite(flag, copy_buffer[k - size_diff], result[k]);
    }
    return result;
}

FChar Strings::rfind(const FStr &s_in, const FStr &pat) {    // mod.rs:727-790
    const FChar one = t(1), zero = t(0);
    FStr s = s_in;
    s.push_back(zero);
    FChar pos = t(FHS_MAX_FIND_LENGTH);
    if (s.size() >= FHS_MAX_FIND_LENGTH + pat.size()) {
        err = {FHS_ERR_LIMIT, "Maximum supported size for find reached"};
        return zero;
    }
    if (fused()) return f_rfind(s, pat);
    if (pat.empty()) {
        FChar last = zero;
        for (size_t i = 0; i < s.size(); i++) last = ch_ite(ch_ne(s[i], zero), t((uint8_t)(i + 1)), last);
        return last;
    }
    if (pat.size() > s.size()) return t(255);
    const size_t end = adjust_end_of_pattern(s.size() - pat.size());
    for (size_t i = 0; i < end; i++) {
        FChar flag = one;
        for (size_t j = 0; j < pat.size(); j++) flag = ch_bitand(flag, ch_eq(pat[j], s[i + j]));
        pos = ch_ite(flag, t((uint8_t)i), pos);
    }
    return pos;
}

FChar Strings::find(const FStr &s, const FStr &pat) {        // mod.rs:1010-1053
    if (s.empty() && pat.empty()) return t(0);
    const FChar one = t(1);
    FChar pos = t(FHS_MAX_FIND_LENGTH);
    if (s.size() >= FHS_MAX_FIND_LENGTH + pat.size()) {      // :1025-1027
        err = {FHS_ERR_LIMIT, "Maximum supported size for find reached"};
        return t(0);
    }
    if (pat.size() > s.size()) return t(255);
    if (fused()) return f_find(s, pat);
    for (size_t i = s.size() - pat.size() + 1; i-- > 0;) {
        FChar flag = one;
        for (size_t j = pat.size(); j-- > 0;) flag = ch_bitand(flag, ch_eq(pat[j], s[i + j]));
        pos = ch_ite(flag, t((uint8_t)i), pos);
    }
    return pos;
}

FChar Strings::eq(const FStr &a, const FStr &b) {            // mod.rs:1122-1149
    if (fused()) return f_eq(a, b);
    const FChar zero = t(0), one = t(1);
    FChar is_eq = one;
    FChar len_ne = ch_ne(len(a), len(b));
    for (size_t i = 0; i < std::min(a.size(), b.size()); i++) {
        FChar same = ch_eq(a[i], b[i]);
        FChar both0 = ch_bitand(ch_eq(a[i], zero), ch_eq(b[i], zero));
        is_eq = ch_bitand(is_eq, ch_bitor(both0, same));
    }
    return ch_ite(len_ne, zero, is_eq);
}
FChar Strings::ne(const FStr &a, const FStr &b) {            // mod.rs:1178-1186
    FChar r = eq(a, b);
    if (fused()) {   // 1 - flag on the single live block, no PBS
        Ref one = trivial_block(e_, 1);
        return ch_flag(e_, lin(e_, {{1, &one}, {-1, &r.b[0]}}));
    }
    return ch_flip(r);
}
FChar Strings::eq_ignore_case(const FStr &a, const FStr &b) {   // mod.rs:1221-1231
    if (fused()) return f_eq_ignore_case(a, b);
    return eq(to_lower(a), to_lower(b));
}

// eq(to_lower(a), to_lower(b)) without folding either string: two characters are equal ignoring case iff their low
// nibbles are equal and their high nibbles are equal, or differ in the case bit 0x20 only with both characters letters
// (top two bits 01, and the low nibble in 1..15 on the rows 0x4_ / 0x6_, in 0..10 on the rows 0x5_ / 0x7_: with equal
// low nibbles one range test serves both).  6 bootstraps per position in 3 levels instead of 12 (case flags and folds
// of both strings, then the nibble tests); the length condition of eq is the tail test of f_eq (NUL has no case).
FChar Strings::f_eq_ignore_case(const FStr &a, const FStr &b) {
    std::vector<Ref> f;
    const size_t common = std::min(a.size(), b.size());
    auto clean = [&](FChar c) {                              // operands that are sums of bootstrap outputs (a select,
        for (int k = 0; k < 4; k++)                          // a case shift) are refreshed: the packings below weigh
            if (e_->sum_c2(c.b[k].id()) > 1) c.b[k] = pbs(c.b[k], LUT_MSG);   // them with up to 4
        return c;
    };
    for (size_t i = 0; i < common; i++) {
        const FChar x = clean(a[i]), y = clean(b[i]);
        Ref e_lo = pbs(lin(e_, {{1, &x.b[0]}, {4, &x.b[1]}, {-1, &y.b[0]}, {-4, &y.b[1]}}), LUT_IS0);
        Ref b3 = pbs(lin(e_, {{1, &x.b[3]}, {4, &y.b[3]}}), LUT_EQIC_B3);
        Ref z = pbs(lin(e_, {{1, &x.b[2]}, {4, &y.b[2]}}), LUT_EQIC_Z);
        Ref lo = pbs(lin(e_, {{1, &x.b[0]}, {4, &x.b[1]}}), LUT_EQIC_LO);
        Ref s1 = pbs(lin(e_, {{1, &z}, {4, &lo}}), LUT_EQIC_S1);
        // the three partial verdicts as one sum without collisions between a passing and a failing combination:
        // passing (B3, S1, E_lo) = (1, 1, 1), (2, 1, 1), (2, 2, 1) -> 11, 12, 15; sum c^2 = 59
        f.push_back(pbs(lin(e_, {{1, &b3}, {3, &s1}, {7, &e_lo}}), LUT_EQIC_FIN));
    }
    const FStr &longer = a.size() > b.size() ? a : b;
    for (size_t i = common; i < longer.size(); i++) f.push_back(char_zero_test(longer[i], true));
    return ch_flag(e_, and_tree(f));
}

FStr Strings::strip_prefix(const FStr &s, const FStr &pat, FChar *found) {   // mod.rs:1261-1307
    const FChar zero = t(0), one = t(1);
    FStr result = s;
    FChar flag = one;
    const size_t end = std::min(pat.size(), result.size());
    if (pat.size() > result.size()) {
        *found = zero;
        return result;
    }
    if (end == 0 && !pat.empty() && s.empty()) flag = zero;
    for (size_t j = 0; j < end; j++) flag = ch_bitand(flag, ch_eq(pat[j], result[j]));
    for (size_t j = 0; j < std::min(pat.size(), result.size()); j++) result[j] = ch_ite(flag, zero, result[j]);
    *found = flag;
    return bubble_zeroes_right(result);
}

FStr Strings::strip_suffix(const FStr &s_in, const FStr &needle, FChar *found) {   // mod.rs:1335-1404
    const FChar one = t(1), zero = t(0), t255 = t(255);
    FStr s = s_in;
    if (needle.size() > s.size()) {
        *found = zero;
        return s;
    }
    const size_t end = s.size() - needle.size();
    if (fused()) {
        // pos == i exactly for the window picked by ends_with (last window without padding, if it matches)
        std::vector<Ref> pick;
        FChar r = f_ends_with(s, needle, &pick);
        *found = r;
        for (size_t k = 0; k < s.size(); k++) {
            std::vector<Ref> cover;   // windows that contain position k; at most one pick is set overall
            for (size_t i = (k + 1 >= needle.size() ? k + 1 - needle.size() : 0); i <= std::min(k, end); i++)
                cover.push_back(pick[i]);
            if (cover.empty()) continue;
            Ref zk = cover.size() <= 15 ? sum_refs(e_, cover.data(), cover.size()) : or_tree(cover);
            s[k] = ite_flag(zk, zero, s[k]);
        }
        return s;
    }
    FChar pos = t(255);
    for (size_t i = 0; i <= end; i++) {
        FChar fnd = one, nonzero = one;
        for (size_t j = 0; j < needle.size(); j++) {
            fnd = ch_bitand(fnd, ch_eq(s[i + j], needle[j]));
            nonzero = ch_bitand(nonzero, ch_ne(s[i + j], zero));
        }
        FChar cur = ch_ite(fnd, t((uint8_t)i), t255);
        pos = ch_ite(nonzero, cur, pos);
    }
    *found = ch_ne(pos, t255);
    for (size_t i = 0; i <= end; i++) {
        FChar mask = ch_eq(t((uint8_t)i), pos);
        for (size_t j = 0; j < needle.size(); j++) s[i + j] = ch_ite(mask, zero, s[i + j]);
    }
    return s;
}

FChar Strings::comparison(const FStr &a_in, const FStr &b_in, int cmp) {   // mod.rs:1470-1541
    if (fused()) return f_comparison(a_in, b_in, cmp);
    const FChar zero = t(0), t255 = t(255);
    FStr a = a_in, b = b_in;
    size_t min_len = std::min(a.size(), b.size());
    FChar seen = zero, became = zero, ret = t(255);
    if (min_len == 0) {                                      // :1490-1494
        a.push_back(zero);
        b.push_back(zero);
        min_len = 1;
    }
    auto op = [&](const FChar &x, const FChar &y) {
        switch (cmp) {
            case 0: return ch_lt(x, y);
            case 1: return ch_le(x, y);
            case 2: return ch_gt(x, y);
            default: return ch_ge(x, y);
        }
    };
    for (size_t i = 0; i < min_len; i++) {
        FChar c = op(a[i], b[i]);
        seen = ch_bitor(seen, ch_ne(a[i], b[i]));
        FChar flag = ch_bitand(seen, ch_flip(became));
        became = ch_bitor(became, flag);
        ret = ch_ite(flag, c, ret);
    }
    FChar sub_eq = ch_eq(ret, t255);
    FChar l1 = len(a), l2 = len(b);
    FChar leq = ch_eq(l1, l2), lgt = ch_gt(l1, l2), llt = ch_lt(l1, l2);
    FChar by_len;
    switch (cmp) {
        case 3: by_len = ch_bitor(leq, lgt); break;
        case 1: by_len = ch_bitor(leq, llt); break;
        case 2: by_len = lgt; break;
        default: by_len = llt; break;
    }
    return ch_ite(sub_eq, by_len, ret);
}

FStr Strings::concatenate(const FStr &a, const FStr &b) {    // mod.rs:1864-1875
    FStr r = a;
    r.insert(r.end(), b.begin(), b.end());
    return bubble_zeroes_right(r);
}

FStr Strings::trim_end(const FStr &s) {                      // trim.rs:36-57
    if (fused()) return f_trim(s, true);
    const FChar zero = t(0);
    FChar stop = zero;
    FStr r(s.size(), zero);
    for (size_t i = s.size(); i-- > 0;) {
        FChar not_ws = ch_flip(ch_is_whitespace(s[i]));
        stop = ch_bitor(stop, ch_bitand(not_ws, ch_ne(s[i], zero)));
        r[i] = ch_ite(stop, s[i], zero);
    }
    return r;
}
FStr Strings::trim_start(const FStr &s) {                    // trim.rs:86-115
    if (fused()) return f_compact(f_trim(s, false));
    const FChar zero = t(0);
    FChar stop = zero;
    FStr r(s.size(), zero);
    for (size_t i = 0; i < s.size(); i++) {
        FChar not_ws = ch_flip(ch_is_whitespace(s[i]));
        stop = ch_bitor(stop, ch_bitand(not_ws, ch_ne(s[i], zero)));
        r[i] = ch_ite(stop, s[i], zero);
    }
    return bubble_zeroes_right(r);
}
FStr Strings::trim(const FStr &s) { return trim_start(trim_end(s)); }   // trim.rs:146-149

// ---------------------------------------------------------------------------------------------
// split family (src/server_key/split.rs), same loops as the reference; in fused mode the char-level
// helpers work on single-block flags and the buffers are compacted instead of bubble-sorted
// ---------------------------------------------------------------------------------------------
FChar Strings::s_eq(const FChar &a, const FChar &b) {
    return fused() ? ch_flag(e_, and_tree(block_eq_flags(a, b))) : ch_eq(a, b);
}
FChar Strings::s_ite(const FChar &flag, const FChar &tv, const FChar &fv) {
    return fused() ? ite_flag(flag.b[0], tv, fv) : ch_ite(flag, tv, fv);
}
FChar Strings::s_not(const FChar &flag) {
    if (!fused()) return ch_flip(flag);
    Ref one = trivial_block(e_, 1);
    return ch_flag(e_, lin(e_, {{1, &one}, {-1, &flag.b[0]}}));
}

FChar Strings::split_match(bool reverse, size_t i, const FStr &s, const FStr &pat, FStr &mask) {
    const FChar zero = t(0), one = t(1);
    FChar found = one;
    if (reverse) {                                           // rsplit_pattern_matching, split.rs:10-67
        if (pat.empty()) {
            FChar cur_pad = s_eq(s[i], zero);
            if (i >= 1) {
                FChar prev_nonpad = s_not(s_eq(s[i - 1], zero));
                found = s_ite(ch_bitand(prev_nonpad, cur_pad), one, zero);
                found = ch_bitor(found, s_ite(cur_pad, zero, one));
            } else found = s_ite(cur_pad, zero, one);
        } else if (pat.size() > s.size() || i + pat.size() >= s.size()) {
            found = zero;
        } else {
            for (size_t j = 0; j < pat.size(); j++) {
                found = ch_bitand(found, s_eq(s[i + j], pat[j]));
                found = ch_bitand(found, mask[i + j]);
            }
        }
    } else {                                                 // split_pattern_matching, split.rs:69-108
        if (pat.size() > s.size() || i + 1 < pat.size()) {
            found = zero;
        } else {
            for (size_t j = 0; j < pat.size(); j++) {
                const size_t k = i + 1 - pat.size() + j;
                found = ch_bitand(found, s_eq(s[k], pat[j]));
                found = ch_bitand(found, mask[k]);
            }
        }
    }
    for (size_t j = 0; j < pat.size(); j++)                  // no overlapping re-match
        if (i + j < s.size()) mask[i + j] = ch_bitand(mask[i + j], s_ite(found, zero, one));
    return found;
}

std::vector<FStr> Strings::xsplit(const FStr &s_in, const FStr &pat, bool inclusive, bool terminator,
                                  const FChar *n, bool reverse, FChar *found_out) {   // _rsplit :307-393, _split :883-988
    const FChar zero = t(0), one = t(1);
    FStr s = s_in;
    s.push_back(zero);
    const size_t size = s.size();
    std::vector<FStr> result(size, FStr(size, zero));
    if (fused() && f_split_distribute(s, pat, n, reverse, result, found_out)) {
        split_cleanup(result, pat, inclusive, terminator, n);
        return result;
    }
    FChar cur_buf = zero, stop_inc = zero, global_found = zero, allow = zero;
    FStr mask(size, one);
    if (n) allow = ch_ne(*n, zero);
    if (!reverse && pat.empty() && n) {                      // split.rs:925-937
        FChar enc_len = len(s);
        FChar skip = ch_bitand(ch_gt(*n, one), ch_le(*n, enc_len));
        cur_buf = s_ite(skip, t(1), cur_buf);
    }
    for (size_t step = 0; step < size; step++) {
        const size_t i = reverse ? size - 1 - step : step;
        for (size_t j = 0; j < size; j++) {                  // copy_logic, split.rs:110-134
            FChar flag = s_eq(t((uint8_t)j), cur_buf);
            if (n) flag = ch_bitand(flag, allow);
            result[j][i] = s_ite(flag, s[i], result[j][i]);
        }
        FChar f = split_match(reverse, i, s, pat, mask);
        global_found = ch_bitor(global_found, f);
        if (!n) {                                            // handle_n_case, split.rs:136-173
            cur_buf = s_ite(f, ch_add(cur_buf, one), cur_buf);
        } else {
            stop_inc = ch_bitor(stop_inc, s_eq(cur_buf, ch_sub(*n, one)));
            cur_buf = s_ite(ch_bitand(f, s_not(stop_inc)), ch_add(cur_buf, one), cur_buf);
        }
    }
    split_cleanup(result, pat, inclusive, terminator, n);
    *found_out = global_found;
    return result;
}

// Runs `op(row, a, b)` on the SUPPORT of a buffer only: leading and trailing characters that are trivially (provably)
// NUL cannot take part in a match of a NUL-free pattern and end up behind the payload after the bubble anyway, so the
// O(L log L) cleanup works on the L positions a buffer can actually receive instead of the whole row.
FStr Strings::on_support(const FStr &row, FStr (Strings::*op)(const FStr &, const FStr &, const FStr &), const FStr &a,
                         const FStr &b) {
    auto triv0 = [&](const FChar &c) {
        for (int k = 0; k < 4; k++)
            if (!e_->is_triv(c.b[k].id()) || e_->triv_val(c.b[k].id()) != 0) return false;
        return true;
    };
    size_t lo = 0, hi = row.size();
    while (lo < hi && triv0(row[lo])) lo++;
    while (hi > lo && triv0(row[hi - 1])) hi--;
    FStr out(row.size(), t(0));
    if (lo == hi) return out;
    FStr sub(row.begin() + lo, row.begin() + hi);
    FStr r = (this->*op)(sub, a, b);
    for (size_t k = 0; k < std::min(r.size(), out.size()); k++) out[k] = r[k];
    return out;
}

// clear_pattern_from_result, split.rs:175-305
void Strings::split_cleanup(std::vector<FStr> &result, const FStr &pat, bool inclusive, bool terminator, const FChar *n) {
    const FChar zero = t(0), one = t(1);
    const size_t size = result.size();
    FStr to(pat.size(), zero);
    // fused mode: the same operations on each buffer's support (see on_support); as written: on the whole row
    auto do_replace = [&](const FStr &row) {
        return fused() ? on_support(row, &Strings::replace, pat, to) : replace(row, pat, to);
    };
    auto bubble3 = [](Strings *S, const FStr &r) { return S->bubble_zeroes_right(r); };
    (void)bubble3;
    auto do_bubble = [&](const FStr &row) {
        if (!fused()) return bubble_zeroes_right(row);
        struct H { static FStr f(Strings *S, const FStr &r) { return S->bubble_zeroes_right(r); } };
        // support trick without the member-pointer signature: trim by hand
        size_t lo = 0, hi = row.size();
        auto triv0 = [&](const FChar &c) {
            for (int k = 0; k < 4; k++)
                if (!e_->is_triv(c.b[k].id()) || e_->triv_val(c.b[k].id()) != 0) return false;
            return true;
        };
        while (lo < hi && triv0(row[lo])) lo++;
        while (hi > lo && triv0(row[hi - 1])) hi--;
        FStr out(row.size(), zero);
        if (lo == hi) return out;
        FStr r = H::f(this, FStr(row.begin() + lo, row.begin() + hi));
        for (size_t k = 0; k < r.size(); k++) out[k] = r[k];
        return out;
    };
    if (n) {
        FChar stop = zero;
        for (size_t i = 0; i < size; i++) {
            stop = ch_bitor(stop, s_eq(*n, ch_add(t((uint8_t)i), one)));
            FStr cur = do_bubble(result[i]);
            FStr rep = do_replace(cur);
            for (size_t j = 0; j < size; j++) result[i][j] = s_ite(stop, cur[j], rep[j]);
        }
    } else {
        for (size_t i = 0; i < size; i++)
            result[i] = inclusive ? do_bubble(result[i]) : do_replace(result[i]);
        if (terminator) {                                    // split.rs:266-302
            FChar nonzero_found = zero;
            for (size_t i = size; i-- > 0;) {
                FChar is_zero = one;
                for (size_t j = 0; j < size; j++) is_zero = ch_bitand(is_zero, s_eq(result[i][j], zero));
                FStr head(result[i].begin(), result[i].begin() + size);
                FChar starts = starts_with(head, pat);
                FChar del = ch_bitand(ch_bitand(starts, is_zero), s_not(nonzero_found));
                for (size_t j = 0; j < size; j++) result[i][j] = s_ite(del, zero, result[i][j]);
                nonzero_found = ch_bitor(nonzero_found, s_not(is_zero));
            }
        }
    }
}

// Fused distribution phase of the split family (split.rs:110-173, :307-393, :883-988 re-associated).
//
// The reference walks the string once; at step t it copies the current character into buffer `cur_buf` (an n x n grid
// of if_then_else on an encrypted u8 index) and then bumps `cur_buf` if a pattern occurrence ends (starts, for the
// reverse family) there and is not masked.  Here:
//   1. window flags for every step at once;
//   2. the mask as a countdown state machine: an accepted occurrence masks positions i .. i+m-1, which blocks the
//      next 2m-2 steps of the forward scan (m-1 of the reverse scan); 2 PBS per step, like the greedy replace;
//   3. cur_buf[t] = number of accepted occurrences before step t = an exclusive prefix count (log depth), as a
//      multi-digit base-4 number, so buffer indices are not limited to a u8; with a limit n (splitn / rsplitn /
//      rsplit_once) the counter stops at n - 1: cur_buf = min(count, n - 1);
//   4. membership of character t in buffer j = AND over the digits of [digit_d(cur_buf) == digit_d(j)] (one shared
//      LUT per digit value and step, one AND per (step, buffer)), only for the buffers step t can reach
//      (j <= ceil(t / (block + 1))), and a 1-PBS-per-block select instead of the 11-PBS equality + if_then_else.
// Returns false (nothing done) for the cases left to the loop above: empty pattern, or a pattern so long that the
// countdown does not fit the LUT (m > 4 forward, m > 8 reverse).
bool Strings::f_split_distribute(const FStr &s, const FStr &pat, const FChar *n, bool reverse, std::vector<FStr> &result,
                                 FChar *found_out) {
    const size_t size = s.size(), m = pat.size();
    if (m == 0) return false;
    const size_t block = reverse ? m - 1 : 2 * m - 2;        // steps blocked after an accepted occurrence
    if (block > 7) return false;
    const FChar zero = t(0), one = t(1);
    auto pos = [&](size_t step) { return reverse ? size - 1 - step : step; };

    // 1. window flags in scan order
    std::vector<Ref> w(size);
    for (size_t step = 0; step < size; step++) {
        const size_t i = pos(step);
        if (m > size) { w[step] = trivial_block(e_, 0); continue; }
        if (reverse) w[step] = i + m < size ? window_match(s, i, pat) : trivial_block(e_, 0);          // :47-49
        else w[step] = i + 1 >= m ? window_match(s, i + 1 - m, pat) : trivial_block(e_, 0);            // :85-87
    }
    // 2. accepted occurrences
    std::vector<Ref> f(size);
    if (block == 0) f = w;
    else {
        Ref state = trivial_block(e_, 0);
        for (size_t step = 0; step < size; step++) {
            Ref v = lin(e_, {{2, &state}, {1, &w[step]}});
            f[step] = pbs(v, LUT_GREEDY_SEL);
            state = pbs(v, LUT_GREEDY_NEXT0 + (int)block);
        }
    }
    *found_out = ch_flag(e_, or_tree(f));

    // 3. buffer index of every step
    const size_t max_count = (size + block) / (block + 1);   // occurrences are at least block + 1 steps apart
    size_t D = 1;
    while (((size_t)1 << (2 * D)) <= max_count) D++;
    if (n) D = std::max<size_t>(D, 4);
    std::vector<Num> cur = flag_prefix_counts(f, D);
    Ref allow;
    if (n) {
        allow = blk_nonzero_flag(*n);                        // n == 0: nothing is ever copied (:124-127)
        FChar n1 = ch_sub(*n, one);                          // the counter stops at n - 1 (u8 arithmetic, :153-157)
        for (size_t step = 0; step < size; step++) {
            FChar low;
            for (int d = 0; d < 4; d++) low.b[d] = cur[step][d];
            Ref lt = blk_cmp_flag(low, n1, LUT_CMP_LT);      // count < n - 1 on the low 4 digits
            if (D > 4) {
                Term tt[16];
                size_t k = 0;
                for (size_t d = 4; d < D; d++) tt[k++] = {1, cur[step][d].id()};
                Ref hi0 = pbs(Ref(e_, e_->lin(tt, k, 0)), LUT_IS0);
                lt = pbs(lin(e_, {{1, &lt}, {1, &hi0}}), LUT_IS2);
            }
            Num clamped(D);
            for (size_t d = 0; d < D; d++) {
                Ref a = pbs(lin(e_, {{4, &lt}, {1, &cur[step][d]}}), LUT_SEL_T);
                if (d < 4) {
                    Ref b = pbs(lin(e_, {{4, &lt}, {1, &n1.b[d]}}), LUT_SEL_F);
                    clamped[d] = lin(e_, {{1, &a}, {1, &b}});
                } else clamped[d] = a;
            }
            cur[step] = clamped;
        }
    }

    // 4. membership and copy
    for (size_t step = 0; step < size; step++) {
        const size_t i = pos(step);
        size_t jmax = std::min(size - 1, (step + block) / (block + 1));
        std::vector<Ref> q(D * 4);                           // q[4 d + v] = [digit d of cur_buf == v], made on demand
        auto digit_flag = [&](size_t d, int v) -> Ref & {
            Ref &r = q[4 * d + v];
            if (!r) r = pbs(cur[step][d], lut_is_k(v));
            return r;
        };
        for (size_t j = 0; j <= jmax; j++) {
            if ((j >> (2 * D)) != 0) break;                  // not representable: unreachable by construction
            std::vector<Ref> fl;
            for (size_t d = 0; d < D; d++) fl.push_back(digit_flag(d, (int)((j >> (2 * d)) & 3)));
            if (n) fl.push_back(allow);
            Ref mem = and_tree(fl);
            if (e_->is_triv(mem.id()) && e_->triv_val(mem.id()) == 0) continue;
            for (int b = 0; b < 4; b++) result[j][i].b[b] = pbs(lin(e_, {{4, &mem}, {1, &s[i].b[b]}}), LUT_SEL_T);
        }
    }
    return true;
}

std::vector<FStr> Strings::split_ws(const FStr &s, FChar *found_out) {   // split.rs:1377-1447
    const FChar zero = t(0), one = t(1);
    const size_t size = s.size();
    FChar cur_buf = zero, prev_ws = t(1), global_found = zero;
    std::vector<FStr> result(size, FStr(size, zero));
    auto is_ws = [&](const FChar &c) {
        if (!fused()) return ch_is_whitespace(c);
        // significant = neither NUL nor whitespace; whitespace = !significant & nonzero
        Ref sig = char_significant(c), nz = char_nonzero(c);
        return ch_flag(e_, lin(e_, {{1, &nz}, {-1, &sig}}));
    };
    for (size_t i = 0; i < size; i++) {
        FChar f = is_ws(s[i]);
        global_found = ch_bitor(global_found, f);
        FChar inc = ch_bitand(f, s_not(prev_ws));
        cur_buf = s_ite(inc, ch_add(cur_buf, one), cur_buf);
        FChar not_ws = s_not(f);
        for (size_t j = 0; j < size; j++) {
            FChar flag = ch_bitand(s_eq(t((uint8_t)j), cur_buf), not_ws);
            result[j][i] = s_ite(flag, s[i], result[j][i]);
        }
        prev_ws = f;
    }
    // (:1428-1436 re-tests every buffer char for whitespace: only non-whitespace was ever copied)
    for (size_t j = 0; j < size; j++) result[j] = bubble_zeroes_right(result[j]);
    *found_out = global_found;
    return result;
}

std::vector<FStr> Strings::split_family(int kind, const FStr &s, const FStr &pat, const FChar *n, FChar *found) {
    const FChar two = t(2);
    switch (kind) {
        case SPLIT: return xsplit(s, pat, false, false, nullptr, false, found);             // split.rs:989
        case SPLIT_INCLUSIVE: return xsplit(s, pat, true, false, nullptr, false, found);    // :1020
        case SPLIT_TERMINATOR: return xsplit(s, pat, false, true, nullptr, false, found);   // :1051
        case SPLITN: return xsplit(s, pat, false, false, n, false, found);                  // :1448
        case RSPLIT: return xsplit(s, pat, false, false, nullptr, true, found);             // :394
        case RSPLIT_TERMINATOR: return xsplit(s, pat, false, true, nullptr, true, found);   // :504
        case RSPLITN: return xsplit(s, pat, false, false, n, true, found);                  // :421
        case RSPLIT_ONCE: return xsplit(s, pat, false, false, &two, true, found);           // :462
        default: return split_ws(s, found);                                                 // :1377
    }
}

// ---------------------------------------------------------------------------------------------
// fused mode: single-block 0/1 flags, sums of up to 15 flags per PBS (the carry space holds 15),
// log_15-depth AND/OR trees.  Decrypts identically to the as-written mode.
// ---------------------------------------------------------------------------------------------
static Ref sum_refs(Engine *e, const Ref *r, size_t n) {
    // any length: the callers bound the VALUE of the sum (15 flags, one-hot picks, noise budget), not the number of
    // terms -- trivial blocks carry no noise and fold into the constant, so a sum over a mostly plaintext string can be
    // long (first_index on 256 trivial windows: 64 blocks; found by tests/test_folded_strings.py)
    std::vector<Term> tt(n);
    for (size_t i = 0; i < n; i++) tt[i] = {1, r[i].id()};
    return Ref(e, e->lin(tt.data(), n, 0));
}

// Flags that are the SAME block (one bootstrap shared by several windows of a mostly plaintext string: the engine's
// common-subexpression table hands out one node) count once in an AND / OR -- summed, they would come back as ONE term
// with a coefficient of up to 15, i.e. 225 times the variance of a flag.
// "The same" is structural: a flag may be a linear form over a block (1 - bad of char_significant, 1 - same ...), a new
// node every time it is built, so two nodes are the same flag when they are the same block or the same constant + terms
// (a string that holds one ciphertext several times -- `repeat`, a caller's own FheString -- produces those).
// One canonical key per flag -- the block id, or for a linear form its constant and its sorted (block, coefficient)
// terms -- and an order-preserving de-duplication through a sorted set: O(n log n) in the number of flags (an and_tree
// over 4096 characters compares thousands; pairwise comparison with two sorted temporaries each was O(n^2) allocations
// on the host planning path).
using FlagKey = std::vector<int64_t>;
static FlagKey flag_key(const Engine *e, Bid a) {
    const BlockNode &x = e->node(a);
    if (x.kind != BlockNode::LIN) return {0, (int64_t)a};
    std::vector<std::pair<Bid, int64_t>> tt;
    tt.reserve(x.terms.size());
    for (const Term &t : x.terms) tt.emplace_back(t.blk, t.coef);
    std::sort(tt.begin(), tt.end());
    if (tt.size() == 1 && tt[0].second == 1 && x.konst == 0) return {0, (int64_t)tt[0].first};   // 1 * block: that block
    FlagKey k{1, (int64_t)x.konst};
    for (auto &t : tt) { k.push_back((int64_t)t.first); k.push_back(t.second); }
    return k;
}
static void distinct_flags(const Engine *e, std::vector<Ref> &f) {
    if (f.size() < 2) return;
    std::set<FlagKey> seen;
    std::vector<Ref> out;
    out.reserve(f.size());
    for (Ref &x : f)
        if (seen.insert(flag_key(e, x.id())).second) out.push_back(x);
    f.swap(out);
}

Ref Strings::and_tree(std::vector<Ref> f) {
    std::vector<Ref> cur;
    for (Ref &x : f) {
        if (e_->is_triv(x.id())) {
            if ((e_->triv_val(x.id()) & 1) == 0) return trivial_block(e_, 0);
        } else cur.push_back(x);
    }
    if (cur.empty()) return trivial_block(e_, 1);
    while (cur.size() > 1) {
        distinct_flags(e_, cur);
        if (cur.size() == 1) break;
        std::vector<Ref> nxt;
        for (size_t i = 0; i < cur.size(); i += 15) {
            const size_t n = std::min<size_t>(15, cur.size() - i);
            if (n == 1) nxt.push_back(cur[i]);
            else nxt.push_back(pbs(sum_refs(e_, &cur[i], n), lut_is_k((int)n)));
        }
        cur.swap(nxt);
    }
    return cur[0];
}

Ref Strings::or_tree(std::vector<Ref> f) {
    std::vector<Ref> cur;
    for (Ref &x : f) {
        if (e_->is_triv(x.id())) {
            if (e_->triv_val(x.id()) & 1) return trivial_block(e_, 1);
        } else cur.push_back(x);
    }
    if (cur.empty()) return trivial_block(e_, 0);
    while (cur.size() > 1) {
        distinct_flags(e_, cur);
        if (cur.size() == 1) break;
        std::vector<Ref> nxt;
        for (size_t i = 0; i < cur.size(); i += 15) {
            const size_t n = std::min<size_t>(15, cur.size() - i);
            if (n == 1) nxt.push_back(cur[i]);
            else nxt.push_back(pbs(sum_refs(e_, &cur[i], n), LUT_NZ));
        }
        cur.swap(nxt);
    }
    return cur[0];
}

// OR of flags of which AT MOST ONE is set.  Their sum never exceeds 1, so it is the noise budget, not the 4-bit message
// space, that limits a group: up to FHS_NOISE_BUDGET_SUM_C2 fresh flags per refreshing bootstrap instead of 15, and the
// last few partial sums (sum c^2 <= 4) are handed back as a linear combination without a bootstrap of their own.
// 4097 flags: 65 + 2 bootstraps in 2 levels instead of 274 + 19 + 2 + 1 in 4 (the verdict of a 4096-character `le`).
Ref Strings::onehot_or(std::vector<Ref> f) {
    std::vector<Ref> cur;
    for (Ref &x : f) {
        if (e_->is_triv(x.id())) {
            if (e_->triv_val(x.id()) & 1) return trivial_block(e_, 1);
        } else cur.push_back(x);
    }
    if (cur.empty()) return trivial_block(e_, 0);
    for (;;) {
        if (cur.size() == 1) return cur[0];
        int64_t total = 0;
        for (const Ref &x : cur) total += e_->sum_c2(x.id());
        if (total <= 4) {
            Term tt[4];
            for (size_t i = 0; i < cur.size(); i++) tt[i] = {1, cur[i].id()};
            return Ref(e_, e_->lin(tt, cur.size(), 0));
        }
        std::vector<Ref> nxt;
        Term tt[FHS_NOISE_BUDGET_SUM_C2];
        size_t m = 0;
        int64_t c2 = 0;
        auto close = [&] {
            if (m == 1) nxt.push_back(Ref(e_, (e_->retain(tt[0].blk), tt[0].blk)));
            else if (m > 1) nxt.push_back(pbs(Ref(e_, e_->lin(tt, m, 0)), LUT_NZ));
            m = 0;
            c2 = 0;
        };
        for (const Ref &x : cur) {
            const int64_t w = std::max<int64_t>(1, e_->sum_c2(x.id()));
            if (c2 + w > FHS_NOISE_BUDGET_SUM_C2 || m == (size_t)FHS_NOISE_BUDGET_SUM_C2) close();
            tt[m++] = {1, x.id()};
            c2 += w;
        }
        close();
        cur.swap(nxt);
    }
}

FChar Strings::flags_or(const FStr &flags) {
    std::vector<Ref> f;
    for (const FChar &c : flags) f.push_back(c.b[0]);
    return ch_flag(e_, or_tree(f));
}
FChar Strings::flags_and(const FStr &flags) {
    std::vector<Ref> f;
    for (const FChar &c : flags) f.push_back(c.b[0]);
    return ch_flag(e_, and_tree(f));
}

// Equality of two chars as the AND of 2 flags: nibbles (x0 + 4 x1, x2 + 4 x3) are packed linearly,
// and (a_nib - b_nib) in [-15, 15] is tested with the `is0` LUT.  That LUT is safe under the
// padding-bit rule: a negative difference lands on 32-k, whose PBS value is -f(16-k) = -0 = 0.
std::vector<Ref> Strings::block_eq_flags(const FChar &a_in, const FChar &b_in) {
    // Operands are char blocks (clean 2-bit digits), possibly short sums of bootstrap outputs (a select is SEL_T +
    // SEL_F, a case shift adds 2 x flag, ...).  If the weights 1, 4, -1, -4 would carry more than the noise budget into
    // the bootstrap, the blocks that are sums are refreshed first (LUT_MSG keeps a clean digit).
    FChar a = a_in, b = b_in;
    std::vector<Ref> f;
    for (int h = 0; h < 2; h++) {
        Ref *blk[4] = {&a.b[2 * h], &a.b[2 * h + 1], &b.b[2 * h], &b.b[2 * h + 1]};
        Ref d = lin(e_, {{1, &a.b[2 * h]}, {4, &a.b[2 * h + 1]}, {-1, &b.b[2 * h]}, {-4, &b.b[2 * h + 1]}});
        // measured on the FLATTENED combination: blocks that share bootstrap outputs (common subexpressions) add
        // their coefficients before squaring
        if (e_->sum_c2(d.id()) > FHS_NOISE_BUDGET_SUM_C2) {
            for (int k = 0; k < 4; k++)
                if (e_->sum_c2(blk[k]->id()) > 1) *blk[k] = pbs(*blk[k], LUT_MSG);
            d = lin(e_, {{1, &a.b[2 * h]}, {4, &a.b[2 * h + 1]}, {-1, &b.b[2 * h]}, {-4, &b.b[2 * h + 1]}});
        }
        f.push_back(pbs(d, LUT_IS0));
    }
    return f;
}

Ref Strings::window_match(const FStr &s, size_t at, const FStr &pat) {
    std::vector<Ref> f;
    for (size_t j = 0; j < pat.size(); j++)
        for (Ref &x : block_eq_flags(s[at + j], pat[j])) f.push_back(x);
    return and_tree(f);
}

FChar Strings::f_contains(const FStr &s, const FStr &needle) {
    if (needle.empty()) return t(1);
    std::vector<Ref> w;
    for (size_t i = 0; i + needle.size() <= s.size(); i++) w.push_back(window_match(s, i, needle));
    return ch_flag(e_, or_tree(w));
}

// exclusive prefix OR of flags in log_15 depth: p[i] = OR_{k<i} f[k]
std::vector<Ref> Strings::prefix_or(const std::vector<Ref> &f) {
    const size_t n = f.size();
    std::vector<Ref> p(n);
    if (n == 0) return p;
    const size_t nchunks = (n + 14) / 15;
    std::vector<Ref> q, any;                 // q: exclusive prefix OR over whole chunks, any: the chunks' totals
    if (nchunks > 1) {
        any.resize(nchunks - 1);             // the last chunk's total is never needed
        for (size_t j = 0; j + 1 < nchunks; j++) {
            std::vector<Ref> grp(f.begin() + 15 * j, f.begin() + 15 * j + 15);
            distinct_flags(e_, grp);                              // an OR: a flag that occurs twice counts once (noise)
            any[j] = pbs(sum_refs(e_, grp.data(), grp.size()), LUT_NZ);
        }
        any.push_back(trivial_block(e_, 0));
        q = prefix_or(any);
    } else {
        q.push_back(trivial_block(e_, 0));
    }
    auto depth = [&](const Ref &r) { return e_->node(r.id()).level; };
    for (size_t i = 0; i < n; i++) {
        const size_t j = i / 15, k = i % 15;
        std::vector<Ref> terms(f.begin() + 15 * j, f.begin() + 15 * j + k);
        // the 16th chunk's prefix is the OR of 16 totals, one level deeper than its neighbours': the previous chunk's
        // prefix and total say the same one level earlier (find on 256 characters ends one launch sooner)
        if (j >= 1 && k >= 1 && k <= 13 && depth(q[j]) > std::max(depth(q[j - 1]), depth(any[j - 1]))) {
            terms.push_back(q[j - 1]);
            terms.push_back(any[j - 1]);
            distinct_flags(e_, terms);
            p[i] = pbs(sum_refs(e_, terms.data(), terms.size()), LUT_NZ);
            continue;
        }
        const bool q0 = e_->is_triv(q[j].id()) && e_->triv_val(q[j].id()) == 0;
        if (k == 0) { p[i] = q[j]; continue; }               // nothing of this chunk yet: the flag itself, no bootstrap
        if (k == 1 && q0) { p[i] = f[15 * j]; continue; }    // OR of one 0/1 flag
        terms.push_back(q[j]);
        distinct_flags(e_, terms);               // an OR: the same flag twice counts once (<= 14 + 1 terms)
        p[i] = pbs(sum_refs(e_, terms.data(), terms.size()), LUT_NZ);   // folds to a constant when everything is trivial
    }
    return p;
}

// find (mod.rs:1010-1053) re-associated: window flags, exclusive prefix OR, first = f & !prefix,
// position digits as sums of first_i * digit(i) (at most one term is non-zero), 255 when absent.
FChar Strings::f_find(const FStr &s, const FStr &pat) { return f_find_from_flags(f_find_window_flags(s, pat)); }

// the match flag of every window of `s` (the first two dependency levels of find: nibble tests, AND per window)
std::vector<Ref> Strings::f_find_window_flags(const FStr &s, const FStr &pat) {
    if (s.size() < pat.size()) return {};
    const size_t W = s.size() - pat.size() + 1;
    std::vector<Ref> f(W);
    for (size_t i = 0; i < W; i++) f[i] = pat.empty() ? trivial_block(e_, 1) : window_match(s, i, pat);
    return f;
}

// ... and the rest of find on ALL window flags in string order: index of the first one set, 255 if none.  The sharded
// find exchanges the flags and runs this part on every rank (fhs_dist_str_find).
FChar Strings::f_find_from_flags(const std::vector<Ref> &f) {
    const size_t W = f.size();
    std::vector<Ref> p = prefix_or(f);
    Ref found = or_tree(f);
    if (W <= 256) return first_index(p, found);              // 255 = 3,3,3,3 when absent (:1023)
    std::vector<Ref> first(W);
    for (size_t i = 0; i < W; i++) first[i] = pbs(lin(e_, {{2, &f[i]}, {1, &p[i]}}), LUT_IS2);
    Ref one = trivial_block(e_, 1);
    Ref nf = lin(e_, {{1, &one}, {-1, &found}});
    return position_of(first, 0, &nf, 255);
}

// Index of the first set flag from the exclusive prefix ORs before[j] = OR(f[0 .. j-1]) (before[W] = found): the flags
// g[j] = found - before[j] = [the first match sits at index >= j] form a thermometer, so floor(index / 4^d) = sum_r
// g[r 4^d] and every base-4 digit is LINEAR in the prefix ORs,
//     digit_d = sum_q L_q,   L_q = g[(4q+1) S] + g[(4q+2) S] + g[(4q+3) S] - 3 g[(4q+4) S],   S = 4^d,
// with at most one block q non-zero (`found` cancels inside a block).  The two high digits are used as they are, the
// low ones refresh their blocks first (noise budget); 255 when nothing is set.  ~90 bootstraps in 1-3 levels instead of
// the one-hot route (first = f & !before, weighted sums: ~560 in 6) -- find on 256 characters: 8 levels instead of 11.
FChar Strings::first_index(const std::vector<Ref> &before, const Ref &found) {
    const size_t W = before.size();                          // index in [0, W), W <= 256
    Ref one = trivial_block(e_, 1);
    Ref nf = lin(e_, {{1, &one}, {-1, &found}});
    auto g_terms = [&](size_t j, int64_t coef, std::vector<Term> &tt) {       // coef * g[j], j >= 1
        if (j > W) return;                                   // index >= j is impossible
        tt.push_back({coef, found.id()});
        tt.push_back({-coef, j == W ? found.id() : before[j].id()});
    };
    FChar r;
    for (int d = 0; d < 4; d++) {
        const size_t S = (size_t)1 << (2 * d);
        std::vector<Ref> blocks;                             // L_q, values 0..3, at most one non-zero
        for (size_t q = 0; (4 * q + 1) * S <= W; q++) {
            std::vector<Term> tt;
            for (size_t i = 1; i <= 3; i++) g_terms((4 * q + i) * S, 1, tt);
            g_terms((4 * q + 4) * S, -3, tt);
            blocks.push_back(Ref(e_, e_->lin(tt.data(), tt.size(), 0)));
        }
        // add the blocks up within the noise budget: while the whole sum (+ the `absent` term) would pass it, merge
        // neighbours up to sum c^2 <= 60 and refresh each merged group (one value in 0..3: at most one block is non-zero)
        // (variances are those of the SUMS: blocks of a mostly plaintext string share bootstrap outputs, their
        // coefficients add up inside a sum)
        auto var_of = [&](const std::vector<Ref> &v) {
            return v.empty() ? (int64_t)0 : e_->sum_c2(v.size() == 1 ? v[0].id() : sum_refs(e_, v.data(), v.size()).id());
        };
        bool stalled_before = false;
        for (;;) {
            if (blocks.size() <= 1 || var_of(blocks) + 9 <= 48) break;
            std::vector<Ref> nxt, grp;
            auto close = [&] {
                if (grp.empty()) return;
                nxt.push_back(pbs(grp.size() == 1 ? grp[0] : sum_refs(e_, grp.data(), grp.size()), LUT_MSG));
                grp.clear();
            };
            for (const Ref &b : blocks) {
                grp.push_back(b);
                if (grp.size() > 1 && var_of(grp) > 60) {     // b does not fit: close the group without it
                    grp.pop_back();
                    close();
                    grp.push_back(b);
                }
            }
            close();
            // A pass that merged nothing still REFRESHED every block (each was too noisy to share a group): the fresh
            // blocks merge in the next pass (64 refreshed blocks + 9 = 73 used to leave here un-merged).  Only
            // a second pass in a row without progress -- fresh blocks that are one shared node, say -- ends the loop.
            const bool stalled = nxt.size() >= blocks.size();
            blocks.swap(nxt);
            if (stalled && stalled_before) break;
            stalled_before = stalled;
        }
        Ref digit = blocks.empty() ? trivial_block(e_, 0) : (blocks.size() == 1 ? blocks[0] : sum_refs(e_, blocks.data(), blocks.size()));
        // 255 = 3,3,3,3 when absent.  The digit is a sum of at most 48 + 9 bootstrap-output variances, inside the budget of
        // one more bootstrap, and its value never leaves 0..3 (found: nf = 0; absent: every block is 0): it is handed back
        // as it is -- like a comparison's verdict -- instead of paying one more dependency level for a refresh; consumers
        // that weigh their operands refresh them by the engine's bookkeeping.
        r.b[d] = lin(e_, {{1, &digit}, {3, &nf}});
    }
    return r;
}

// Lexicographic comparison (mod.rs:1470-1541) as a tree of three-state values.  A leaf is the sign of one nibble
// difference (a0 + 4 a1 - b0 - 4 b1 in [-15, 15] through the padding bit: -1 / 0 / 1), most significant first: position 0
// before position 1, the high nibble before the low one, a character one buffer does not have counts as NUL.  Three
// states reduce to one with sign(4 s1 + 2 s2 + s3) -- the first non-zero state decides, sum c^2 = 21 -- so the order of
// 2 n nibbles is known after 1 + log3(2 n) levels and ~3 n bootstraps.  (Rounds 1-2: per pair two signs, `lt` and `==`
// look-ups, a prefix OR over the `differs` flags, one pick per position and an OR over the picks: ~6 n bootstraps.)
// The reference decides by len(a) vs len(b) (numbers of non-zero characters) when no common position differs; with every
// common position equal those differ exactly by the non-zero characters in the longer buffer's tail, which is what the
// tail leaves (tail character against NUL) say.
std::vector<Ref> Strings::cmp_leaves(const FStr &a, const FStr &b) {
    const size_t n = std::max(a.size(), b.size());
    std::vector<Ref> st;
    st.reserve(2 * n);
    for (size_t i = 0; i < n; i++) {
        FChar ca = i < a.size() ? a[i] : t(0), cb = i < b.size() ? b[i] : t(0);
        for (int p = 1; p >= 0; p--) {
            Ref *blk[4] = {&ca.b[2 * p], &ca.b[2 * p + 1], &cb.b[2 * p], &cb.b[2 * p + 1]};
            Ref d = lin(e_, {{1, blk[0]}, {4, blk[1]}, {-1, blk[2]}, {-4, blk[3]}});
            if (e_->sum_c2(d.id()) > FHS_NOISE_BUDGET_SUM_C2) {  // operands that are sums of bootstrap outputs: refresh
                for (int k = 0; k < 4; k++)
                    if (e_->sum_c2(blk[k]->id()) > 1) *blk[k] = pbs(*blk[k], LUT_MSG);
                d = lin(e_, {{1, blk[0]}, {4, blk[1]}, {-1, blk[2]}, {-4, blk[3]}});
            }
            st.push_back(pbs(d, LUT_SIGN));
        }
    }
    return st;
}

Ref Strings::cmp_root_sum(std::vector<Ref> st) {
    while (st.size() > 3) {
        std::vector<Ref> nx;
        nx.reserve(st.size() / 3 + 1);
        for (size_t i = 0; i < st.size(); i += 3) {
            const size_t k = std::min<size_t>(3, st.size() - i);
            if (k == 1) nx.push_back(st[i]);
            else if (k == 2) nx.push_back(pbs(lin(e_, {{2, &st[i]}, {1, &st[i + 1]}}), LUT_SIGN));
            else nx.push_back(pbs(lin(e_, {{4, &st[i]}, {2, &st[i + 1]}, {1, &st[i + 2]}}), LUT_SIGN));
        }
        st.swap(nx);
    }
    if (st.size() == 3) return lin(e_, {{4, &st[0]}, {2, &st[1]}, {1, &st[2]}}, 8);
    if (st.size() == 2) return lin(e_, {{2, &st[0]}, {1, &st[1]}}, 8);
    if (st.size() == 1) return lin(e_, {{1, &st[0]}}, 8);
    return trivial_block(e_, 8);
}

FChar Strings::f_comparison(const FStr &a, const FStr &b, int cmp) {
    static const int root[4] = {LUT_LT8, LUT_LE8, LUT_GT8, LUT_GE8};
    return ch_flag(e_, pbs(cmp_root_sum(cmp_leaves(a, b)), root[cmp & 3]));   // two empty buffers: 8 = equal (:1490-1494)
}

// The positional half of the comparison on two equally long slices: (any position differs, verdict at the FIRST
// differing position; 0 when none differs).  Multi-GPU position sharding combines these per-range partials: the
// first range that differs decides (fhestring_amd/parallel.py ShardedCmp).
void Strings::f_cmp_partial(const FStr &a, const FStr &b, int cmp, FChar *any_diff_out, FChar *verdict_out) {
    const size_t n = std::min(a.size(), b.size());
    if (n == 0) {
        *any_diff_out = t(0);
        *verdict_out = t(0);
        return;
    }
    FStr ca(a.begin(), a.begin() + n), cb(b.begin(), b.begin() + n);
    Ref v = cmp_root_sum(cmp_leaves(ca, cb));
    Ref one = trivial_block(e_, 1), same = pbs(v, LUT_IS8);
    *verdict_out = ch_flag(e_, pbs(v, (cmp == 0 || cmp == 1) ? LUT_LT8 : LUT_GT8));
    *any_diff_out = ch_flag(e_, lin(e_, {{1, &one}, {-1, &same}}));
}

// Combines per-range partials of f_cmp_partial (ranges in string order): the first range that differs decides;
// if none differs the result is `tie` (1 for le / ge, 0 for lt / gt: equal buffers are equal strings).
FChar Strings::flags_first_decides(const FStr &any_diff, const FStr &verdict, int tie) {
    const size_t n = std::min(any_diff.size(), verdict.size());
    if (n == 0) return t(tie ? 1 : 0);
    std::vector<Ref> d(n), pick(n);
    for (size_t r = 0; r < n; r++) d[r] = any_diff[r].b[0];
    std::vector<Ref> before = prefix_or(d);                  // exclusive: some earlier range differs
    for (size_t r = 0; r < n; r++) pick[r] = pbs(lin(e_, {{2, &verdict[r].b[0]}, {1, &before[r]}}), LUT_IS2);
    Ref ret = onehot_or(pick);                               // at most one pick is set
    if (tie) {                                               // ret = 1 implies a difference: the sum stays in {0, 1}
        Ref one = trivial_block(e_, 1), any = or_tree(d);
        ret = lin(e_, {{1, &ret}, {1, &one}, {-1, &any}});
    }
    return ch_flag(e_, ret);
}

std::vector<Ref> Strings::suffix_or(const std::vector<Ref> &f) {
    std::vector<Ref> rev(f.rbegin(), f.rend());
    std::vector<Ref> p = prefix_or(rev);
    return std::vector<Ref>(p.rbegin(), p.rend());
}

Ref Strings::char_nonzero(const FChar &c) { return char_zero_test(c, false); }

// [c == 0] or [c != 0] in ONE bootstrap: the four base-4 digits are non-negative, so their sum (<= 12, inside the message
// space) is zero iff every digit is -- instead of two nibble tests and a combining bootstrap.  Digits that are sums of
// bootstrap outputs are refreshed first only if together they would leave the noise budget.
Ref Strings::char_zero_test(const FChar &c_in, bool want_zero) {
    FChar c = c_in;
    // measured on the FLATTENED sum, like block_eq_flags: the four digits of a selected character share the select's
    // flag outputs (sel / covered of f_replace_expand: one bootstrap output in all four digits), whose coefficients add
    // up BEFORE squaring -- per-block sums under-count that (120 for replace with an 8-character `from`)
    Ref d = lin(e_, {{1, &c.b[0]}, {1, &c.b[1]}, {1, &c.b[2]}, {1, &c.b[3]}});
    if (e_->sum_c2(d.id()) > FHS_NOISE_BUDGET_SUM_C2) {
        for (int k = 0; k < 4; k++)
            if (e_->sum_c2(c.b[k].id()) > 1) c.b[k] = pbs(c.b[k], LUT_MSG);
        d = lin(e_, {{1, &c.b[0]}, {1, &c.b[1]}, {1, &c.b[2]}, {1, &c.b[3]}});
    }
    return pbs(d, want_zero ? LUT_IS0 : LUT_NZ);
}

// NUL = 0x00, whitespace = 0x20, 0x09..0x0D (fheasciichar.rs:106-130): high nibble 0 with low nibble in
// {0, 9..13}, or high nibble 2 with low nibble 0 -- same 3-bootstrap shape as the case detector (row class of the high
// nibble, two range flags of the low nibble, the pick of the flag that belongs to the row)
Ref Strings::char_significant(const FChar &c) {
    Ref lo = lin(e_, {{1, &c.b[0]}, {4, &c.b[1]}});
    Ref hi = lin(e_, {{1, &c.b[2]}, {4, &c.b[3]}});
    Ref row = pbs(hi, LUT_HI_ROW02), cls = pbs(lo, LUT_LO_WSCLS);
    Ref bad = pbs(lin(e_, {{1, &row}, {4, &cls}}), LUT_CLS_PICK);
    Ref one = trivial_block(e_, 1);
    return lin(e_, {{1, &one}, {-1, &bad}});
}

// u8 position encoded by one-hot flags: sum_i pick_i * (i + offset) digit-wise (at most one pick is set);
// when `absent_flag` is set the result is `absent_value` (255 for find/rfind, mod.rs:1023).
// Noise: a sum of k bootstrap outputs with weights c carries sum c^2 output variances into the next bootstrap, so
// the weighted picks are grouped by WEIGHT (sum of digit^2 <= FHS_NOISE_BUDGET_SUM_C2, not by count), every group is
// refreshed with LUT_MSG before groups are added up, and the digits handed back to the caller are refreshed too:
// like every op of the reference (fheasciichar.rs:35-104) the result is a fresh, clean-carry ciphertext.
FChar Strings::position_of(const std::vector<Ref> &pick, size_t off, const Ref *absent_flag, int absent_value) {
    const size_t W = pick.size();
    FChar r;
    // At most one pick is set.  A block that serves as the pick of TWO positions (mostly plaintext strings: one shared
    // bootstrap) can therefore never be set and is left out -- summed, its digits would add up to one large coefficient.
    std::vector<bool> twice(W, false);
    {
        std::vector<std::pair<Bid, size_t>> ids;
        ids.reserve(W);
        for (size_t i = 0; i < W; i++)
            if (!e_->is_triv(pick[i].id())) ids.push_back({pick[i].id(), i});
        std::sort(ids.begin(), ids.end());
        for (size_t a = 0; a + 1 < ids.size(); a++)
            if (ids[a].first == ids[a + 1].first) twice[ids[a].second] = twice[ids[a + 1].second] = true;
    }
    for (int blk = 0; blk < 4; blk++) {
        std::vector<Ref> cur;
        {
            Term tt[64];
            size_t m = 0;
            int64_t c2 = 0;
            // a lone group also takes the `absent` term (weight <= 3) before its refresh
            const int64_t limit = FHS_NOISE_BUDGET_SUM_C2 - (absent_flag ? 9 : 0);
            auto close = [&] {
                if (m) cur.push_back(Ref(e_, e_->lin(tt, m, 0)));
                m = 0;
                c2 = 0;
            };
            for (size_t i = 0; i < W; i++) {
                const int dig = (int)(((i + off) >> (2 * blk)) & 3);
                if (twice[i]) continue;
                if (!dig || e_->is_triv(pick[i].id())) {
                    if (dig && e_->triv_val(pick[i].id())) {                                    // folds into the constant
                        if (m == 64) close();               // tt is full: never write tt[64] (either branch may fill it)
                        tt[m++] = {dig, pick[i].id()};
                    }
                    continue;
                }
                if (c2 + dig * dig > limit || m == 64) close();
                tt[m++] = {dig, pick[i].id()};
                c2 += dig * dig;
            }
            close();
        }
        // tree of refreshed partial sums: <= 15 fresh digits (one-hot: the value stays <= 3) per node
        while (cur.size() > 1) {
            std::vector<Ref> nxt;
            for (size_t g = 0; g < cur.size(); g += 15) {
                std::vector<Ref> fresh;
                for (size_t i = g; i < std::min(cur.size(), g + 15); i++) fresh.push_back(pbs(cur[i], LUT_MSG));
                nxt.push_back(fresh.size() == 1 ? fresh[0] : sum_refs(e_, fresh.data(), fresh.size()));
            }
            cur.swap(nxt);
        }
        Ref digit = cur.empty() ? trivial_block(e_, 0) : cur[0];
        if (absent_flag) digit = lin(e_, {{1, &digit}, {(absent_value >> (2 * blk)) & 3, absent_flag}});
        r.b[blk] = pbs(digit, LUT_MSG);                      // folds when everything was trivial
    }
    return r;
}

// rfind (mod.rs:727-790) re-associated: the LAST matching window wins (the loop ascends and overwrites);
// `s` already carries the NUL pushed at :737
FChar Strings::f_rfind(const FStr &s, const FStr &pat) {
    if (pat.empty()) {                                      // :747-760: index after the last non-NUL char
        std::vector<Ref> nz(s.size());
        for (size_t i = 0; i < s.size(); i++) nz[i] = char_nonzero(s[i]);
        std::vector<Ref> after = suffix_or(nz);
        std::vector<Ref> last(s.size());
        for (size_t i = 0; i < s.size(); i++) last[i] = pbs(lin(e_, {{2, &nz[i]}, {1, &after[i]}}), LUT_IS2);
        return position_of(last, 1, nullptr, 0);
    }
    if (pat.size() > s.size()) return t(255);
    const size_t E = std::max<size_t>(1, s.size() - pat.size());   // adjust_end_of_pattern, exclusive bound (:768-771)
    std::vector<Ref> f(E);
    for (size_t i = 0; i < E; i++) f[i] = window_match(s, i, pat);
    std::vector<Ref> after = suffix_or(f);
    std::vector<Ref> last(E);
    for (size_t i = 0; i < E; i++) last[i] = pbs(lin(e_, {{2, &f[i]}, {1, &after[i]}}), LUT_IS2);
    Ref found = or_tree(f);
    Ref one = trivial_block(e_, 1);
    Ref nf = lin(e_, {{1, &one}, {-1, &found}});
    return position_of(last, 0, &nf, 255);
}

// ends_with (mod.rs:241-288) re-associated: the result is the match flag of the LAST window that
// contains no NUL ("use the last result that has not encountered padding", :281-286), 0 if none.
FChar Strings::f_ends_with(const FStr &s, const FStr &needle, std::vector<Ref> *pick_out) {
    const size_t W = s.size() - needle.size() + 1;
    std::vector<Ref> nzc(s.size());
    for (size_t k = 0; k < s.size(); k++) nzc[k] = char_nonzero(s[k]);
    std::vector<Ref> valid(W), match(W);
    for (size_t i = 0; i < W; i++) {
        std::vector<Ref> v(nzc.begin() + i, nzc.begin() + i + needle.size());
        valid[i] = and_tree(v);
        match[i] = needle.empty() ? trivial_block(e_, 1) : window_match(s, i, needle);
    }
    std::vector<Ref> after = suffix_or(valid);
    std::vector<Ref> pick(W);
    for (size_t i = 0; i < W; i++)                           // match & valid & !after
        pick[i] = pbs(lin(e_, {{1, &match[i]}, {1, &valid[i]}, {4, &after[i]}}), LUT_IS2);
    Ref res = or_tree(pick);
    if (pick_out) *pick_out = pick;
    return ch_flag(e_, res);
}

// replace with |from| < |to| (handle_shorter_from, mod.rs:885-980) re-associated.  The reference scans
// left to right, splices `to` in place and masks the inserted text, i.e. it replaces the greedy
// leftmost NON-overlapping matches.  Here: match flags on the original string; a countdown state
// machine (1 block, 2 PBS per position) picks the greedy matches; every position then emits a slot of
// |to| characters (`to` at a selected start, nothing inside a match, itself otherwise) and the slots
// are compacted.  Same plaintext as the reference; the buffer is (n+1)*|to| chars instead of
// |to|*(n+1) + (n+1).
FStr Strings::f_replace_expand(const FStr &s_in, const FStr &from, const FStr &to) {
    const FChar zero = t(0);
    FStr s = s_in;
    s.push_back(zero);                                       // :898
    const size_t n = s.size(), m = from.size(), L = to.size();
    std::vector<Ref> sel(n), covered(n), keep_flag(n);
    Ref state = trivial_block(e_, 0);                        // positions still blocked by the last match
    // 1 - sel - covered is a sum of m outputs and enters the select below with weight 4: 16 m + 1 > 64 from m = 4 on.
    // The same flag is [countdown == 0 and no match here] = [v == 0], one more look-up on the state machine's own input
    // (same dependency level, sum c^2 = 5), used from the pattern length on where the linear form leaves the budget.
    const bool keep_by_lut = 16 * (int64_t)m + 1 > FHS_NOISE_BUDGET_SUM_C2;
    for (size_t i = 0; i < n; i++) {
        Ref f = i + m <= n ? window_match(s, i, from) : trivial_block(e_, 0);
        Ref v = lin(e_, {{2, &state}, {1, &f}});
        sel[i] = pbs(v, LUT_GREEDY_SEL);
        if (keep_by_lut) keep_flag[i] = pbs(v, LUT_IS0);
        state = pbs(v, LUT_GREEDY_NEXT0 + (int)(m - 1));     // selected ? m - 1 : max(countdown - 1, 0)
    }
    for (size_t i = 0; i < n; i++) {
        Term tt[8];
        size_t k = 0;
        for (size_t d = 1; d < m && d <= i; d++) tt[k++] = {1, sel[i - d].id()};
        covered[i] = Ref(e_, e_->lin(tt, k, 0));             // inside a selected match (not its start)
    }
    Ref one = trivial_block(e_, 1);
    FStr slots;
    slots.reserve(n * L);
    for (size_t i = 0; i < n; i++) {
        Ref keep = keep_by_lut ? keep_flag[i] : lin(e_, {{1, &one}, {-1, &sel[i]}, {-1, &covered[i]}});
        for (size_t j = 0; j < L; j++) {
            FChar c;
            for (int b = 0; b < 4; b++) {
                Ref a = pbs(lin(e_, {{4, &sel[i]}, {1, &to[j].b[b]}}), LUT_SEL_T);
                if (j == 0) {
                    Ref o = pbs(lin(e_, {{4, &keep}, {1, &s[i].b[b]}}), LUT_SEL_T);
                    c.b[b] = lin(e_, {{1, &a}, {1, &o}});
                } else c.b[b] = a;
            }
            slots.push_back(c);
        }
    }
    return f_compact(slots);
}

// trim_end / the marking half of trim_start (trim.rs:36-57, 86-115): a char survives iff some
// non-NUL, non-whitespace char sits at or after it (from_end) / at or before it
FStr Strings::f_trim(const FStr &s, bool from_end) {
    const size_t n = s.size();
    std::vector<Ref> sig(n);
    for (size_t i = 0; i < n; i++) sig[i] = char_significant(s[i]);
    std::vector<Ref> other = from_end ? suffix_or(sig) : prefix_or(sig);
    const FChar zero = t(0);
    FStr r(n);
    for (size_t i = 0; i < n; i++) {
        Ref stop = pbs(lin(e_, {{1, &sig[i]}, {1, &other[i]}}), LUT_NZ);   // inclusive OR
        r[i] = ite_flag(stop, s[i], zero);
    }
    return r;
}

// cond ? t : f with cond a clean single-block 0/1 flag (no scalar_ne needed)
FChar Strings::ite_flag(const Ref &flag_in, const FChar &tv_in, const FChar &fv_in) {
    FChar r;
    // the flag enters with weight 4: a flag that is itself a sum of several bootstrap outputs (one-hot cover sums,
    // 1 - x forms) is refreshed first when 16 x its sum c^2 would leave the noise budget
    Ref flag = flag_in;
    if (16 * e_->sum_c2(flag.id()) + 4 > FHS_NOISE_BUDGET_SUM_C2) flag = pbs(flag_in, LUT_NZ);
    const int64_t room = FHS_NOISE_BUDGET_SUM_C2 - 16 * e_->sum_c2(flag.id());
    FChar tv = tv_in, fv = fv_in;
    for (int i = 0; i < 4; i++) {                           // the selected digits enter with weight 1
        if (e_->sum_c2(tv.b[i].id()) > room) tv.b[i] = pbs(tv.b[i], LUT_MSG);
        if (e_->sum_c2(fv.b[i].id()) > room) fv.b[i] = pbs(fv.b[i], LUT_MSG);
    }
    for (int i = 0; i < 4; i++) {
        Ref a = pbs(lin(e_, {{4, &flag}, {1, &tv.b[i]}}), LUT_SEL_T);
        Ref b = pbs(lin(e_, {{4, &flag}, {1, &fv.b[i]}}), LUT_SEL_F);
        r.b[i] = lin(e_, {{1, &a}, {1, &b}});
    }
    return r;
}

// sum of up to 4 base-4 numbers (+ carry) with a sequential carry chain; result has `digits` digits
Strings::Num Strings::num_add(const std::vector<const Num *> &ops, size_t digits) {
    Num r(digits);
    Ref carry;
    for (size_t d = 0; d < digits; d++) {
        Term tt[8];
        size_t k = 0;
        for (const Num *o : ops)
            if (d < o->size()) tt[k++] = {1, (*o)[d].id()};
        if (carry) tt[k++] = {1, carry.id()};
        Ref sm(e_, e_->lin(tt, k, 0));                      // <= 4*3 + 3
        r[d] = pbs(sm, LUT_MSG);
        if (d + 1 < digits) carry = pbs(sm, LUT_CARRY);
    }
    return r;
}

// exclusive prefix sums of numbers, 4-ary recursion (depth log4(n) x 2 adds)
std::vector<Strings::Num> Strings::num_exclusive_scan(const std::vector<Num> &x, size_t digits) {
    const size_t n = x.size();
    std::vector<Num> out(n);
    Num zero(digits);
    for (auto &d : zero) d = trivial_block(e_, 0);
    if (n == 0) return out;
    if (n == 1) { out[0] = zero; return out; }
    const size_t ng = (n + 3) / 4;
    std::vector<Num> local(n), totals(ng);
    for (size_t g = 0; g < ng; g++) {
        const size_t lo = 4 * g, hi = std::min(n, lo + 4);
        std::vector<const Num *> ops;
        for (size_t i = lo; i < hi; i++) {
            local[i] = ops.empty() ? zero : (ops.size() == 1 ? *ops[0] : num_add(ops, digits));
            ops.push_back(&x[i]);
        }
        totals[g] = ops.size() == 1 ? *ops[0] : num_add(ops, digits);
    }
    std::vector<Num> offs = num_exclusive_scan(totals, digits);
    for (size_t i = 0; i < n; i++) out[i] = num_add({&offs[i / 4], &local[i]}, digits);
    return out;
}

// The number of set flags among k <= 15 as a base-4 number (low digit, high digit, zeros): two look-ups on their sum.
// Flags that are ONE block (the same character twice in a string: `repeat`; NUL tests of identical neighbourhoods of a
// mostly plaintext string, shared by the common-subexpression table) come back from the sum as one term with their
// multiplicity as coefficient -- a count of k equal flags IS k times the flag -- and 9 of them already pass the noise
// budget (81 > 64).  Then the flags are counted in two halves and the halves added as numbers.
Strings::Num Strings::count_digits(const Ref *flags, size_t k, size_t D) {
    Ref sum = k ? sum_refs(e_, flags, k) : trivial_block(e_, 0);
    if (k > 1 && e_->sum_c2(sum.id()) > FHS_NOISE_BUDGET_SUM_C2) {
        Num a = count_digits(flags, k / 2, D), b = count_digits(flags + k / 2, k - k / 2, D);
        return num_add({&a, &b}, D);
    }
    Num v(D);
    for (size_t d = 0; d < D; d++) v[d] = d == 0 ? pbs(sum, LUT_MSG) : (d == 1 ? pbs(sum, LUT_CARRY) : trivial_block(e_, 0));
    return v;
}

// Exclusive prefix counts of 0/1 flags as base-4 numbers: chunks of 15 flags give a (low, high) digit pair each, an
// exclusive scan over the chunk totals (4-ary, log depth) and one add per position.
std::vector<Strings::Num> Strings::flag_prefix_counts(const std::vector<Ref> &z, size_t D) {
    const size_t n = z.size();
    std::vector<Num> out(n);
    if (n == 0) return out;
    const size_t nch = (n + 14) / 15;
    std::vector<Num> tot(nch);
    for (size_t j = 0; j < nch; j++) tot[j] = count_digits(&z[15 * j], std::min<size_t>(15, n - 15 * j), D);
    std::vector<Num> offs = num_exclusive_scan(tot, D);
    for (size_t i = 0; i < n; i++) {
        const size_t j = i / 15, k = i % 15;
        Num loc = count_digits(&z[15 * j], k, D);
        out[i] = num_add({&offs[j], &loc}, D);
    }
    return out;
}

// Oblivious order-preserving compaction: non-NUL characters move left by the number of NULs before
// them.  Shifts are prefix counts (base-4 numbers), routed LSB first through log2(n) conditional
// moves by 2^k -- collision-free for monotone compaction.  Same result as the reference's n-pass
// bubble (utils.rs:28-46) in O(n log n) PBS and O(log n) wide levels instead of O(n^2) / O(n).
// The not-yet-used part of every shift travels with its character as base-4 DIGITS, and the bit a stage needs is
// extracted from its digit where it is used (1 bootstrap).  A stage costs ONE bootstrap per block or digit: the part
// that moves is m = (bit ? x : 0), and what stays is x - m -- linear, exact (m is x or 0) -- so
//     next[p] = cur[p] - m[p] + m[p + 2^k].
// The blocks are sums of 1 + 2k bootstrap outputs by then (sum c^2 <= 23, the select's input 16 + 21), inside the noise
// budget; the result is refreshed once at the end.  101 bootstraps per position for n = 1025 instead of 209 (bits
// routed separately, stay and move selected separately).
FStr Strings::f_compact(const FStr &s) {
    const size_t n = s.size();
    if (n <= 1) return s;
    int K = 0;
    while (((size_t)1 << K) < n) K++;                       // shifts are < n
    const size_t D = (size_t)(K + 1) / 2;
    const FChar zero = t(0);
    std::vector<Ref> z(n);
    for (size_t i = 0; i < n; i++) z[i] = char_zero_test(s[i], true);   // is-NUL flag: one bootstrap on the digit sum
    // per-position shift = number of NULs before it; NUL positions get shift 0 (they stay and contribute zeros)
    std::vector<Num> shifts = flag_prefix_counts(z, D);
    std::vector<std::vector<Ref>> dg(n, std::vector<Ref>(D));
    for (size_t i = 0; i < n; i++)
        for (size_t q = 0; q < D; q++) {
            if (((size_t)1 << (2 * q)) > i) { dg[i][q] = trivial_block(e_, 0); continue; }   // shift <= i < 4^q
            dg[i][q] = pbs(lin(e_, {{1, &shifts[i][q]}, {4, &z[i]}}), LUT_SEL_F);               // digit unless NUL
        }
    FStr cur = s;
    for (int k = 0; k < K; k++) {
        const size_t d = (size_t)1 << k, q = (size_t)k / 2;
        const bool hi = k & 1;
        // the stage's move flag: bit k of the shift, from its digit; statically 0 where the shift cannot reach 2^k
        std::vector<Ref> mv(n);
        std::vector<bool> moves(n);
        for (size_t p = 0; p < n; p++) {
            if (d > p || (e_->is_triv(dg[p][q].id()) && ((e_->triv_val(dg[p][q].id()) >> (hi ? 1 : 0)) & 1) == 0))
                mv[p] = trivial_block(e_, 0);
            else mv[p] = pbs(dg[p][q], hi ? LUT_BIT1_UNLESS : LUT_BIT0_UNLESS);
            moves[p] = !(e_->is_triv(mv[p].id()) && e_->triv_val(mv[p].id()) == 0);
        }
        // digits still needed after this stage: those holding a bit above k that exists (bits 0 .. K-1)
        size_t q0 = hi ? q + 1 : q;
        if ((size_t)k + 1 >= (size_t)K) q0 = D;              // last stage: nothing travels on
        // the moving part of every block and digit
        FStr md(n);
        std::vector<std::vector<Ref>> mdg(n, std::vector<Ref>(D));
        for (size_t p = 0; p < n; p++) {
            if (!moves[p]) continue;
            for (int blk = 0; blk < 4; blk++) md[p].b[blk] = pbs(lin(e_, {{4, &mv[p]}, {1, &cur[p].b[blk]}}), LUT_SEL_T);
            for (size_t jq = q0; jq < D; jq++) mdg[p][jq] = pbs(lin(e_, {{4, &mv[p]}, {1, &dg[p][jq]}}), LUT_SEL_T);
        }
        FStr nxt(n);
        std::vector<std::vector<Ref>> nd(n, std::vector<Ref>(D));
        for (size_t p = 0; p < n; p++) {
            const bool in = p + d < n && moves[p + d];
            auto combine = [&](const Ref &here, const Ref *out, const Ref *inc) {
                Term tt[3];
                size_t m = 0;
                tt[m++] = {1, here.id()};
                if (out) tt[m++] = {-1, out->id()};
                if (inc) tt[m++] = {1, inc->id()};
                return m == 1 ? here : Ref(e_, e_->lin(tt, m, 0));
            };
            for (int blk = 0; blk < 4; blk++)
                nxt[p].b[blk] = combine(cur[p].b[blk], moves[p] ? &md[p].b[blk] : nullptr, in ? &md[p + d].b[blk] : nullptr);
            for (size_t jq = 0; jq < D; jq++) {
                if (jq < q0) { nd[p][jq] = trivial_block(e_, 0); continue; }
                nd[p][jq] = combine(dg[p][jq], moves[p] ? &mdg[p][jq] : nullptr, in ? &mdg[p + d][jq] : nullptr);
            }
        }
        cur.swap(nxt);
        dg.swap(nd);
    }
    for (size_t p = 0; p < n; p++)                           // hand fresh ciphertexts to whatever comes next
        for (int blk = 0; blk < 4; blk++)
            if (e_->sum_c2(cur[p].b[blk].id()) > 1) cur[p].b[blk] = pbs(cur[p].b[blk], LUT_MSG);
    return cur;
}

// sum of 0/1 flags mod 256: groups of 15 -> (low, high) digit pair, then 4-operand radix adds
FChar Strings::count_flags(std::vector<Ref> flags) {
    std::vector<FChar> nums;
    for (size_t i = 0; i < flags.size(); i += 15) {
        const size_t n = std::min<size_t>(15, flags.size() - i);
        Ref s = sum_refs(e_, &flags[i], n);
        FChar c;
        if (e_->sum_c2(s.id()) > FHS_NOISE_BUDGET_SUM_C2) {  // the same flag many times over (`repeat`): count in halves
            Num v = count_digits(&flags[i], n, 2);
            c.b[0] = v[0];
            c.b[1] = v[1];
        } else {
            c.b[0] = pbs(s, LUT_MSG);
            c.b[1] = n >= 4 ? pbs(s, LUT_CARRY) : trivial_block(e_, 0);
        }
        c.b[2] = trivial_block(e_, 0);
        c.b[3] = trivial_block(e_, 0);
        nums.push_back(c);
    }
    if (nums.empty()) return t(0);
    while (nums.size() > 1) {
        std::vector<FChar> nxt;
        for (size_t i = 0; i < nums.size(); i += 4) {
            const size_t n = std::min<size_t>(4, nums.size() - i);
            if (n == 1) { nxt.push_back(nums[i]); continue; }
            FChar r;
            Ref carry;
            for (int blk = 0; blk < 4; blk++) {
                Term tt[8];
                size_t k = 0;
                for (size_t u = 0; u < n; u++) tt[k++] = {1, nums[i + u].b[blk].id()};
                if (carry) tt[k++] = {1, carry.id()};
                Ref sm(e_, e_->lin(tt, k, 0));   // <= 4*3 + 3 = 15
                r.b[blk] = pbs(sm, LUT_MSG);
                if (blk < 3) carry = pbs(sm, LUT_CARRY);
            }
            nxt.push_back(r);
        }
        nums.swap(nxt);
    }
    return nums[0];
}

FChar Strings::f_len(const FStr &s) {
    std::vector<Ref> nz;
    for (const FChar &c : s) nz.push_back(char_zero_test(c, false));
    return count_flags(nz);
}

FChar Strings::f_eq(const FStr &a, const FStr &b) {
    // (both zero) or equal == equal, so the per-position test of mod.rs:1137-1146 is a plain equality
    std::vector<Ref> f;
    const size_t common = std::min(a.size(), b.size());
    for (size_t i = 0; i < common; i++)
        for (Ref &x : block_eq_flags(a[i], b[i])) f.push_back(x);
    // len(a) == len(b) (mod.rs:1133-1135,1148; len = number of non-zero characters): once every common position holds
    // equal characters the two counts differ exactly by the non-zero characters of the longer buffer's tail, so the
    // condition is "that tail is all zero" -- one digit-sum test per extra character instead of two 8-bit popcounts with
    // their carry chains (eq_ignore_case on 4096 characters: 19 levels / 68 541 bootstraps before)
    const FStr &longer = a.size() > b.size() ? a : b;
    for (size_t i = common; i < longer.size(); i++) f.push_back(char_zero_test(longer[i], true));
    return ch_flag(e_, and_tree(f));
}

// 'A'..'Z' = 0x41..0x5A (high nibble 4: low in 1..15; high nibble 5: low in 0..10); 'a'..'z' = +0x20
Ref Strings::is_upper_flag(const FChar &c, bool lower) {
    Ref lo = lin(e_, {{1, &c.b[0]}, {4, &c.b[1]}});
    Ref hi = lin(e_, {{1, &c.b[2]}, {4, &c.b[3]}});
    Ref row = pbs(hi, lower ? LUT_HI_ROW67 : LUT_HI_ROW45);
    Ref cls = pbs(lo, LUT_EQIC_LO);                          // (low >= 1) + 2 (low <= 10)
    return pbs(lin(e_, {{1, &row}, {4, &cls}}), LUT_CLS_PICK);
}

FStr Strings::f_case(const FStr &s, bool to_lower) {
    FStr r;
    for (const FChar &c : s) {
        Ref f = is_upper_flag(c, /*lower=*/!to_lower);
        FChar o = c;
        o.b[2] = lin(e_, {{1, &c.b[2]}, {to_lower ? 2 : -2, &f}});
        r.push_back(o);
    }
    return r;
}

}  // namespace fhs
