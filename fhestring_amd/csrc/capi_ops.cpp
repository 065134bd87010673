// extern "C" boundary, part 2: FheAsciiChar ops, MyServerKey string methods, statistics.
#include <algorithm>

#include "capi_internal.h"

using namespace fhs;

namespace {

FChar load(Engine &e, fhs_char_t h) {
    FChar c;
    const Bid *b = e.char_blocks(h);
    for (int i = 0; i < 4; i++) {
        e.retain(b[i]);
        c.b[i] = Ref(&e, b[i]);
        // results handed back as sums of bootstrap outputs (find's index digits: up to 57 variances) are refreshed when
        // they come back in as operands; verdict-like sums (<= 4) pass, every operator budgets for those
        if (e.sum_c2(b[i]) > 4) c.b[i] = pbs(c.b[i], LUT_MSG);
    }
    return c;
}
fhs_char_t store(Engine &e, FChar &c) {
    Bid b[4];
    for (int i = 0; i < 4; i++) b[i] = c.b[i].detach();
    return e.new_char(b);
}
bool ok(fhs_ctx *c, fhs_char_t h) { return c && c->eng.valid_char(h); }
bool ok_all(fhs_ctx *c, const fhs_char_t *h, size_t n) {
    if (!c || (n && !h)) return false;
    for (size_t i = 0; i < n; i++)
        if (!c->eng.valid_char(h[i])) return false;
    return true;
}
FStr load_str(Engine &e, const fhs_char_t *h, size_t n) {
    FStr s;
    s.reserve(n);
    for (size_t i = 0; i < n; i++) s.push_back(load(e, h[i]));
    return s;
}
void store_str(Engine &e, FStr &s, fhs_char_t *out) {
    for (size_t i = 0; i < s.size(); i++) out[i] = store(e, s[i]);
}
int bad(fhs_ctx *c) { return c ? c->eng.ctx.fail(FHS_ERR_ARG, "invalid handle or null argument") : FHS_ERR_ARG; }

template <class F> fhs_char_t binop(fhs_ctx *c, fhs_char_t a, fhs_char_t b, F f) {
    if (!ok(c, a) || !ok(c, b)) { bad(c); return 0; }
    FChar r = f(load(c->eng, a), load(c->eng, b));
    return store(c->eng, r);
}
template <class F> fhs_char_t unop(fhs_ctx *c, fhs_char_t a, F f) {
    if (!ok(c, a)) { bad(c); return 0; }
    FChar r = f(load(c->eng, a));
    return store(c->eng, r);
}
int finish(fhs_ctx *c, Strings &S) {
    if (S.err.code) return c->eng.ctx.fail(S.err.code, S.err.msg);
    return FHS_OK;
}

}  // namespace

extern "C" {

fhs_char_t fhs_trivial(fhs_ctx *c, uint8_t v) {
    if (!c) return 0;
    FChar r = ch_trivial(&c->eng, v);
    return store(c->eng, r);
}
fhs_char_t fhs_upload(fhs_ctx *c, const uint64_t *blocks) {
    fhs_char_t h = 0;
    return fhs_upload_string(c, blocks, 1, &h) == FHS_OK ? h : 0;
}
int fhs_upload_string(fhs_ctx *c, const uint64_t *blocks, size_t n, fhs_char_t *out) {
    if (!c || (n && (!blocks || !out))) return bad(c);
    if (!c->eng.planner && hipSetDevice(c->eng.ctx.device) != hipSuccess) return c->eng.ctx.fail(FHS_ERR_HIP, "hipSetDevice failed");
    std::vector<Bid> b(4 * n);
    if (c->eng.from_host_many(blocks, 4 * n, b.data())) return c->eng.ctx.fail(FHS_ERR_HIP, "upload failed (device allocation or copy)");
    for (size_t i = 0; i < n; i++) out[i] = c->eng.new_char(&b[4 * i]);
    return FHS_OK;
}
fhs_char_t fhs_import_device(fhs_ctx *c, const uint64_t *d_blocks) {
    if (!c || !d_blocks) { bad(c); return 0; }
    Bid b[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        b[i] = c->eng.from_device(d_blocks + (size_t)i * FHS_BIG_CT);
        if (!b[i]) {
            for (int k = 0; k < i; k++) c->eng.release(b[k]);
            c->eng.ctx.fail(FHS_ERR_HIP, "import failed");
            return 0;
        }
    }
    return c->eng.new_char(b);
}
fhs_char_t fhs_eq(fhs_ctx *c, fhs_char_t a, fhs_char_t b) { return binop(c, a, b, ch_eq); }
fhs_char_t fhs_ne(fhs_ctx *c, fhs_char_t a, fhs_char_t b) { return binop(c, a, b, ch_ne); }
fhs_char_t fhs_le(fhs_ctx *c, fhs_char_t a, fhs_char_t b) { return binop(c, a, b, ch_le); }
fhs_char_t fhs_lt(fhs_ctx *c, fhs_char_t a, fhs_char_t b) { return binop(c, a, b, ch_lt); }
fhs_char_t fhs_ge(fhs_ctx *c, fhs_char_t a, fhs_char_t b) { return binop(c, a, b, ch_ge); }
fhs_char_t fhs_gt(fhs_ctx *c, fhs_char_t a, fhs_char_t b) { return binop(c, a, b, ch_gt); }
fhs_char_t fhs_bitand(fhs_ctx *c, fhs_char_t a, fhs_char_t b) { return binop(c, a, b, ch_bitand); }
fhs_char_t fhs_bitor(fhs_ctx *c, fhs_char_t a, fhs_char_t b) { return binop(c, a, b, ch_bitor); }
fhs_char_t fhs_sub(fhs_ctx *c, fhs_char_t a, fhs_char_t b) { return binop(c, a, b, ch_sub); }
fhs_char_t fhs_add(fhs_ctx *c, fhs_char_t a, fhs_char_t b) { return binop(c, a, b, ch_add); }
fhs_char_t fhs_if_then_else(fhs_ctx *c, fhs_char_t cond, fhs_char_t t, fhs_char_t f) {
    if (!ok(c, cond) || !ok(c, t) || !ok(c, f)) { bad(c); return 0; }
    FChar r = ch_ite(load(c->eng, cond), load(c->eng, t), load(c->eng, f));
    return store(c->eng, r);
}
fhs_char_t fhs_flip(fhs_ctx *c, fhs_char_t a) { return unop(c, a, ch_flip); }
fhs_char_t fhs_is_whitespace(fhs_ctx *c, fhs_char_t a) { return unop(c, a, ch_is_whitespace); }
fhs_char_t fhs_is_uppercase(fhs_ctx *c, fhs_char_t a) { return unop(c, a, ch_is_uppercase); }
fhs_char_t fhs_is_lowercase(fhs_ctx *c, fhs_char_t a) { return unop(c, a, ch_is_lowercase); }
fhs_char_t fhs_clone(fhs_ctx *c, fhs_char_t a) {
    if (!ok(c, a)) { bad(c); return 0; }
    FChar r = load(c->eng, a);
    return store(c->eng, r);
}
int fhs_release(fhs_ctx *c, fhs_char_t a) {
    if (!ok(c, a)) return bad(c);
    c->eng.free_char(a);
    return FHS_OK;
}
int fhs_flush(fhs_ctx *c) {
    if (!c) return FHS_ERR_ARG;
    int rc = c->eng.flush();
    if (rc) return rc;
    if (c->eng.planner) return FHS_OK;
    if (hipStreamSynchronize(c->eng.ctx.stream) != hipSuccess) return c->eng.ctx.fail(FHS_ERR_HIP, "stream sync failed");
    return FHS_OK;
}
int fhs_set_auto_flush(fhs_ctx *c, size_t n_pending) {
    if (!c) return FHS_ERR_ARG;
    c->eng.auto_flush_pending = n_pending;
    return FHS_OK;
}
int fhs_set_tick_balance(fhs_ctx *c, size_t slots) {
    if (!c) return FHS_ERR_ARG;
    c->eng.balance_slots = slots;
    return FHS_OK;
}
int fhs_resident_slots(const fhs_ctx *c) {
    if (!c) return FHS_ERR_ARG;
    // ciphertexts the selected blind-rotation kernel works on at a time: 4 workgroups of 2 wavefronts per CU for the
    // f64-FFT kernels (persistent), 2 workgroups of 4 wavefronts per CU for the exact ones
    const int a = c->eng.ctx.arith;
    return (a == 1 || a == 2) ? c->eng.ctx.wg_slots : c->eng.ctx.wg_slots / 2;
}

int fhs_submit(fhs_ctx *c) {
    if (!c) return FHS_ERR_ARG;
    return c->eng.submit();
}
int fhs_pump(fhs_ctx *c, size_t n_ticks) {
    if (!c) return FHS_ERR_ARG;
    return c->eng.pump(n_ticks);
}
int fhs_flush_async(fhs_ctx *c) {
    if (!c) return FHS_ERR_ARG;
    return c->eng.flush();
}
int fhs_download(fhs_ctx *c, fhs_char_t a, uint64_t *blocks) {
    if (!ok(c, a) || !blocks) return bad(c);
    const Bid *b = c->eng.char_blocks(a);
    for (int i = 0; i < 4; i++) {
        int rc = c->eng.read_block(b[i], blocks + (size_t)i * FHS_BIG_CT);
        if (rc) return rc;
    }
    return FHS_OK;
}
int fhs_download_string(fhs_ctx *c, const fhs_char_t *chars, size_t n, uint64_t *blocks) {
    if (!c || (n && (!chars || !blocks))) return bad(c);
    std::vector<Bid> b(4 * n);
    for (size_t i = 0; i < n; i++) {
        if (!c->eng.valid_char(chars[i])) return bad(c);
        const Bid *cb = c->eng.char_blocks(chars[i]);
        for (int k = 0; k < 4; k++) b[4 * i + k] = cb[k];
    }
    return c->eng.read_many(b.data(), b.size(), blocks);
}
int fhs_export_device(fhs_ctx *c, fhs_char_t a, uint64_t *d_blocks) {
    if (!ok(c, a) || !d_blocks) return bad(c);
    const Bid *b = c->eng.char_blocks(a);
    for (int i = 0; i < 4; i++) {
        int rc = c->eng.copy_block_to_device(b[i], d_blocks + (size_t)i * FHS_BIG_CT);
        if (rc) return rc;
    }
    return FHS_OK;
}

int fhs_export_device_async(fhs_ctx *c, fhs_char_t a, uint64_t *d_blocks) {
    if (!ok(c, a) || !d_blocks) return bad(c);
    const Bid *b = c->eng.char_blocks(a);
    for (int i = 0; i < 4; i++) {
        int rc = c->eng.copy_block_to_device(b[i], d_blocks + (size_t)i * FHS_BIG_CT, false);
        if (rc) return rc;
    }
    return FHS_OK;
}
void *fhs_stream_handle(fhs_ctx *c) { return c ? reinterpret_cast<void *>(c->eng.ctx.stream) : nullptr; }

int fhs_set_mode(fhs_ctx *c, int mode) {
    if (!c || (mode != FHS_MODE_AS_WRITTEN && mode != FHS_MODE_FUSED)) return bad(c);
    c->eng.mode = mode;
    return FHS_OK;
}

#define STR_PAT_OP(NAME, METHOD)                                                                         \
    int NAME(fhs_ctx *c, const fhs_char_t *s, size_t n, const fhs_char_t *pat, size_t m, fhs_char_t *out) { \
        if (!ok_all(c, s, n) || !ok_all(c, pat, m) || !out) return bad(c);                               \
        Strings S(&c->eng);                                                                              \
        FChar r = S.METHOD(load_str(c->eng, s, n), load_str(c->eng, pat, m));                            \
        if (int rc = finish(c, S)) return rc;                                                            \
        *out = store(c->eng, r);                                                                         \
        return FHS_OK;                                                                                   \
    }
STR_PAT_OP(fhs_str_contains, contains)
STR_PAT_OP(fhs_str_starts_with, starts_with)
STR_PAT_OP(fhs_str_ends_with, ends_with)
STR_PAT_OP(fhs_str_find, find)
STR_PAT_OP(fhs_str_rfind, rfind)
STR_PAT_OP(fhs_str_eq, eq)
STR_PAT_OP(fhs_str_ne, ne)
STR_PAT_OP(fhs_str_eq_ignore_case, eq_ignore_case)

int fhs_str_contains_clear(fhs_ctx *c, const fhs_char_t *s, size_t n, const char *pat, size_t m, fhs_char_t *out) {
    if (!ok_all(c, s, n) || (m && !pat) || !out) return bad(c);
    Strings S(&c->eng);
    FChar r = S.contains(load_str(c->eng, s, n), S.clear(pat, m));
    *out = store(c->eng, r);
    return FHS_OK;
}
int fhs_str_find_clear(fhs_ctx *c, const fhs_char_t *s, size_t n, const char *pat, size_t m, fhs_char_t *out) {
    if (!ok_all(c, s, n) || (m && !pat) || !out) return bad(c);
    Strings S(&c->eng);
    FChar r = S.find(load_str(c->eng, s, n), S.clear(pat, m));
    if (int rc = finish(c, S)) return rc;
    *out = store(c->eng, r);
    return FHS_OK;
}
int fhs_str_is_empty(fhs_ctx *c, const fhs_char_t *s, size_t n, fhs_char_t *out) {
    if (!ok_all(c, s, n) || !out) return bad(c);
    Strings S(&c->eng);
    FChar r = S.is_empty(load_str(c->eng, s, n));
    *out = store(c->eng, r);
    return FHS_OK;
}
int fhs_str_len(fhs_ctx *c, const fhs_char_t *s, size_t n, fhs_char_t *out) {
    if (!ok_all(c, s, n) || !out) return bad(c);
    Strings S(&c->eng);
    FChar r = S.len(load_str(c->eng, s, n));
    *out = store(c->eng, r);
    return FHS_OK;
}
int fhs_str_compare(fhs_ctx *c, const fhs_char_t *a, size_t na, const fhs_char_t *b, size_t nb, int cmp, fhs_char_t *out) {
    if (!ok_all(c, a, na) || !ok_all(c, b, nb) || !out || cmp < 0 || cmp > 3) return bad(c);
    Strings S(&c->eng);
    FChar r = S.comparison(load_str(c->eng, a, na), load_str(c->eng, b, nb), cmp);
    *out = store(c->eng, r);
    return FHS_OK;
}

int fhs_str_compare_partial(fhs_ctx *c, const fhs_char_t *a, size_t na, const fhs_char_t *b, size_t nb, int cmp,
                            fhs_char_t *any_diff, fhs_char_t *verdict) {
    if (!ok_all(c, a, na) || !ok_all(c, b, nb) || !any_diff || !verdict || cmp < 0 || cmp > 3 || na != nb) return bad(c);
    Strings S(&c->eng);
    FChar d, v;
    S.f_cmp_partial(load_str(c->eng, a, na), load_str(c->eng, b, nb), cmp, &d, &v);
    *any_diff = store(c->eng, d);
    *verdict = store(c->eng, v);
    return FHS_OK;
}

int fhs_flags_first_decides(fhs_ctx *c, const fhs_char_t *any_diff, const fhs_char_t *verdict, size_t n, int tie,
                            fhs_char_t *out) {
    if (!ok_all(c, any_diff, n) || !ok_all(c, verdict, n) || !out) return bad(c);
    Strings S(&c->eng);
    FChar r = S.flags_first_decides(load_str(c->eng, any_diff, n), load_str(c->eng, verdict, n), tie);
    *out = store(c->eng, r);
    return FHS_OK;
}

#define STR_MAP_OP(NAME, METHOD)                                                  \
    int NAME(fhs_ctx *c, const fhs_char_t *s, size_t n, fhs_char_t *out) {        \
        if (!ok_all(c, s, n) || (n && !out)) return bad(c);                       \
        Strings S(&c->eng);                                                       \
        FStr r = S.METHOD(load_str(c->eng, s, n));                                \
        store_str(c->eng, r, out);                                                \
        return FHS_OK;                                                            \
    }
STR_MAP_OP(fhs_str_to_upper, to_upper)
STR_MAP_OP(fhs_str_to_lower, to_lower)
STR_MAP_OP(fhs_str_trim_end, trim_end)
STR_MAP_OP(fhs_str_trim_start, trim_start)
STR_MAP_OP(fhs_str_trim, trim)
STR_MAP_OP(fhs_bubble_zeroes_right, bubble_zeroes_right)

size_t fhs_str_replace_len(size_t n, size_t mf, size_t mt) {
    const size_t d = n + 1;                       // the reference pushes one NUL (mod.rs:841,898)
    if (mf >= mt) return d;                       // handle_longer_from
    if (mf == 0) return (d + (d + 1) * mt) + 1;   // mod.rs:910-914
    return mt * d + d;                            // mod.rs:903-907
}
int fhs_str_replace(fhs_ctx *c, const fhs_char_t *s, size_t n, const fhs_char_t *from, size_t mf,
                    const fhs_char_t *to, size_t mt, fhs_char_t *out, size_t out_cap, size_t *out_len) {
    if (!ok_all(c, s, n) || !ok_all(c, from, mf) || !ok_all(c, to, mt) || !out || !out_len) return bad(c);
    if (out_cap < fhs_str_replace_len(n, mf, mt)) return c->eng.ctx.fail(FHS_ERR_ARG, "output capacity too small");
    Strings S(&c->eng);
    FStr r = S.replace(load_str(c->eng, s, n), load_str(c->eng, from, mf), load_str(c->eng, to, mt));
    store_str(c->eng, r, out);
    *out_len = r.size();
    return FHS_OK;
}
int fhs_str_replacen(fhs_ctx *c, const fhs_char_t *s, size_t n, const fhs_char_t *from, size_t mf,
                     const fhs_char_t *to, size_t mt, fhs_char_t count, fhs_char_t *out, size_t out_cap,
                     size_t *out_len) {
    if (!ok_all(c, s, n) || !ok_all(c, from, mf) || !ok_all(c, to, mt) || !ok(c, count) || !out || !out_len)
        return bad(c);
    if (out_cap < fhs_str_replace_len(n, mf, mt)) return c->eng.ctx.fail(FHS_ERR_ARG, "output capacity too small");
    Strings S(&c->eng);
    FStr r = S.replacen(load_str(c->eng, s, n), load_str(c->eng, from, mf), load_str(c->eng, to, mt),
                        load(c->eng, count));
    store_str(c->eng, r, out);
    *out_len = r.size();
    return FHS_OK;
}
int fhs_str_repeat(fhs_ctx *c, const fhs_char_t *s, size_t n, fhs_char_t count, fhs_char_t *out) {
    if (!ok_all(c, s, n) || !ok(c, count) || (n && !out)) return bad(c);
    Strings S(&c->eng);
    FStr r = S.repeat(load_str(c->eng, s, n), load(c->eng, count));
    store_str(c->eng, r, out);
    return FHS_OK;
}
int fhs_str_repeat_clear(fhs_ctx *c, const fhs_char_t *s, size_t n, size_t count, fhs_char_t *out) {
    if (!ok_all(c, s, n) || (n && count && !out)) return bad(c);
    Strings S(&c->eng);
    FStr r = S.repeat_clear(load_str(c->eng, s, n), count);
    store_str(c->eng, r, out);
    return FHS_OK;
}
int fhs_str_concatenate(fhs_ctx *c, const fhs_char_t *a, size_t na, const fhs_char_t *b, size_t nb, fhs_char_t *out) {
    if (!ok_all(c, a, na) || !ok_all(c, b, nb) || ((na + nb) && !out)) return bad(c);
    Strings S(&c->eng);
    FStr r = S.concatenate(load_str(c->eng, a, na), load_str(c->eng, b, nb));
    store_str(c->eng, r, out);
    return FHS_OK;
}
int fhs_str_strip_prefix(fhs_ctx *c, const fhs_char_t *s, size_t n, const fhs_char_t *pat, size_t m,
                         fhs_char_t *out, fhs_char_t *found) {
    if (!ok_all(c, s, n) || !ok_all(c, pat, m) || (n && !out) || !found) return bad(c);
    Strings S(&c->eng);
    FChar f;
    FStr r = S.strip_prefix(load_str(c->eng, s, n), load_str(c->eng, pat, m), &f);
    store_str(c->eng, r, out);
    *found = store(c->eng, f);
    return FHS_OK;
}
int fhs_str_strip_suffix(fhs_ctx *c, const fhs_char_t *s, size_t n, const fhs_char_t *pat, size_t m,
                         fhs_char_t *out, fhs_char_t *found) {
    if (!ok_all(c, s, n) || !ok_all(c, pat, m) || (n && !out) || !found) return bad(c);
    Strings S(&c->eng);
    FChar f;
    FStr r = S.strip_suffix(load_str(c->eng, s, n), load_str(c->eng, pat, m), &f);
    store_str(c->eng, r, out);
    *found = store(c->eng, f);
    return FHS_OK;
}

size_t fhs_str_split_dim(int kind, size_t n) { return kind == Strings::SPLIT_ASCII_WHITESPACE ? n : n + 1; }
int fhs_str_split(fhs_ctx *c, int kind, const fhs_char_t *s, size_t n, const fhs_char_t *pat, size_t m,
                  fhs_char_t count, fhs_char_t *out, size_t out_cap, size_t *dim, fhs_char_t *found) {
    if (!ok_all(c, s, n) || !ok_all(c, pat, m) || !out || !dim || !found || kind < 0 || kind > 8) return bad(c);
    const bool needs_n = kind == Strings::SPLITN || kind == Strings::RSPLITN;
    if (needs_n && !ok(c, count)) return bad(c);
    const size_t d = fhs_str_split_dim(kind, n);
    if (out_cap < d * d) return c->eng.ctx.fail(FHS_ERR_ARG, "output capacity too small");
    Strings S(&c->eng);
    FChar nn, f;
    if (needs_n) nn = load(c->eng, count);
    std::vector<FStr> r = S.split_family(kind, load_str(c->eng, s, n), load_str(c->eng, pat, m),
                                         needs_n ? &nn : nullptr, &f);
    for (size_t i = 0; i < d; i++)
        for (size_t j = 0; j < d; j++) out[i * d + j] = store(c->eng, r[i][j]);
    *dim = d;
    *found = store(c->eng, f);
    return FHS_OK;
}

int fhs_flags_or(fhs_ctx *c, const fhs_char_t *flags, size_t n, fhs_char_t *out) {
    if (!ok_all(c, flags, n) || !out) return bad(c);
    Strings S(&c->eng);
    FChar r = S.flags_or(load_str(c->eng, flags, n));
    *out = store(c->eng, r);
    return FHS_OK;
}
int fhs_flags_and(fhs_ctx *c, const fhs_char_t *flags, size_t n, fhs_char_t *out) {
    if (!ok_all(c, flags, n) || !out) return bad(c);
    Strings S(&c->eng);
    FChar r = S.flags_and(load_str(c->eng, flags, n));
    *out = store(c->eng, r);
    return FHS_OK;
}

int fhs_dist_config(fhs_ctx *c, int rank, int world) {
    if (!c || world < 1 || rank < 0 || rank >= world) return bad(c);
    c->eng.dist_rank = rank;
    c->eng.dist_world = world;
    return FHS_OK;
}
int fhs_flush_plan(fhs_ctx *c, uint64_t *n_levels, uint64_t *max_level_width) {
    if (!c || !n_levels || !max_level_width) return bad(c);
    int rc = c->eng.plan_flush();
    if (rc) return rc;
    *n_levels = c->eng.planned_levels();
    *max_level_width = c->eng.planned_levels() ? c->eng.planned_max_width() : 0;
    return FHS_OK;
}
int fhs_flush_level_exec(fhs_ctx *c, uint64_t level, uint64_t *d_slice, uint64_t *width, uint64_t *cap) {
    if (!c || !d_slice || !width || !cap || level >= c->eng.planned_levels()) return bad(c);
    const size_t w = c->eng.level_width(level), world = (size_t)c->eng.dist_world;
    const size_t cp = (w + world - 1) / world;
    const size_t lo = std::min(w, (size_t)c->eng.dist_rank * cp), hi = std::min(w, lo + cp);
    *width = w;
    *cap = cp;
    return c->eng.exec_level(level, lo, hi, d_slice);
}
int fhs_flush_level_commit(fhs_ctx *c, uint64_t level, const uint64_t *d_all) {
    if (!c || !d_all || level >= c->eng.planned_levels()) return bad(c);
    return c->eng.commit_level(level, d_all);
}
int fhs_stream_sync(fhs_ctx *c) {
    if (!c) return FHS_ERR_ARG;
    if (c->eng.planner) return FHS_OK;
    if (hipStreamSynchronize(c->eng.ctx.stream) != hipSuccess) return c->eng.ctx.fail(FHS_ERR_HIP, "stream sync failed");
    return FHS_OK;
}

int fhs_get_stats(fhs_ctx *c, fhs_stats *out) {
    if (!c || !out) return bad(c);
    out->pbs_executed = c->eng.stats.pbs_executed;
    out->pbs_folded = c->eng.stats.pbs_folded;
    out->levels = c->eng.stats.levels;
    out->max_level_width = c->eng.stats.max_level_width;
    out->blocks_live = c->eng.blocks_live();
    out->max_input_sum_c2 = c->eng.stats.max_input_sum_c2;
    out->pbs_shared = c->eng.stats.pbs_shared;
    out->pbs_extracted = c->eng.stats.pbs_extracted;
    return FHS_OK;
}
int fhs_char_sum_c2(fhs_ctx *c, fhs_char_t h, uint64_t *out) {
    if (!ok(c, h) || !out) return bad(c);
    const Bid *b = c->eng.char_blocks(h);
    int64_t m = 0;
    for (int i = 0; i < 4; i++) m = std::max<int64_t>(m, c->eng.sum_c2(b[i]));
    *out = (uint64_t)m;
    return FHS_OK;
}
int fhs_char_set_noise(fhs_ctx *c, fhs_char_t h, uint64_t sum_c2) {
    if (!ok(c, h)) return bad(c);
    const Bid *b = c->eng.char_blocks(h);
    for (int i = 0; i < 4; i++)                               // all four blocks are checked before any is changed:
        if (int rc = c->eng.set_var(b[i], sum_c2, true)) return rc;   // a refused declaration leaves the handle as it was
    for (int i = 0; i < 4; i++) (void)c->eng.set_var(b[i], sum_c2);
    return FHS_OK;
}
int fhs_trivial_value(fhs_ctx *c, fhs_char_t h, int *is_trivial, uint8_t *value) {
    if (!ok(c, h) || !is_trivial || !value) return bad(c);
    const Bid *b = c->eng.char_blocks(h);
    unsigned v = 0;
    *is_trivial = 1;
    for (int i = 0; i < 4; i++) {
        if (!c->eng.is_triv(b[i])) { *is_trivial = 0; *value = 0; return FHS_OK; }
        v += (unsigned)(c->eng.triv_val(b[i]) & 15) << (2 * i);      // carries add up like in a decryption
    }
    *value = (uint8_t)(v & 255);
    return FHS_OK;
}
int fhs_level_widths(fhs_ctx *c, uint32_t *out, size_t cap, size_t *n) {
    if (!c || !n) return bad(c);
    const auto &w = c->eng.stats.level_widths;
    *n = w.size();
    if (out) std::copy(w.begin(), w.begin() + std::min(cap, w.size()), out);
    return FHS_OK;
}
int fhs_set_rotation_sharing(fhs_ctx *c, int on) {
    if (!c) return FHS_ERR_ARG;
    c->eng.share_rotations = on != 0;
    return FHS_OK;
}
int fhs_launch_groups(fhs_ctx *c, uint32_t *out, size_t cap, size_t *n) {
    if (!c || !n) return bad(c);
    const auto &w = c->eng.stats.group_rows;
    *n = w.size();
    if (out) std::copy(w.begin(), w.begin() + std::min(cap, w.size()), out);
    return FHS_OK;
}
int fhs_reset_stats(fhs_ctx *c) {
    if (!c) return FHS_ERR_ARG;
    c->eng.stats = EngineStats();
    return FHS_OK;
}

}  // extern "C"
