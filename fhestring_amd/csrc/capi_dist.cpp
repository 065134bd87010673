// extern "C" boundary, part 3: multi-GPU inside the library (include/fhestring_hip.h "multi-GPU").
// One process per GPU; fhs_dist_init creates this context's RCCL communicator; the sharded string entry points run
// the local partial on this rank's slice, exchange one or two blocks per rank with ONE ncclAllGather on the context's
// HIP stream and evaluate the combine on every rank (so every rank ends with the result, like the reference's return
// value).  Characters of an FheString are independent PBS batches (src/server_key/mod.rs:170-177, :1122-1149,
// :1470-1541), which is what makes the slices independent.
#include <algorithm>

#include "capi_internal.h"

using namespace fhs;

namespace {

bool ok_all(fhs_ctx *c, const fhs_char_t *h, size_t n) {
    if (!c || (n && !h)) return false;
    for (size_t i = 0; i < n; i++)
        if (!c->eng.valid_char(h[i])) return false;
    return true;
}
int bad(fhs_ctx *c) { return c ? c->eng.ctx.fail(FHS_ERR_ARG, "invalid handle or null argument") : FHS_ERR_ARG; }
FChar load(Engine &e, fhs_char_t h) {
    FChar c;
    const Bid *b = e.char_blocks(h);
    for (int i = 0; i < 4; i++) {
        e.retain(b[i]);
        c.b[i] = Ref(&e, b[i]);
    }
    return c;
}
FStr load_str(Engine &e, const fhs_char_t *h, size_t n) {
    FStr s;
    s.reserve(n);
    for (size_t i = 0; i < n; i++) s.push_back(load(e, h[i]));
    return s;
}
fhs_char_t store(Engine &e, FChar &c) {
    Bid b[4];
    for (int i = 0; i < 4; i++) b[i] = c.b[i].detach();
    return e.new_char(b);
}
// the sharded formulations are the re-associated (fused) DAGs whatever the context's mode
struct FusedScope {
    Engine &e;
    int saved;
    explicit FusedScope(Engine &en) : e(en), saved(en.mode) { e.mode = 1; }
    ~FusedScope() { e.mode = saved; }
};
// gathers k blocks per rank; out[r][j] = block j of rank r
// The receivers book every imported block as ONE bootstrap output (they cannot know a sender's figure without a host
// round trip), so that is what a sender hands over: a block that is a sum of several outputs is refreshed first.  The
// partials of the sharded operations are single outputs already (found / match flags, refreshed position digits,
// 1 - same); this only bites for caller-supplied handles of fhs_dist_allgather_flags / _chars.
int gather(Engine &e, const std::vector<Ref> &local_in, std::vector<std::vector<Ref>> &out) {
    std::vector<Ref> local = local_in;
    for (Ref &x : local)
        if (e.sum_c2(x.id()) > 1) x = pbs(x, LUT_MSG);
    std::vector<Bid> ids(local.size()), got;
    for (size_t i = 0; i < local.size(); i++) ids[i] = local[i].id();
    int rc = e.gather_blocks(ids.data(), ids.size(), got);
    if (rc) return rc;
    const size_t world = (size_t)e.ctx.dist.world, k = local.size();
    out.assign(world, std::vector<Ref>());
    for (size_t r = 0; r < world; r++)
        for (size_t j = 0; j < k; j++) out[r].push_back(Ref(&e, got[r * k + j]));
    return 0;
}
FChar flag_char(Engine &e, const Ref &b) { return ch_flag(&e, b); }

}  // namespace

extern "C" {

int fhs_dist_unique_id(void *id) {
    if (!id) return FHS_ERR_ARG;
    std::string err;
    return Dist::unique_id(id, err);
}

int fhs_dist_available(void) {
    std::string err;
    return Dist::available(err) ? 1 : 0;
}

int fhs_dist_stats(const fhs_ctx *c, uint64_t *n_allgather, uint64_t *bytes_sent, int *transport) {
    if (!c) return FHS_ERR_ARG;
    const Dist &d = c->eng.ctx.dist;
    if (n_allgather) *n_allgather = d.n_gathers;
    if (bytes_sent) *bytes_sent = d.bytes_sent;
    if (transport) *transport = d.stream_ordered() ? FHS_TRANSPORT_RCCL : (d.active() ? FHS_TRANSPORT_HOST : FHS_TRANSPORT_NONE);
    return FHS_OK;
}

int fhs_dist_init(fhs_ctx *c, int rank, int world, const void *nccl_unique_id) {
    if (!c || world < 1 || rank < 0 || rank >= world || !nccl_unique_id) return bad(c);
    if (hipSetDevice(c->eng.ctx.device) != hipSuccess) return c->eng.ctx.fail(FHS_ERR_HIP, "hipSetDevice failed");
    return c->eng.ctx.dist.init_rccl(rank, world, nccl_unique_id, c->eng.ctx.err);
}

int fhs_dist_init_host_transport(fhs_ctx *c, int rank, int world, fhs_allgather_fn fn, void *user) {
    if (!c || world < 1 || rank < 0 || rank >= world || !fn) return bad(c);
    return c->eng.ctx.dist.init_host(rank, world, fn, user, c->eng.ctx.err);
}

int fhs_dist_abort(fhs_ctx *c) {
    if (!c) return FHS_ERR_ARG;
    c->eng.level_parallel = false;
    c->eng.ctx.dist.shutdown(true);                          // no flush, no stream wait, no collective teardown
    return FHS_OK;
}

int fhs_dist_shutdown(fhs_ctx *c) {
    if (!c) return FHS_ERR_ARG;
    if (int rc = c->eng.flush()) return rc;
    if (c->eng.ctx.stream) (void)hipStreamSynchronize(c->eng.ctx.stream);
    c->eng.level_parallel = false;
    c->eng.ctx.dist.shutdown();
    return FHS_OK;
}

int fhs_dist_rank(const fhs_ctx *c) { return c ? c->eng.ctx.dist.rank : FHS_ERR_ARG; }
int fhs_dist_world(const fhs_ctx *c) { return c ? c->eng.ctx.dist.world : FHS_ERR_ARG; }

int fhs_dist_level_parallel(fhs_ctx *c, int on) {
    if (!c) return FHS_ERR_ARG;
    if (on && !c->eng.ctx.dist.active()) return c->eng.ctx.fail(FHS_ERR_STATE, "fhs_dist_init first");
    if (int rc = c->eng.flush()) return rc;          // pending work runs under the mode it was recorded in
    c->eng.level_parallel = on != 0;
    return FHS_OK;
}

void fhs_dist_plan_windows(size_t n_chars, size_t m, int world, int rank, size_t *w0, size_t *w1, size_t *c0, size_t *c1) {
    const size_t n_win = m <= n_chars ? n_chars - m + 1 : 0;
    const size_t base = n_win / (size_t)world, extra = n_win % (size_t)world, r = (size_t)rank;
    const size_t lo = r * base + std::min(r, extra), cnt = base + (r < extra ? 1 : 0);
    *w0 = lo;
    *w1 = lo + cnt;
    *c0 = cnt ? lo : 0;
    *c1 = cnt ? std::min(n_chars, lo + cnt + m - 1) : 0;
}
void fhs_dist_plan_positions(size_t n_chars, int world, int rank, size_t *c0, size_t *c1) {
    const size_t base = n_chars / (size_t)world, extra = n_chars % (size_t)world, r = (size_t)rank;
    *c0 = r * base + std::min(r, extra);
    *c1 = *c0 + base + (r < extra ? 1 : 0);
}

int fhs_dist_allgather_chars(fhs_ctx *c, const fhs_char_t *local, size_t n, fhs_char_t *out) {
    if (!ok_all(c, local, n) || (n && !out)) return bad(c);
    Engine &e = c->eng;
    std::vector<Ref> blocks;
    for (size_t i = 0; i < n; i++) {
        FChar ch = load(e, local[i]);
        for (int b = 0; b < 4; b++) blocks.push_back(ch.b[b]);
    }
    std::vector<std::vector<Ref>> got;
    if (int rc = gather(e, blocks, got)) return rc;
    for (size_t r = 0; r < got.size(); r++)
        for (size_t i = 0; i < n; i++) {
            FChar ch;
            for (int b = 0; b < 4; b++) ch.b[b] = got[r][4 * i + b];
            out[r * n + i] = store(e, ch);
        }
    return FHS_OK;
}

// n flags per rank (block 0 of each char; flag chars carry three trivial-zero blocks): out[r * n + i]
int fhs_dist_allgather_flags(fhs_ctx *c, const fhs_char_t *local, size_t n, fhs_char_t *out) {
    if (!ok_all(c, local, n) || (n && !out)) return bad(c);
    Engine &e = c->eng;
    std::vector<Ref> blocks;
    for (size_t i = 0; i < n; i++) blocks.push_back(load(e, local[i]).b[0]);
    std::vector<std::vector<Ref>> got;
    if (int rc = gather(e, blocks, got)) return rc;
    for (size_t r = 0; r < got.size(); r++)
        for (size_t i = 0; i < n; i++) {
            FChar ch = flag_char(e, got[r][i]);
            out[r * n + i] = store(e, ch);
        }
    return FHS_OK;
}

static int dist_contains(fhs_ctx *c, const fhs_char_t *s, size_t n, const FStr &pat, fhs_char_t *out) {
    Engine &e = c->eng;
    FusedScope fs(e);
    Strings S(&e);
    FChar local = n >= pat.size() && !(n == 0 && !pat.empty()) ? S.contains(load_str(e, s, n), pat) : ch_trivial(&e, 0);
    std::vector<std::vector<Ref>> got;
    if (int rc = gather(e, {local.b[0]}, got)) return rc;
    FStr flags;
    for (auto &g : got) flags.push_back(flag_char(e, g[0]));
    FChar r = S.flags_or(flags);
    *out = store(e, r);
    return FHS_OK;
}
int fhs_dist_str_contains(fhs_ctx *c, const fhs_char_t *shard, size_t n, const fhs_char_t *pat, size_t m, fhs_char_t *out) {
    if (!ok_all(c, shard, n) || !ok_all(c, pat, m) || !out) return bad(c);
    return dist_contains(c, shard, n, load_str(c->eng, pat, m), out);
}
int fhs_dist_str_contains_clear(fhs_ctx *c, const fhs_char_t *shard, size_t n, const char *pat, size_t m, fhs_char_t *out) {
    if (!ok_all(c, shard, n) || (m && !pat) || !out) return bad(c);
    Strings S(&c->eng);
    return dist_contains(c, shard, n, S.clear(pat, m), out);
}

static int dist_find(fhs_ctx *c, const fhs_char_t *s, size_t n, const FStr &pat, size_t first_window, size_t total_chars,
                     fhs_char_t *out) {
    Engine &e = c->eng;
    if (total_chars >= FHS_MAX_FIND_LENGTH + pat.size())          // mod.rs:1025-1027, on the whole string
        return e.ctx.fail(FHS_ERR_LIMIT, "Maximum supported size for find reached");
    FusedScope fs(e);
    Strings S(&e);
    if (pat.empty()) {                                              // mod.rs:1017-1019: "" is found at position 0 (every window matches)
        FChar zero = ch_trivial(&e, 0);
        *out = store(e, zero);
        return FHS_OK;
    }
    // Every rank bootstraps the match flags of ITS windows (the two wide levels of find: nibble tests, AND per window --
    // 1 / world of the work), ONE all-gather hands every rank all flags (ceil(W / world) blocks per rank: 0.5 MB at
    // 256 characters on 8 GPUs), and the narrow rest -- prefix OR, index of the first flag set (Strings::first_index),
    // 255 if none -- runs replicated on every rank.  Six dependency levels like the single-GPU find; the earlier
    // formulation (a (found, position) partial per slice + "the first slice that found one decides") needed eleven,
    // which a 256-character string cannot amortise: its levels are bootstrap latencies, not throughput.
    const int world = e.ctx.dist.world, rank = e.ctx.dist.rank;
    const size_t m = pat.size();
    const size_t n_win = m <= total_chars ? total_chars - m + 1 : 0;
    size_t w0 = 0, w1 = 0, c0 = 0, c1 = 0;
    fhs_dist_plan_windows(total_chars, m, world, rank, &w0, &w1, &c0, &c1);
    if (n != c1 - c0 || (w1 > w0 && first_window != w0))
        return e.ctx.fail(FHS_ERR_ARG, "fhs_dist_str_find: the shard is not this rank's slice of fhs_dist_plan_windows");
    if (n_win == 0) {                                               // pattern longer than the string: mod.rs:1029-1031
        FChar none = ch_trivial(&e, 255);
        *out = store(e, none);
        return FHS_OK;
    }
    const size_t per = (n_win + (size_t)world - 1) / (size_t)world;  // blocks every rank contributes (short slices pad)
    std::vector<Ref> local = w1 > w0 ? S.f_find_window_flags(load_str(e, s, n), pat) : std::vector<Ref>();
    if (local.size() != w1 - w0) return e.ctx.fail(FHS_ERR_STATE, "internal: window count of the slice");
    while (local.size() < per) local.push_back(trivial_block(&e, 0));
    std::vector<std::vector<Ref>> got;
    if (int rc = gather(e, local, got)) return rc;
    std::vector<Ref> flags;
    flags.reserve(n_win);
    for (int r = 0; r < world; r++) {
        size_t a = 0, b = 0, x = 0, y = 0;
        fhs_dist_plan_windows(total_chars, m, world, r, &a, &b, &x, &y);
        for (size_t k = 0; k < b - a; k++) flags.push_back(got[(size_t)r][k]);
    }
    FChar r = S.f_find_from_flags(flags);
    *out = store(e, r);
    return FHS_OK;
}
int fhs_dist_str_find(fhs_ctx *c, const fhs_char_t *shard, size_t n, const fhs_char_t *pat, size_t m, size_t first_window,
                      size_t total_chars, fhs_char_t *out) {
    if (!ok_all(c, shard, n) || !ok_all(c, pat, m) || !out) return bad(c);
    return dist_find(c, shard, n, load_str(c->eng, pat, m), first_window, total_chars, out);
}
int fhs_dist_str_find_clear(fhs_ctx *c, const fhs_char_t *shard, size_t n, const char *pat, size_t m, size_t first_window,
                            size_t total_chars, fhs_char_t *out) {
    if (!ok_all(c, shard, n) || (m && !pat) || !out) return bad(c);
    Strings S(&c->eng);
    return dist_find(c, shard, n, S.clear(pat, m), first_window, total_chars, out);
}

// eq / eq_ignore_case of two padded strings of the SAME buffer length, character positions sharded: on well-formed
// padded strings (NULs only at the end) the reference's predicate (mod.rs:1122-1149) is the conjunction of the same
// predicate over the ranges.
int fhs_dist_str_eq(fhs_ctx *c, const fhs_char_t *a, size_t na, const fhs_char_t *b, size_t nb, int ignore_case,
                    fhs_char_t *out) {
    if (!ok_all(c, a, na) || !ok_all(c, b, nb) || !out || na != nb) return bad(c);
    Engine &e = c->eng;
    FusedScope fs(e);
    Strings S(&e);
    FChar local = ch_trivial(&e, 1);
    if (na) {
        FStr x = load_str(e, a, na), y = load_str(e, b, nb);
        local = ignore_case ? S.eq_ignore_case(x, y) : S.eq(x, y);
    }
    std::vector<std::vector<Ref>> got;
    if (int rc = gather(e, {local.b[0]}, got)) return rc;
    FStr flags;
    for (auto &g : got) flags.push_back(flag_char(e, g[0]));
    FChar r = S.flags_and(flags);
    *out = store(e, r);
    return FHS_OK;
}

// lt / le / gt / ge (cmp 0..3), positions sharded: each rank reduces its slices to (some position differs, verdict at the
// first differing position) -- the positional half of mod.rs:1497-1518 -- and the first range that differs decides;
// nothing differs => equal buffers => le / ge = 1.  NUL padding sorts below every character (the length tie-break).
int fhs_dist_str_compare(fhs_ctx *c, const fhs_char_t *a, size_t na, const fhs_char_t *b, size_t nb, int cmp, fhs_char_t *out) {
    if (!ok_all(c, a, na) || !ok_all(c, b, nb) || !out || na != nb || cmp < 0 || cmp > 3) return bad(c);
    Engine &e = c->eng;
    FusedScope fs(e);
    Strings S(&e);
    FChar d = ch_trivial(&e, 0), v = ch_trivial(&e, 0);
    if (na) S.f_cmp_partial(load_str(e, a, na), load_str(e, b, nb), cmp, &d, &v);
    std::vector<std::vector<Ref>> got;
    if (int rc = gather(e, {d.b[0], v.b[0]}, got)) return rc;
    FStr ds, vs;
    for (auto &g : got) {
        ds.push_back(flag_char(e, g[0]));
        vs.push_back(flag_char(e, g[1]));
    }
    FChar r = S.flags_first_decides(ds, vs, (cmp == 1 || cmp == 3) ? 1 : 0);
    *out = store(e, r);
    return FHS_OK;
}

}  // extern "C"
