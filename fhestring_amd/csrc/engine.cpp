#include "engine.h"
#include "../../include/fhestring_hip.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <map>

namespace fhs {

namespace {
constexpr size_t POOL_STRIDE = 2050;        // u64 words per pooled block (16-byte aligned rows)
constexpr size_t POOL_CHUNK_BLOCKS = 2048;  // 33.6 MB per hipMalloc
}  // namespace

int Engine::on_key_loaded() {
    if (!d_luts_) {
        std::vector<uint64_t> host((size_t)LUT_COUNT * POLY_N);
        for (int id = 0; id < LUT_COUNT; id++) make_lut_poly(id, host.data() + (size_t)id * POLY_N);
        hipError_t e = hipMalloc(&d_luts_, host.size() * 8);
        if (e != hipSuccess) return ctx.hip_fail(e, "hipMalloc luts");
        e = hipMemcpy(d_luts_, host.data(), host.size() * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) return ctx.hip_fail(e, "copy luts");
    }
    return 0;
}

void Engine::shutdown() {
    if (planner) return;
    if (ctx.stream) (void)hipStreamSynchronize(ctx.stream);
    for (void *c : chunks_) (void)hipFree(c);
    chunks_.clear();
    free_blocks_.clear();
    if (d_luts_) (void)hipFree(d_luts_);
    d_luts_ = nullptr;
    if (last_group_done_) (void)hipEventDestroy(last_group_done_);
    last_group_done_ = nullptr;
    if (upload_done_) (void)hipEventDestroy(upload_done_);
    upload_done_ = nullptr;
    if (upload_pin_) (void)hipHostFree(upload_pin_);
    if (upload_dev_) (void)hipFree(upload_dev_);
    upload_pin_ = upload_dev_ = nullptr;
    upload_words_ = 0;
    for (Staging &c : staging_) {
        if (c.done) (void)hipEventDestroy(c.done);
        if (c.p) (void)hipHostFree(c.p);
    }
    staging_.clear();
    plan_buf_.release();
    tick_buf_.release();
    batch_in_.release();
    ctx.shutdown();
}

uint64_t *Engine::alloc_block() {
    if (planner) {                                   // distinct non-null tokens, never dereferenced
        live_dev_blocks_++;
        return reinterpret_cast<uint64_t *>((uintptr_t)0x1000 + 8 * (uintptr_t)(++planner_tokens_));
    }
    if (free_blocks_.empty()) {
        void *c = nullptr;
        if (hipMalloc(&c, POOL_CHUNK_BLOCKS * POOL_STRIDE * 8) != hipSuccess) return nullptr;
        chunks_.push_back(c);
        uint64_t *base = static_cast<uint64_t *>(c);
        for (size_t i = POOL_CHUNK_BLOCKS; i-- > 0;) free_blocks_.push_back(base + i * POOL_STRIDE);
    }
    uint64_t *p = free_blocks_.back();
    free_blocks_.pop_back();
    live_dev_blocks_++;
    return p;
}
void Engine::free_block(uint64_t *p) {
    // A scheduled job level that has not been enqueued yet may still read this block: it becomes reusable once every
    // tick scheduled so far is in the stream (stream order then protects the readers).
    if (!planner && last_sched_tick_ >= next_tick_) free_after_[last_sched_tick_].push_back(p);
    else if (!planner) free_blocks_.push_back(p);
    live_dev_blocks_--;
}

Bid Engine::new_node() {
    Bid id;
    if (!free_nodes_.empty()) {
        id = free_nodes_.back();
        free_nodes_.pop_back();
    } else {
        id = (Bid)nodes_.size();
        nodes_.emplace_back();
    }
    BlockNode &n = nodes_[id];
    const uint32_t g = n.gen + 1;
    n = BlockNode();
    n.gen = g;
    n.refs = 1;
    return id;
}

void Engine::retain(Bid b) { nodes_[b].refs++; }

void Engine::release(Bid b) {
    if (nodes_[b].refs > 1) {                                 // the common case: no heap traffic, nothing dies
        nodes_[b].refs--;
        return;
    }
    // iterative: chains of pending nodes can be tens of thousands deep (as-written `len`)
    std::vector<Bid> stack{b};
    while (!stack.empty()) {
        Bid id = stack.back();
        stack.pop_back();
        BlockNode &n = nodes_[id];
        if (--n.refs > 0) continue;
        if (n.kind == BlockNode::MAT && n.dev) free_block(n.dev);
        if (n.kind == BlockNode::PBS && n.src) stack.push_back(n.src);
        if (n.kind == BlockNode::LIN)
            for (const Term &t : n.terms) stack.push_back(t.blk);
        n.terms.clear();
        n.terms.shrink_to_fit();
        n.kind = BlockNode::FREE;
        n.dev = nullptr;
        n.src = 0;
        free_nodes_.push_back(id);
    }
}

Bid Engine::triv(int v) {
    Bid id = new_node();
    nodes_[id].kind = BlockNode::TRIV;
    nodes_[id].triv = (uint8_t)(v & 31);
    return id;
}

Bid Engine::from_host(const uint64_t *ct) {
    if (!planner) (void)hipSetDevice(ctx.device);   // one process may see several GPUs (torch sets its own current device)
    uint64_t *d = alloc_block();
    if (!d) return 0;
    if (!planner && hipMemcpyAsync(d, ct, BIG_CT * 8, hipMemcpyHostToDevice, ctx.stream) != hipSuccess) {
        free_block(d);
        return 0;
    }
    Bid id = new_node();
    nodes_[id].kind = BlockNode::MAT;
    nodes_[id].dev = d;
    if (planner && trace_plan) { trace_.push_back(TR_UPLOAD); trace_.push_back((uint64_t)(uintptr_t)d); }
    return id;
}

// pinned host buffer + device mirror for `n_blocks` rows and their pointer table (uploads and downloads of whole strings)
bool Engine::ensure_staging(size_t n_blocks) {
    const size_t words = n_blocks * BIG_CT + n_blocks;
    if (upload_words_ < words) {
        if (upload_done_) (void)hipEventSynchronize(upload_done_);
        if (upload_pin_) (void)hipHostFree(upload_pin_);
        if (upload_dev_) { (void)hipStreamSynchronize(ctx.stream); (void)hipFree(upload_dev_); }
        upload_pin_ = upload_dev_ = nullptr;
        upload_words_ = 0;
        const size_t want = std::max<size_t>(n_blocks, 260) * (BIG_CT + 1);
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, want * 8, hipHostMallocDefault) == hipSuccess && hipMalloc(&dp, want * 8) == hipSuccess) {
            upload_pin_ = static_cast<uint64_t *>(hp);
            upload_dev_ = static_cast<uint64_t *>(dp);
            upload_words_ = want;
        } else {
            if (hp) (void)hipHostFree(hp);
            if (dp) (void)hipFree(dp);
        }
    }
    return upload_pin_ && (upload_done_ || hipEventCreateWithFlags(&upload_done_, hipEventDisableTiming) == hipSuccess);
}

int Engine::from_host_many(const uint64_t *cts, size_t count, Bid *out) {
    for (size_t i = 0; i < count; i++) out[i] = 0;
    auto undo = [&](size_t n) {
        for (size_t i = 0; i < n; i++) release(out[i]);
        for (size_t i = 0; i < count; i++) out[i] = 0;
        return -1;
    };
    auto one_by_one = [&](size_t from) {
        for (size_t i = from; i < count; i++)
            if (!(out[i] = from_host(cts + i * BIG_CT))) return undo(i);
        return 0;
    };
    if (planner || count < 4) return one_by_one(0);
    (void)hipSetDevice(ctx.device);
    // staging: [count x 2049 words][count destination pointers], pinned on the host and mirrored on the device: one copy,
    // one scatter launch (pool blocks are not neighbours once the free list has been through a few operations)
    constexpr size_t MAX_BATCH = 2048;               // 33.6 MB per pass
    for (size_t done = 0; done < count;) {
        const size_t n = std::min(MAX_BATCH, count - done);
        const size_t words = n * BIG_CT + n;
        if (!ensure_staging(n)) return one_by_one(done);             // no staging memory: block by block
        (void)hipEventSynchronize(upload_done_);     // the previous copy has left the pinned buffer (no-op before the first)
        {
            // pageable -> pinned: one thread copies ~10 GB/s, which for the 537 MB of two 4097-character strings is as long
            // as their (threaded) client encryption; large passes are split over a few host threads
            const size_t bytes = n * BIG_CT * 8;
            const unsigned nt = bytes >= ((size_t)8 << 20) ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1;
            if (nt <= 1) {
                std::memcpy(upload_pin_, cts + done * BIG_CT, bytes);
            } else {
                const char *src = reinterpret_cast<const char *>(cts + done * BIG_CT);
                char *dst = reinterpret_cast<char *>(upload_pin_);
                const size_t part = (bytes / nt + 4095) & ~(size_t)4095;
                std::vector<std::thread> th;
                for (unsigned t = 1; t < nt; t++) {
                    const size_t lo = std::min(bytes, t * part), hi = std::min(bytes, (t + 1) * part);
                    if (hi > lo) th.emplace_back([=] { std::memcpy(dst + lo, src + lo, hi - lo); });
                }
                std::memcpy(dst, src, std::min(bytes, part));
                for (auto &x : th) x.join();
            }
        }
        for (size_t k = 0; k < n; k++) {
            uint64_t *d = alloc_block();
            if (!d) return undo(done + k);
            Bid id = new_node();
            nodes_[id].kind = BlockNode::MAT;
            nodes_[id].dev = d;
            out[done + k] = id;
            upload_pin_[n * BIG_CT + k] = (uint64_t)(uintptr_t)d;
        }
        if (hipMemcpyAsync(upload_dev_, upload_pin_, words * 8, hipMemcpyHostToDevice, ctx.stream) != hipSuccess) return undo(done + n);
        (void)hipEventRecord(upload_done_, ctx.stream);
        if (launch_scatter_blocks(upload_dev_, reinterpret_cast<uint64_t *const *>(upload_dev_ + n * BIG_CT), (int)n, ctx.stream) != hipSuccess)
            return undo(done + n);
        done += n;
    }
    return 0;
}

Bid Engine::from_device(const uint64_t *d_ct) {
    if (!planner) (void)hipSetDevice(ctx.device);
    uint64_t *d = alloc_block();
    if (!d) return 0;
    if (!planner && hipMemcpyAsync(d, d_ct, BIG_CT * 8, hipMemcpyDeviceToDevice, ctx.stream) != hipSuccess) {
        free_block(d);
        return 0;
    }
    Bid id = new_node();
    nodes_[id].kind = BlockNode::MAT;
    nodes_[id].dev = d;
    return id;
}

// sum(coef*blk) + konst.  Trivial terms fold into the constant, nested LINs are flattened and
// duplicate blocks merged, so a LIN only ever refers to MAT or PBS nodes.
Bid Engine::lin(const Term *terms, size_t n, int konst) {
    int64_t k = konst;
    std::vector<Term> flat;
    auto add = [&](int64_t c, Bid b) {
        if (c == 0) return;
        for (Term &t : flat)
            if (t.blk == b) { t.coef += c; return; }
        flat.push_back({c, b});
    };
    for (size_t i = 0; i < n; i++) {
        const int64_t c = terms[i].coef;
        const BlockNode &s = nodes_[terms[i].blk];
        if (c == 0) continue;
        if (s.kind == BlockNode::TRIV) k += c * s.triv;
        else if (s.kind == BlockNode::LIN) {
            k += c * s.konst;
            for (const Term &t : s.terms) add(c * t.coef, t.blk);
        } else add(c, terms[i].blk);
    }
    flat.erase(std::remove_if(flat.begin(), flat.end(), [](const Term &t) { return t.coef == 0; }), flat.end());
    if (flat.empty()) return triv((int)(((k % 32) + 32) % 32));
    if (flat.size() == 1 && flat[0].coef == 1 && (((k % 32) + 32) % 32) == 0) {
        retain(flat[0].blk);   // identity: share the block
        return flat[0].blk;
    }
    Bid id = new_node();
    uint32_t lvl = 0;
    for (const Term &t : flat) {
        retain(t.blk);
        lvl = std::max(lvl, nodes_[t.blk].level);
    }
    BlockNode &nn = nodes_[id];
    nn.kind = BlockNode::LIN;
    nn.konst = (int32_t)(((k % 32) + 32) % 32);
    nn.level = lvl;
    nn.terms = std::move(flat);
    return id;
}

void Engine::describe_block(Bid b, std::vector<uint64_t> &out) const {
    const BlockNode &n = nodes_[b];
    if (n.kind == BlockNode::TRIV) { out.insert(out.end(), {0, (uint64_t)(triv_val(b) & 31), 0}); return; }
    if (n.kind == BlockNode::MAT) { out.insert(out.end(), {1, 0, 1, (uint64_t)(uintptr_t)n.dev, 1}); return; }
    if (n.kind != BlockNode::LIN) { out.insert(out.end(), {3, 0, 0}); return; }
    out.insert(out.end(), {2, (uint64_t)(n.konst & 31), (uint64_t)n.terms.size()});
    for (const Term &t : n.terms) {
        const BlockNode &tb = nodes_[t.blk];
        out.push_back(tb.kind == BlockNode::MAT ? (uint64_t)(uintptr_t)tb.dev : 0);
        out.push_back((uint64_t)t.coef);
    }
}

int64_t Engine::sum_c2(Bid b) const {
    const BlockNode &n = nodes_[b];
    if (n.kind == BlockNode::TRIV) return 0;
    if (n.kind == BlockNode::MAT) return n.var;
    if (n.kind != BlockNode::LIN) return 1;
    return lin_c2(n.terms);
}

int64_t Engine::lin_c2(const std::vector<Term> &terms) const {
    // var(sum c_i e_i) = sum c_i^2 var_i + 2 sum_{i<j} rho_ij c_i c_j sqrt(var_i var_j) over extractions of ONE rotation.
    // Measured on the MI355X (profiles/r05_rotation_sharing_rho.txt, tools/rho_profile.py): rho falls linearly with the
    // constant difference dt from +0.45 (dt -> 0) through 0 (dt = 8) to -0.45 (dt = 15) -- the decomposition-rounding error
    // reaches every coefficient through the negacyclic product with the binary GLWE key -- and is exactly -1 at dt = 16
    // (the same coefficient negated: such rows never share).  The measurement is per key and per sample (max |rho| 0.48
    // +- 0.03 on one key, profiles/r06_margins.json repeats it on three), so the bookkeeping does not rest on it: a group
    // is charged as FULLY correlated, |rho| <= 1, which Cauchy-Schwarz proves whatever the key and the signs:
    //     group cost = (sum |c|)^2   (>= sum c^2 + 2 sum_{i<j} rho_ij c_i c_j for every |rho_ij| <= 1).
    int64_t c2 = 0;
    struct Grp { uint32_t rot; int64_t sum_abs, sum_c2, var; };
    Grp grp[8];
    int ng = 0;
    for (const Term &t : terms) {
        const BlockNode &tb = nodes_[t.blk];
        const int64_t var = tb.kind == BlockNode::MAT ? tb.var : 1;
        if (tb.kind != BlockNode::MAT || tb.rot == 0) { c2 += t.coef * t.coef * var; continue; }
        int g = 0;
        while (g < ng && grp[g].rot != tb.rot) g++;
        if (g == ng) {
            if (ng == 8) { c2 += t.coef * t.coef * var; continue; }       // (more than 8 shared rotations in one sum: unseen)
            grp[ng++] = Grp{tb.rot, 0, 0, 0};
        }
        grp[g].sum_abs += t.coef < 0 ? -t.coef : t.coef;
        grp[g].sum_c2 += t.coef * t.coef;
        grp[g].var = std::max(grp[g].var, var);
    }
    for (int g = 0; g < ng; g++)
        c2 += grp[g].sum_abs * grp[g].sum_abs * grp[g].var;
    return c2;
}

int Engine::set_var(Bid b, uint64_t v, bool check_only) {
    BlockNode &n = nodes_[b];
    if (n.kind == BlockNode::TRIV) return 0;                  // a plaintext block carries no noise whatever is declared
    if (n.kind != BlockNode::MAT) return ctx.fail(-1, "noise can only be declared for uploaded ciphertexts");
    // a declaration can only ADD to what the bookkeeping knows: an upload starts at 1, a block the library itself
    // materialised from a sum keeps at least its own figure (no laundering through this entry point)
    if (v < n.var) return ctx.fail(-1, "fhs_char_set_noise: the declared figure is below the one the library tracks for this block");
    if (!check_only) n.var = (uint16_t)std::min<uint64_t>(std::max<uint64_t>(v, 1), 65535);
    return 0;
}

Bid Engine::pbs(Bid x, int lut) {
    const BlockNode &s = nodes_[x];
    if (s.kind == BlockNode::TRIV) {   // constant folding: no PBS is executed
        stats.pbs_folded++;
        return triv(lut_eval(lut, s.triv));
    }
    const uint32_t lvl = s.level + 1;
    // fused mode: an identical bootstrap (same LUT on the same linear combination of the same blocks) is computed once,
    // e.g. the high-nibble test of a character against pattern characters that share their high nibble
    uint64_t h1 = 0, h2 = 0, nk_hash = 0;
    const bool share = mode == 1;
    if (share) {
        auto mix = [](uint64_t v, uint64_t k) {
            v = (v ^ (v >> 31)) * k;
            v = (v ^ (v >> 29)) * 0xBF58476D1CE4E5B9ull;
            return v ^ (v >> 32);
        };
        auto term = [&](uint64_t blk, uint64_t gen, int64_t coef) {      // commutative: the order of terms is irrelevant
            const uint64_t v = (blk << 32 | gen) + 0x9E3779B97F4A7C15ull * (uint64_t)coef;
            h1 += mix(v, 0x94D049BB133111EBull);
            h2 += mix(v + 0x632BE59BD9B4E019ull, 0xD6E8FEB86659FD93ull);
        };
        if (s.kind == BlockNode::LIN) {
            for (const Term &t : s.terms) term(t.blk, nodes_[t.blk].gen, t.coef);
            nk_hash = mix(h1 ^ h2 ^ ((uint64_t)lut << 48), 0x9E3779B97F4A7C15ull) | 1;      // before the constant goes in
            h1 += mix((uint64_t)(uint32_t)s.konst + 77, 0x94D049BB133111EBull);
            h2 += mix((uint64_t)(uint32_t)s.konst + 131, 0xD6E8FEB86659FD93ull);
        } else {
            term(x, s.gen, 1);
            nk_hash = mix(h1 ^ h2 ^ ((uint64_t)lut << 48), 0x9E3779B97F4A7C15ull) | 1;
        }
        h1 = mix(h1 + (uint64_t)lut * 0x9E3779B97F4A7C15ull, 0xD6E8FEB86659FD93ull);
        h2 = mix(h2 ^ ((uint64_t)lut << 40), 0x94D049BB133111EBull);
        auto it = cse_.find(h1);
        if (it != cse_.end() && it->second.h2 == h2) {
            const Bid id = it->second.id;
            const BlockNode &c = nodes_[id];
            if (c.gen == it->second.gen && c.refs > 0 && (c.kind == BlockNode::PBS || c.kind == BlockNode::MAT)) {
                stats.pbs_shared++;
                retain(id);
                return id;
            }
        }
    }
    retain(x);
    Bid id = new_node();
    BlockNode &n = nodes_[id];
    n.kind = BlockNode::PBS;
    n.src = x;
    n.lut = (uint16_t)lut;
    n.level = lvl;
    n.nk = (share && share_rotations && !(level_parallel && ctx.dist.active())) ? nk_hash : 0;
    pending_.push_back({id, n.gen});
    if (lvl == 1) { n_depth1_++; depth1_add(n); }
    if (auto_flush_pending && !manual_jobs_ && !capture_max_rows && !in_auto_flush_ &&
        sched_.empty() && !(dist_world > 1 && !level_parallel) && (planner || ctx.key_loaded)) {
        // peel the ready level when a whole batch has accumulated -- or, on a real device, as soon as a grid's worth is
        // ready and the previous launch group has finished (polled every 256 recorded bootstraps): the GPU never idles
        // while the host is still recording, and nothing narrower than one full round of workgroups is launched early
        // (level-parallel ranks must all take the same decisions: there only the deterministic count rule applies)
        // counted in blind ROTATIONS: rows that will share one (rotation sharing) are one unit of GPU work
        const size_t ready = depth1_rotations();
        bool go = ready >= auto_flush_pending;
        peel_limit_ = 0;
        const size_t round = balance_slots ? balance_slots : 1024;
        if (!go && ready >= round && !planner && !level_parallel && last_group_done_ && (++idle_poll_ & 255) == 0) {
            go = hipEventQuery(last_group_done_) == hipSuccess;
            // an idle GPU gets whole rounds of the persistent kernel only: 1 191 ready rows launched as they are cost two
            // rounds; the remainder stays pending (it is ready, and joins the next launch)
            peel_limit_ = ready / round * round;
        }
        if (go) {
            in_auto_flush_ = true;
            if (int rc = plan_job(true, true)) {              // kept: the caller's next flush reports THIS cause
                if (!auto_flush_rc_) { auto_flush_rc_ = rc; auto_flush_err_ = ctx.err; }
            }
            in_auto_flush_ = false;
        }
    }
    if (share) {
        if (cse_.empty()) cse_.reserve(1u << 18);
        if (cse_.size() > (1u << 21)) cse_.clear();           // bounded: stale entries are only dead weight
        cse_[h1] = CseEntry{h2, id, n.gen};
    }
    return id;
}

// ------------------------------------------------------------------------------------------
// flush: plan every level on the host, upload the plan once, enqueue all launches
// ------------------------------------------------------------------------------------------
int Engine::flush() {
    if (auto_flush_rc_) {                                     // an automatic partial flush failed while the DAG was recorded
        const int rc = auto_flush_rc_;
        auto_flush_rc_ = 0;
        return ctx.fail(rc, "automatic partial flush failed: " + auto_flush_err_);
    }
    while (!sched_.empty())                                   // drain the scheduled ticks of submitted jobs first
        if (int rc = pump(1)) return rc;
    if (level_parallel && ctx.dist.active()) {                // world 1 too: same stream-ordered path
        if (!ctx.dist.active()) return ctx.fail(-3, "level-parallel flush without a transport (fhs_dist_init)");
        return plan_job(true);
    }
    if (dist_world > 1 && !pending_.empty())
        return ctx.fail(-3, "distributed context: pending PBS must be run with fhs_flush_plan/level_exec/level_commit");
    manual_jobs_ = false;                                     // every scheduled tick is in the stream: automatic partial flushes may resume
    const bool plan_capture = capture_max_rows && !capture_live;   // the all-at-once plan keeps the records (no sharing)
    if (!plan_capture && balance_slots) {
        // round-aligned launch groups inside ONE operation as well: the levels go through the tick scheduler at row
        // granularity (plan_job), every tick is enqueued as soon as it is complete
        if (int rc = plan_job(false, false, true)) return rc;
        while (!sched_.empty())
            if (int rc = pump(1)) return rc;
        return 0;
    }
    if (!plan_capture) return plan_job(true);                 // level by level: planning overlaps execution
    int rc = plan_flush();                                    // capture mode: the all-at-once plan keeps the records
    if (rc) return rc;
    for (size_t k = 0; k < plan_.levels.size(); k++)
        if ((rc = exec_level(k, 0, plan_.levels[k].count, nullptr))) return rc;
    plan_.levels.clear();
    return 0;
}

// ------------------------------------------------------------------------------------------
// level-skewed batching: jobs, ticks
// ------------------------------------------------------------------------------------------
int Engine::plan_job(bool run_now, bool first_level_only, bool stream_pump) {
    if (pending_.empty()) return 0;
    if (!run_now && level_parallel && ctx.dist.active())
        return ctx.fail(-3, "fhs_submit is not available in level-parallel mode");
    if (!planner && !ctx.key_loaded) return ctx.fail(-3, "server key not loaded");
    std::map<uint32_t, std::vector<Bid>> by_level;
    if (first_level_only) {
        // peel depth 1 only: the other pending nodes stay pending, one level shallower than before
        std::vector<Pend> rest;
        n_depth1_ = 0;
        n_depth1_solo_ = 0;
        depth1_keys_.clear();
        const size_t limit = peel_limit_ ? peel_limit_ : ~(size_t)0;     // in ROTATIONS: rows sharing a taken row's key ride along
        size_t left_ready = 0, taken_rot = 0;
        std::unordered_map<uint64_t, char> taken_keys;
        for (const Pend &p : pending_) {
            BlockNode &n = nodes_[p.id];
            if (n.kind != BlockNode::PBS || n.gen != p.gen) continue;
            bool take = false;
            if (n.level <= 1) {
                if (n.nk && taken_keys.count(n.nk)) take = true;
                else if (taken_rot < limit) {
                    take = true;
                    taken_rot++;
                    if (n.nk) taken_keys[n.nk] = 1;
                }
            }
            if (take) by_level[1].push_back(p.id);
            else {
                if (n.level <= 1) { left_ready++; depth1_add(n); }
                rest.push_back(p);
            }
        }
        // every ready row taken: the rest moves one level up.  A partial peel (whole rounds only) leaves the levels alone:
        // a stale level is only ever too HIGH, which keeps the order valid
        if (left_ready == 0) {
            for (const Pend &p : rest)
                if (--nodes_[p.id].level == 1) { n_depth1_++; depth1_add(nodes_[p.id]); }
        } else {
            n_depth1_ = left_ready;
        }
        peel_limit_ = 0;
        pending_.swap(rest);
    } else {
        for (const Pend &p : pending_)
            if (nodes_[p.id].kind == BlockNode::PBS && nodes_[p.id].gen == p.gen) by_level[nodes_[p.id].level].push_back(p.id);
        pending_.clear();
        n_depth1_ = 0;
        n_depth1_solo_ = 0;
        depth1_keys_.clear();
    }
    uint64_t tick = next_tick_ - 1;                           // the job's first level goes to next_tick_ at the earliest
    const uint64_t job = ++job_counter_;
    std::vector<uint64_t> row_need;                           // per row: latest tick that produces one of its inputs
    for (auto lvit = by_level.begin(); lvit != by_level.end(); ++lvit) {
        std::vector<Bid> &lv = lvit->second;
        TickLevel tl;
        tl.job = job;
        // ---- rotation sharing ------------------------------------------------------------------------------------
        // Rows of this level with the same look-up table on the same linear combination up to its trivial CONSTANT
        // (e.g. the nibble of a character tested against the different nibbles of a clear pattern: is0(x - c)) share ONE
        // keyswitch + blind rotation: adding c * Delta to a ciphertext rotates the accumulator by X^(128 c) exactly, so the
        // other rows are further sample extractions of the leader's accumulator (extract_shift_kernel) -- the same
        // ciphertext a bootstrap of their own would give, up to decomposition ties (the CPU oracle restates it:
        // orc_pbs_shifted).  The level is reordered: rotation rows first (R of them), followers behind.
        std::vector<ShareRow> followers;
        size_t R = lv.size();
        if (share_rotations && mode == 1 && lv.size() > 1 && !(level_parallel && ctx.dist.active())) {
            std::unordered_map<uint64_t, std::vector<uint32_t>> seen;     // hash of (lut, terms) -> leader positions
            std::vector<Bid> rot, fol;
            std::vector<ShareRow> fmeta;
            std::vector<uint32_t> used;                                    // per rotation row: bit t = an extraction at shift t exists
            auto key_terms = [&](const BlockNode &src, Bid self, std::vector<std::pair<Bid, int64_t>> &tt, int &konst) {
                tt.clear();
                if (src.kind == BlockNode::LIN) {
                    for (const Term &t : src.terms) tt.emplace_back(t.blk, t.coef);
                    std::sort(tt.begin(), tt.end());
                    konst = src.konst;
                } else {
                    tt.emplace_back(self, 1);
                    konst = 0;
                }
            };
            std::vector<std::pair<Bid, int64_t>> ta, tb;
            for (Bid b : lv) {
                const BlockNode &n = nodes_[b];
                int ka = 0;
                key_terms(nodes_[n.src], n.src, ta, ka);
                uint64_t h = 0x9E3779B97F4A7C15ull * (uint64_t)(n.lut + 1);
                for (auto &t : ta) {
                    h ^= ((uint64_t)t.first << 20) + (uint64_t)t.second * 0xBF58476D1CE4E5B9ull;
                    h = (h ^ (h >> 29)) * 0x94D049BB133111EBull;
                }
                std::vector<uint32_t> &cand = seen[h];
                bool shared = false;
                for (uint32_t pos : cand) {
                    const BlockNode &ln = nodes_[rot[pos]];
                    int kb = 0;
                    key_terms(nodes_[ln.src], ln.src, tb, kb);
                    if (ln.lut != n.lut || ta != tb) continue;
                    // a constant difference of 16 is the SAME coefficient of the accumulator, negated (X^2048 = -1): its
                    // error is exactly minus the other's (rho = -1; the full-correlation charge of lin_c2 covers it, but a
                    // sum a - b of the two would DOUBLE the error for no information) -- no two members of a group may be
                    // 16 apart (such a row joins another group of the same key, or starts one)
                    const uint32_t t = (uint32_t)(((ka - kb) % 32 + 32) % 32);
                    if (used[pos] & (1u << ((t + 16) & 31))) continue;
                    used[pos] |= 1u << t;
                    fol.push_back(b);
                    fmeta.push_back({pos, 128 * t, nullptr});
                    shared = true;
                    break;
                }
                if (!shared) {
                    cand.push_back((uint32_t)rot.size());
                    rot.push_back(b);
                    used.push_back(1u);                                    // the leader itself: shift 0
                }
            }
            if (!fol.empty()) {
                // Sharing must not push a CONSUMER over the noise budget: extractions of one rotation are
                // correlated (lin_c2 charges them as fully correlated), so a later bootstrap whose input sums several members of one group with the same
                // sign is charged cross terms the string layer did not see when it built that sum.  Every pending consumer
                // is known here (deeper levels of this plan; in a partial peel the nodes still pending) -- consumers recorded
                // later see the groups through sum_c2().  A follower whose group would take a consumer from within the
                // budget to beyond it gets a rotation of its own.
                std::unordered_map<Bid, uint32_t> member;              // node -> leader position (leaders with followers too)
                for (size_t i = 0; i < fol.size(); i++) {
                    member[fol[i]] = fmeta[i].lead_row;
                    member[rot[fmeta[i].lead_row]] = fmeta[i].lead_row;
                }
                std::vector<char> unshare(fol.size(), 0);
                bool any_unshare = false;
                auto check = [&](Bid consumer) {
                    const BlockNode &cn = nodes_[consumer];
                    if (cn.kind != BlockNode::PBS) return;
                    const BlockNode &src = nodes_[cn.src];
                    if (src.kind != BlockNode::LIN || src.terms.size() < 2) return;
                    struct G { uint32_t lead; int64_t sum_abs, sum_c2; };
                    G g[8];
                    int ng = 0;
                    bool multi = false;
                    for (const Term &t : src.terms) {
                        auto it = member.find(t.blk);
                        if (it == member.end()) continue;
                        int k = 0;
                        while (k < ng && g[k].lead != it->second) k++;
                        if (k == ng) { if (ng == 8) continue; g[ng++] = G{it->second, 0, 0}; }
                        else multi = true;
                        g[k].sum_abs += t.coef < 0 ? -t.coef : t.coef;
                        g[k].sum_c2 += t.coef * t.coef;
                    }
                    if (!multi) return;
                    int64_t plain = lin_c2(src.terms), extra = 0;      // (members are still pending: counted as independent)
                    for (int k = 0; k < ng; k++) extra += g[k].sum_abs * g[k].sum_abs - g[k].sum_c2;   // as lin_c2 will
                    if (plain + extra <= FHS_NOISE_BUDGET_SUM_C2 || extra == 0) return;
                    for (const Term &t : src.terms)                    // the followers among this sum's terms leave their groups
                        for (size_t i = 0; i < fol.size(); i++)
                            if (fol[i] == t.blk && !unshare[i]) { unshare[i] = 1; any_unshare = true; }
                };
                for (auto nx = std::next(lvit); nx != by_level.end(); ++nx)
                    for (Bid c : nx->second) check(c);
                if (first_level_only)
                    for (const Pend &pd : pending_)
                        if (nodes_[pd.id].gen == pd.gen) check(pd.id);
                if (any_unshare) {
                    std::vector<Bid> fol2;
                    std::vector<ShareRow> fmeta2;
                    for (size_t i = 0; i < fol.size(); i++) {
                        if (unshare[i]) { rot.push_back(fol[i]); used.push_back(1u); }   // a rotation row of its own (appended: positions stay valid)
                        else { fol2.push_back(fol[i]); fmeta2.push_back(fmeta[i]); }
                    }
                    fol.swap(fol2);
                    fmeta.swap(fmeta2);
                }
            }
            if (!fol.empty()) {
                R = rot.size();
                lv.swap(rot);
                lv.insert(lv.end(), fol.begin(), fol.end());
                followers.swap(fmeta);
                tl.body.assign(R, nullptr);
            }
        }
        row_need.assign(R, 0);
        size_t row_i = 0;
        for (size_t li = 0; li < R; li++) {
            const Bid b = lv[li];
            BlockNode &n = nodes_[b];
            const BlockNode &s = nodes_[n.src];
            LinDesc d{};
            d.first_term = (uint32_t)tl.terms.size();
            int64_t c2 = 0;
            uint64_t &need = row_need[row_i++];
            if (s.kind == BlockNode::MAT) {
                tl.terms.push_back({s.dev, 1});
                d.n_terms = 1;
                need = std::max(need, s.ready_tick);
                c2 = s.var;
            } else if (s.kind == BlockNode::LIN) {
                for (const Term &t : s.terms) {
                    const BlockNode &tb = nodes_[t.blk];
                    if (tb.kind != BlockNode::MAT || !tb.dev)
                        return ctx.fail(-3, "internal: lincomb term not materialised at its level");
                    tl.terms.push_back({tb.dev, t.coef});
                    need = std::max(need, tb.ready_tick);
                }
                c2 = lin_c2(s.terms);                        // extractions of one shared rotation count as fully correlated
                d.n_terms = (uint32_t)s.terms.size();
                d.konst_body = (uint64_t)(s.konst & 31) << DELTA_LOG;
            } else {
                return ctx.fail(-3, "internal: PBS source is neither MAT nor LIN (node " + std::to_string(b) + " level " +
                                        std::to_string(n.level) + " lut " + std::to_string(n.lut) + ", source " +
                                        std::to_string(n.src) + " kind " + std::to_string((int)s.kind) + " level " +
                                        std::to_string(s.level) + " refs " + std::to_string(s.refs) + ")");
            }
            stats.max_input_sum_c2 = std::max<uint64_t>(stats.max_input_sum_c2, (uint64_t)c2);
            if (c2 > 64 && std::getenv("FHS_DEBUG_C2")) {
                std::fprintf(stderr, "c2=%lld lut=%d terms:", (long long)c2, (int)n.lut);
                for (const Term &t : s.terms) std::fprintf(stderr, " %lld*b%u", (long long)t.coef, t.blk);
                std::fprintf(stderr, "\n");
            }
            tl.descs.push_back(d);
            tl.lut.push_back(n.lut);
            if (capture_max_rows && capture_live)             // level: filled in when the row runs (run_tick)
                tl.recs.push_back(CaptureRec{0, (uint32_t)li, n.lut, d.n_terms, c2, s.kind == BlockNode::LIN ? s.konst : 0,
                                             (uint32_t)R});
            uint64_t *o = alloc_block();
            if (!o) return ctx.fail(-2, "device block pool exhausted (hipMalloc failed)");
            tl.out.push_back(o);
        }
        for (ShareRow &f : followers) {                      // a follower's own block; its leader also stores the body polynomial
            f.out = alloc_block();
            if (!f.out) return ctx.fail(-2, "device block pool exhausted (hipMalloc failed)");
            if (!tl.body[f.lead_row]) {
                tl.body[f.lead_row] = alloc_block();
                if (!tl.body[f.lead_row]) return ctx.fail(-2, "device block pool exhausted (hipMalloc failed)");
            }
        }
        tl.ext = followers;
        if (run_now) {
            // nothing is scheduled (flush drained it): enqueue this level right away, then plan the next one meanwhile
            std::vector<TickLevel> one;
            one.push_back(std::move(tl));
            if (int rc = run_tick(one, level_parallel && ctx.dist.active())) return rc;
            std::vector<uint32_t> rot_of(followers.empty() ? 0 : R, 0);
            for (const ShareRow &f : followers)
                if (!rot_of[f.lead_row]) rot_of[f.lead_row] = ++rot_counter_ ? rot_counter_ : ++rot_counter_;
            for (size_t k = 0; k < lv.size(); k++) {
                BlockNode &n = nodes_[lv[k]];
                const Bid src = n.src;
                n.kind = BlockNode::MAT;
                n.dev = k < R ? one[0].out[k] : followers[k - R].out;
                if (!followers.empty()) n.rot = rot_of[k < R ? k : followers[k - R].lead_row];
                n.src = 0;
                n.level = 0;
                release(src);                                // immediate recycling is safe: stream order
            }
            continue;
        }
        // ---- list scheduling at ROW granularity ------------------------------------------------------------------
        // A row runs at the first tick after the one that produces its last input (and not before the tick after the
        // previous level's).  Rows of one level may therefore sit on different ticks: the rows that consume a result
        // which was itself sent one tick later (round alignment below, or a dependent job) follow it, the others stay.
        const uint64_t t0 = tick + 1;
        uint64_t base = ~0ull, last = 0;
        for (size_t k = 0; k < R; k++) {
            row_need[k] = std::max(t0, row_need[k] + 1);      // now: the row's own tick
            base = std::min(base, row_need[k]);
            last = std::max(last, row_need[k]);
        }
        tick = base;
        // round alignment: with `cur` rows already scheduled for the base tick, keep only as many of this level's on-time
        // rows as fill whole rounds of the persistent kernel; the excess (less than one round) runs one tick later with
        // whatever is scheduled there -- only ITS consumers follow it, the rest of the next level does not wait
        size_t n_base = 0;
        for (size_t k = 0; k < R; k++) n_base += row_need[k] == base;
        if (balance_slots) {
            size_t cur = 0;
            auto it = sched_.find(base);
            if (it != sched_.end())
                for (const TickLevel &l : it->second) cur += l.descs.size();
            const size_t total = cur + n_base, rem = total % balance_slots;
            const size_t rounds = (total + balance_slots - 1) / balance_slots;
            size_t defer = 0;
            // only when the last round would be less than ~2/3 full overall: a group that already fills 95 % of its rounds
            // is left alone (the split costs its consumers one tick)
            // (jobs scheduled by hand share their ticks with jobs still to come: there only a level that is at least one
            // round wide by itself is split; a streaming flush knows the whole population of the tick)
            if ((stream_pump ? total : n_base) >= balance_slots && rem && rem < n_base &&
                total * 100 < rounds * balance_slots * 95)
                defer = rem;
            // a sliver in front of a wide level (what an automatic partial flush leaves of a level: 16 388 = 2 x 8192 + 4)
            // would pay one whole bootstrap alone: it joins the next tick, where the rows of the next level that do not
            // consume it run anyway
            auto nx = std::next(lvit);
            if (stream_pump && !defer && total * 8 < balance_slots && nx != by_level.end() &&
                nx->second.size() >= 2 * balance_slots)
                defer = n_base;
            for (size_t k = R; k-- > 0 && defer;)
                if (row_need[k] == base) { row_need[k] = base + 1; defer--; }
            for (size_t k = 0; k < R; k++) last = std::max(last, row_need[k]);
        }
        last_sched_tick_ = std::max(last_sched_tick_, last);  // before the releases below
        std::vector<uint32_t> rot_of(tl.ext.empty() ? 0 : R, 0);
        for (const ShareRow &f : tl.ext)
            if (!rot_of[f.lead_row]) rot_of[f.lead_row] = ++rot_counter_ ? rot_counter_ : ++rot_counter_;
        for (size_t k = 0; k < lv.size(); k++) {
            BlockNode &n = nodes_[lv[k]];
            const Bid src = n.src;
            n.kind = BlockNode::MAT;
            n.dev = k < R ? tl.out[k] : tl.ext[k - R].out;
            if (!tl.ext.empty()) n.rot = rot_of[k < R ? k : tl.ext[k - R].lead_row];
            n.src = 0;
            n.level = 0;
            n.ready_tick = k < R ? row_need[k] : row_need[tl.ext[k - R].lead_row];   // a follower is ready with its leader
            release(src);
        }
        // hand the rows to their ticks (ascending): one TickLevel per job and tick, rows of several levels merged
        std::vector<uint64_t> ticks(row_need.begin(), row_need.begin() + R);
        std::sort(ticks.begin(), ticks.end());
        ticks.erase(std::unique(ticks.begin(), ticks.end()), ticks.end());
        for (uint64_t tk : ticks) {
            std::vector<TickLevel> &slot = sched_[tk];
            TickLevel *dst = nullptr;
            for (TickLevel &l : slot)
                if (l.job == job) dst = &l;
            if (!dst) {
                slot.emplace_back();
                dst = &slot.back();
                dst->job = job;
            }
            if (ticks.size() == 1 && dst->descs.empty()) {    // the common case: the whole level on one tick
                tl.job = job;
                *dst = std::move(tl);
                break;
            }
            std::vector<uint32_t> moved(tl.ext.empty() ? 0 : R, 0);   // leader row -> its position in dst
            for (size_t k = 0; k < R; k++) {
                if (row_need[k] != tk) continue;
                LinDesc d = tl.descs[k];
                const uint32_t f = d.first_term;
                d.first_term = (uint32_t)dst->terms.size();
                dst->terms.insert(dst->terms.end(), tl.terms.begin() + f, tl.terms.begin() + f + d.n_terms);
                if (!tl.ext.empty()) {
                    moved[k] = (uint32_t)dst->descs.size();
                    dst->body.resize(dst->descs.size(), nullptr);
                    dst->body.push_back(tl.body[k]);
                }
                dst->descs.push_back(d);
                dst->lut.push_back(tl.lut[k]);
                dst->out.push_back(tl.out[k]);
                if (tl.recs.size() == tl.descs.size()) dst->recs.push_back(tl.recs[k]);
            }
            for (const ShareRow &f : tl.ext)                       // followers travel with their leader's tick
                if (row_need[f.lead_row] == tk) dst->ext.push_back({moved[f.lead_row], f.K, f.out});
        }
        // streaming flush: every tick up to `base` is complete now (later levels cannot reach back) -- enqueue them while
        // the host plans the next level
        if (stream_pump)
            while (!sched_.empty() && sched_.begin()->first <= base)
                if (int rc = pump(1)) return rc;
    }
    return 0;
}

int Engine::pump(size_t n_ticks) {
    for (size_t it = 0; it < n_ticks && !sched_.empty(); it++) {
        auto first = sched_.begin();
        const uint64_t tick = first->first;
        std::vector<TickLevel> levels = std::move(first->second);
        sched_.erase(first);
        if (int rc = run_tick(levels)) return rc;
        next_tick_ = tick + 1;
        // blocks freed while ticks <= `tick` were still pending are safe to hand out now
        while (!free_after_.empty() && free_after_.begin()->first <= tick) {
            for (uint64_t *p : free_after_.begin()->second) free_blocks_.push_back(p);
            free_after_.erase(free_after_.begin());
        }
    }
    if (sched_.empty() && last_sched_tick_ >= next_tick_) next_tick_ = last_sched_tick_ + 1;
    return 0;
}

int Engine::upload_plan(void *d_dst, const void *src, size_t bytes) {
    Staging *st = nullptr;
    for (Staging &c : staging_)
        if (!c.busy || hipEventQuery(c.done) == hipSuccess) { c.busy = false; if (!st || c.cap >= bytes) st = &c; }
    if (!st && staging_.size() < 8) { staging_.emplace_back(); st = &staging_.back(); }
    if (!st) {                                           // every buffer is in flight: wait for the oldest
        st = &staging_[0];
        if (hipEventSynchronize(st->done) != hipSuccess) return ctx.fail(-2, "staging wait failed");
        st->busy = false;
    }
    if (st->cap < bytes) {
        if (st->p) (void)hipHostFree(st->p);
        st->p = nullptr; st->cap = 0;
        const size_t want = std::max(bytes + bytes / 2, (size_t)1 << 20);
        if (hipHostMalloc(&st->p, want) != hipSuccess) return ctx.fail(-2, "hipHostMalloc (plan staging) failed");
        st->cap = want;
    }
    if (!st->done && hipEventCreateWithFlags(&st->done, hipEventDisableTiming) != hipSuccess)
        return ctx.fail(-2, "hipEventCreate failed");
    std::memcpy(st->p, src, bytes);
    hipError_t e = hipMemcpyAsync(d_dst, st->p, bytes, hipMemcpyHostToDevice, ctx.stream);
    if (e == hipSuccess) e = hipEventRecord(st->done, ctx.stream);
    if (e != hipSuccess) return ctx.hip_fail(e, "plan upload");
    st->busy = true;
    return 0;
}

// one launch group over the union of the job levels scheduled for a tick
int Engine::run_tick(std::vector<TickLevel> &levels, bool sharded) {
    size_t width = 0, n_terms = 0;
    for (auto &l : levels) { width += l.descs.size(); n_terms += l.terms.size(); }
    if (width == 0) return 0;
    for (auto &l : levels) {
        stats.levels += 1;
        if (stats.level_widths.size() < (1u << 20)) stats.level_widths.push_back((uint32_t)l.descs.size());
        stats.max_level_width = std::max<uint64_t>(stats.max_level_width, l.descs.size());
    }
    const size_t world = sharded ? (size_t)ctx.dist.world : 1, rank = sharded ? (size_t)ctx.dist.rank : 0;
    const size_t cap = (width + world - 1) / world;
    const size_t lo = std::min(width, rank * cap), hi = std::min(width, lo + cap), cnt = hi - lo;
    stats.pbs_executed += cnt;
    size_t n_ext = 0;
    for (auto &l : levels) n_ext += l.ext.size();
    if (n_ext && sharded) return ctx.fail(-3, "internal: rotation sharing in a level-parallel launch group");
    stats.pbs_extracted += n_ext;
    if (stats.group_rows.size() < (1u << 20)) stats.group_rows.push_back((uint32_t)cnt);
    if (planner && trace_plan) {
        for (auto &l : levels) {
            for (size_t k = 0; k < l.descs.size(); k++) {
                const LinDesc &d = l.descs[k];
                trace_.push_back(TR_ROW);
                trace_.push_back((uint64_t)(uintptr_t)l.out[k]);
                trace_.push_back(l.lut[k]);
                trace_.push_back(d.konst_body >> DELTA_LOG);
                trace_.push_back(d.n_terms);
                for (uint32_t t = 0; t < d.n_terms; t++) {
                    trace_.push_back((uint64_t)(uintptr_t)l.terms[d.first_term + t].src);
                    trace_.push_back((uint64_t)l.terms[d.first_term + t].coef);
                }
            }
            for (const ShareRow &f : l.ext) {
                trace_.push_back(TR_EXT);
                trace_.push_back((uint64_t)(uintptr_t)l.out[f.lead_row]);
                trace_.push_back((uint64_t)(uintptr_t)f.out);
                trace_.push_back(f.K);
            }
        }
        trace_.push_back(TR_GROUP_END);
        trace_.push_back(width);
    }
    if (planner) {
        // nothing runs, but the exchange of a level-parallel launch group is accounted for like Dist::all_gather does
        if (sharded) { ctx.dist.n_gathers++; ctx.dist.bytes_sent += cap * BIG_CT * 8; }
        for (auto &l : levels)
            for (uint64_t *b : l.body)
                if (b) free_block(b);
        return 0;
    }
    if (hipSetDevice(ctx.device) != hipSuccess) return ctx.fail(-2, "hipSetDevice failed");
    const size_t off_desc = 0;
    const size_t off_terms = off_desc + width * sizeof(LinDesc);
    const size_t off_lut = off_terms + n_terms * sizeof(LinTerm);
    const size_t off_out = (off_lut + width * 4 + 15) & ~(size_t)15;
    // rotation sharing: [body pointer per row | one ExtractDesc per follower] behind the output pointers
    const size_t off_body = off_out + width * sizeof(uint64_t *);
    const size_t off_ext = off_body + (n_ext ? width * sizeof(uint64_t *) : 0);
    const size_t total = off_ext + n_ext * sizeof(ExtractDesc);
    std::vector<uint8_t> host(total);
    LinDesc *hd = reinterpret_cast<LinDesc *>(host.data() + off_desc);
    LinTerm *ht = reinterpret_cast<LinTerm *>(host.data() + off_terms);
    uint32_t *hl = reinterpret_cast<uint32_t *>(host.data() + off_lut);
    uint64_t **ho = reinterpret_cast<uint64_t **>(host.data() + off_out);
    uint64_t **hb = reinterpret_cast<uint64_t **>(host.data() + off_body);
    ExtractDesc *hx = reinterpret_cast<ExtractDesc *>(host.data() + off_ext);
    size_t di = 0, ti = 0, xi = 0;
    for (auto &l : levels) {
        for (size_t k = 0; k < l.descs.size(); k++) {
            LinDesc d = l.descs[k];
            d.first_term += (uint32_t)ti;
            hd[di + k] = d;
            hl[di + k] = l.lut[k];
            ho[di + k] = l.out[k];
            if (n_ext) hb[di + k] = k < l.body.size() ? l.body[k] : nullptr;
        }
        for (const ShareRow &f : l.ext) hx[xi++] = ExtractDesc{l.out[f.lead_row], l.body[f.lead_row], f.out, f.K, 0};
        std::memcpy(ht + ti, l.terms.data(), l.terms.size() * sizeof(LinTerm));
        di += l.descs.size();
        ti += l.terms.size();
    }
    hipError_t e = tick_buf_.cap >= total ? hipSuccess : hipStreamSynchronize(ctx.stream);
    if (e == hipSuccess) e = tick_buf_.reserve(total);
    if (e == hipSuccess && batch_in_.cap < width * BIG_CT * 8) {
        e = hipStreamSynchronize(ctx.stream);
        if (e == hipSuccess) e = batch_in_.reserve(width * BIG_CT * 8);
    }
    if (e == hipSuccess && ctx.ks_buf.cap < width * SMALL_CT * 8) {
        e = hipStreamSynchronize(ctx.stream);
        if (e == hipSuccess) e = ctx.ks_buf.reserve(width * SMALL_CT * 8);
    }
    if (e != hipSuccess) return ctx.hip_fail(e, "tick buffers");
    // stream-ordered after the previous tick's kernels, which read the old contents; pinned staging: the host goes on
    if (int rc = upload_plan(tick_buf_.ptr, host.data(), total)) return rc;
    const uint8_t *dp = tick_buf_.as<uint8_t>();
    const LinDesc *d_desc = reinterpret_cast<const LinDesc *>(dp + off_desc);
    const LinTerm *d_terms = reinterpret_cast<const LinTerm *>(dp + off_terms);
    const uint32_t *d_lut = reinterpret_cast<const uint32_t *>(dp + off_lut);
    uint64_t *const *d_out = reinterpret_cast<uint64_t *const *>(dp + off_out);
    if (!sharded) {
        e = launch_lincomb(d_desc, d_terms, batch_in_.as<uint64_t>(), (int)width, ctx.stream);
        if (e != hipSuccess) return ctx.hip_fail(e, "lincomb launch");
        if (capture_max_rows && capture_live) {
            // debug only: a strided sample of every job level's PBS inputs in this launch group goes to the host
            // (synchronous copies); the rows are the ones the production path bootstraps, shared rotations included
            size_t row0 = 0, lvl = stats.levels - levels.size();
            for (auto &l : levels) {
                const size_t cnt_l = l.descs.size();
                if (l.recs.size() == cnt_l) {
                    const size_t stride = (cnt_l + capture_max_rows - 1) / capture_max_rows;
                    for (size_t i = 0; i < cnt_l; i += stride) {
                        const size_t at = capture_rows.size();
                        capture_rows.resize(at + BIG_CT);
                        e = hipMemcpyAsync(capture_rows.data() + at, batch_in_.as<uint64_t>() + (row0 + i) * BIG_CT, BIG_CT * 8,
                                           hipMemcpyDeviceToHost, ctx.stream);
                        if (e == hipSuccess) e = hipStreamSynchronize(ctx.stream);
                        if (e != hipSuccess) return ctx.hip_fail(e, "capture download");
                        CaptureRec r = l.recs[i];
                        r.level = (uint32_t)lvl;
                        capture_recs.push_back(r);
                    }
                }
                row0 += cnt_l;
                lvl++;
            }
        }
        if (int rc = ctx.keyswitch(batch_in_.as<uint64_t>(), width, ctx.stream)) return rc;
        uint64_t *const *d_body = n_ext ? reinterpret_cast<uint64_t *const *>(dp + off_body) : nullptr;
        if (int rc = ctx.blind_rotate(ctx.ks_buf.as<uint64_t>(), d_lut, d_luts_, nullptr, d_out, width, ctx.stream, d_body)) return rc;
        if (n_ext) {
            // the followers: further sample extractions of their leaders' accumulators, then the body polynomials' blocks
            // go back to the pool (stream order: whoever gets them next runs behind this kernel)
            e = launch_extract_shift(reinterpret_cast<const ExtractDesc *>(dp + off_ext), (int)n_ext, ctx.stream);
            if (e != hipSuccess) return ctx.hip_fail(e, "extract launch");
            for (auto &l : levels)
                for (uint64_t *b : l.body)
                    if (b) free_block(b);
        }
        if (!last_group_done_) (void)hipEventCreateWithFlags(&last_group_done_, hipEventDisableTiming);
        if (last_group_done_) (void)hipEventRecord(last_group_done_, ctx.stream);
        return 0;
    }
    // level-parallel: own slice -> dense exchange buffer -> all-gather -> scatter (all enqueued, no host wait with RCCL)
    if (ctx.xchg_send.cap < cap * BIG_CT * 8 || ctx.xchg_recv.cap < world * cap * BIG_CT * 8) {
        e = hipStreamSynchronize(ctx.stream);
        if (e == hipSuccess) e = ctx.xchg_send.reserve(cap * BIG_CT * 8);
        if (e == hipSuccess) e = ctx.xchg_recv.reserve(world * cap * BIG_CT * 8);
        if (e != hipSuccess) return ctx.hip_fail(e, "level exchange buffers");
    }
    if (cnt) {
        e = launch_lincomb(d_desc + lo, d_terms, batch_in_.as<uint64_t>(), (int)cnt, ctx.stream);
        if (e != hipSuccess) return ctx.hip_fail(e, "lincomb launch");
        if (int rc = ctx.keyswitch(batch_in_.as<uint64_t>(), cnt, ctx.stream)) return rc;
        if (int rc = ctx.blind_rotate(ctx.ks_buf.as<uint64_t>(), d_lut + lo, d_luts_, ctx.xchg_send.as<uint64_t>(), nullptr, cnt,
                                      ctx.stream))
            return rc;
    }
    if (int rc = ctx.dist.all_gather(ctx.xchg_send.ptr, ctx.xchg_recv.ptr, cap * BIG_CT * 8, ctx.stream, ctx.err)) return rc;
    e = launch_scatter_blocks(ctx.xchg_recv.as<uint64_t>(), d_out, (int)width, ctx.stream);
    if (e != hipSuccess) return ctx.hip_fail(e, "scatter launch");
    if (!last_group_done_) (void)hipEventCreateWithFlags(&last_group_done_, hipEventDisableTiming);
    if (last_group_done_) (void)hipEventRecord(last_group_done_, ctx.stream);
    return 0;
}

int Engine::gather_blocks(const Bid *local, size_t n, std::vector<Bid> &out) {
    out.clear();
    if (!ctx.dist.active()) return ctx.fail(-3, "no distributed transport (fhs_dist_init)");
    if (n == 0) return 0;
    const size_t world = (size_t)ctx.dist.world, row = (size_t)BIG_CT * 8;
    // The partial blocks must be in the stream before the exchange is enqueued -- nothing more: unsubmitted work is
    // submitted as a job and only the ticks that produce these blocks are pumped, so jobs of later requests that are
    // scheduled (level-skewed batching) stay scheduled.  Level-parallel mode and linear-combination blocks drain.
    int rc = 0;
    bool drain = level_parallel;
    if (!drain && !pending_.empty()) rc = submit();
    if (rc) return rc;
    uint64_t need = 0;
    for (size_t k = 0; k < n; k++) {
        const BlockNode &b = nodes_[local[k]];
        if (b.kind == BlockNode::LIN || b.kind == BlockNode::PBS) drain = true;
        else if (b.kind == BlockNode::MAT) need = std::max(need, b.ready_tick);
    }
    if (drain) rc = flush();
    else
        while (!rc && !sched_.empty() && sched_.begin()->first <= need) rc = pump(1);
    if (rc) return rc;
    // Extractions of ONE shared rotation are correlated (lin_c2); a gathered block comes back as a fresh node on every
    // rank, so that relation would be lost in the exchange.  The sharded operations never send two members of one group
    // (their partials are AND / OR results over different inputs); a caller-supplied set that does is refused rather than
    // under-charged.  The rank's OWN blocks keep their bookkeeping across the exchange (below).
    for (size_t k = 0; k < n; k++) {
        const BlockNode &a = nodes_[local[k]];
        if (a.kind != BlockNode::MAT || !a.rot) continue;
        for (size_t q = 0; q < k; q++)
            if (nodes_[local[q]].kind == BlockNode::MAT && nodes_[local[q]].rot == a.rot)
                return ctx.fail(-3, "all-gather of two extractions of one shared blind rotation: their noise correlation would "
                                    "be lost on the receivers (refresh one of them, or fhs_set_rotation_sharing(ctx, 0))");
    }
    auto keep_own = [&](std::vector<Bid> &got) {             // own slice: the figures this rank already tracks
        const size_t r0 = (size_t)ctx.dist.rank * n;
        for (size_t k = 0; k < n && r0 + k < got.size(); k++) {
            const BlockNode &a = nodes_[local[k]];
            if (a.kind != BlockNode::MAT) continue;
            nodes_[got[r0 + k]].var = std::max(nodes_[got[r0 + k]].var, a.var);
            nodes_[got[r0 + k]].rot = a.rot;
        }
    };
    if (planner) {
        // a planner context exchanges nothing: count the all-gather like Dist::all_gather does and hand back world * n
        // fresh single-output blocks, so that the combine DAG can be recorded and levelised (multi-GPU projections)
        ctx.dist.n_gathers++;
        ctx.dist.bytes_sent += n * row;
        for (size_t i = 0; i < world * n; i++) out.push_back(from_device(nullptr));
        keep_own(out);
        return 0;
    }
    if (hipSetDevice(ctx.device) != hipSuccess) return ctx.fail(-2, "hipSetDevice failed");
    if (ctx.xchg_send.cap < n * row || ctx.xchg_recv.cap < world * n * row) {
        hipError_t e = hipStreamSynchronize(ctx.stream);
        if (e == hipSuccess) e = ctx.xchg_send.reserve(n * row);
        if (e == hipSuccess) e = ctx.xchg_recv.reserve(world * n * row);
        if (e != hipSuccess) return ctx.hip_fail(e, "exchange buffers");
    }
    for (size_t k = 0; k < n; k++)
        if ((rc = copy_block_to_device(local[k], ctx.xchg_send.as<uint64_t>() + k * BIG_CT, false, false))) return rc;
    if ((rc = ctx.dist.all_gather(ctx.xchg_send.ptr, ctx.xchg_recv.ptr, n * row, ctx.stream, ctx.err))) return rc;
    out.reserve(world * n);
    for (size_t i = 0; i < world * n; i++) {
        Bid b = from_device(ctx.xchg_recv.as<uint64_t>() + i * BIG_CT);
        if (!b) {
            for (Bid x : out) release(x);
            out.clear();
            return ctx.fail(-2, "import of a gathered block failed");
        }
        out.push_back(b);
    }
    keep_own(out);
    return 0;
}

// Builds the plan for every pending level and uploads it; nodes get their output blocks here.
int Engine::plan_flush() {
    plan_.levels.clear();
    plan_.recs.clear();
    if (pending_.empty()) return 0;
    if (!planner) {
        if (!ctx.key_loaded) return ctx.fail(-3, "server key not loaded");
        if (hipSetDevice(ctx.device) != hipSuccess) return ctx.fail(-2, "hipSetDevice failed");
    }

    std::map<uint32_t, std::vector<Bid>> by_level;
    for (const Pend &p : pending_)
        if (nodes_[p.id].kind == BlockNode::PBS && nodes_[p.id].gen == p.gen) by_level[nodes_[p.id].level].push_back(p.id);
    pending_.clear();
    n_depth1_ = 0;
    n_depth1_solo_ = 0;
    depth1_keys_.clear();
    if (by_level.empty()) return 0;

    std::vector<LevelPlan> &levels = plan_.levels;
    std::vector<LinDesc> descs;
    std::vector<LinTerm> terms;
    std::vector<uint32_t> lut_idx;
    std::vector<uint64_t *> out_ptrs;
    size_t max_width = 0;

    for (auto &kv : by_level) {
        std::vector<Bid> &lv = kv.second;
        const size_t first = descs.size();
        for (Bid b : lv) {
            BlockNode &n = nodes_[b];
            const BlockNode &s = nodes_[n.src];
            LinDesc d{};
            d.first_term = (uint32_t)terms.size();
            if (s.kind == BlockNode::MAT) {
                terms.push_back({s.dev, 1});
                d.n_terms = 1;
                d.konst_body = 0;
            } else if (s.kind == BlockNode::LIN) {
                for (const Term &t : s.terms) {
                    const BlockNode &tb = nodes_[t.blk];
                    if (tb.kind != BlockNode::MAT || !tb.dev)
                        return ctx.fail(-3, "internal: lincomb term not materialised at its level");
                    terms.push_back({tb.dev, t.coef});
                }
                d.n_terms = (uint32_t)s.terms.size();
                d.konst_body = (uint64_t)(s.konst & 31) << DELTA_LOG;
            } else {
                return ctx.fail(-3, "internal: PBS source is neither MAT nor LIN");
            }
            descs.push_back(d);
            lut_idx.push_back(n.lut);
            {   // noise bookkeeping: sum of squared coefficients of the (flattened) linear combination entering this
                // bootstrap, i.e. its noise variance in units of one bootstrap output's (uploads counted like outputs)
                int64_t c2 = 0;
                if (s.kind == BlockNode::LIN) c2 = lin_c2(s.terms);
                else c2 = s.var;
                stats.max_input_sum_c2 = std::max<uint64_t>(stats.max_input_sum_c2, (uint64_t)c2);
                if (c2 > 64 && std::getenv("FHS_DEBUG_C2")) {
                    std::fprintf(stderr, "c2=%lld lut=%d terms:", (long long)c2, (int)n.lut);
                    for (const Term &t : s.terms) std::fprintf(stderr, " %lld*b%u", (long long)t.coef, t.blk);
                    std::fprintf(stderr, "\n");
                }
                if (capture_max_rows)
                    plan_.recs.push_back(CaptureRec{(uint32_t)levels.size(), (uint32_t)(descs.size() - 1 - first), n.lut,
                                                    d.n_terms, c2, s.kind == BlockNode::LIN ? s.konst : 0,
                                                    (uint32_t)lv.size()});
            }
            uint64_t *o = alloc_block();
            if (!o) return ctx.fail(-2, "device block pool exhausted (hipMalloc failed)");
            out_ptrs.push_back(o);
        }
        // commit the level: nodes become materialised, their inputs can be recycled by later levels
        for (size_t k = 0; k < lv.size(); k++) {
            BlockNode &n = nodes_[lv[k]];
            const Bid src = n.src;
            n.kind = BlockNode::MAT;
            n.dev = out_ptrs[first + k];
            n.src = 0;
            n.level = 0;
            release(src);
        }
        levels.push_back({first, lv.size()});
        max_width = std::max(max_width, lv.size());
    }

    plan_.max_width = max_width;
    if (planner) return 0;                           // nothing to upload: the plan only feeds the statistics
    // one upload: [descs | terms | lut_idx | out_ptrs]
    const size_t off_desc = 0;
    const size_t off_terms = off_desc + descs.size() * sizeof(LinDesc);
    const size_t off_lut = off_terms + terms.size() * sizeof(LinTerm);
    const size_t off_out = (off_lut + lut_idx.size() * 4 + 15) & ~(size_t)15;
    const size_t total = off_out + out_ptrs.size() * sizeof(uint64_t *);
    std::vector<uint8_t> host(total);
    std::memcpy(host.data() + off_desc, descs.data(), descs.size() * sizeof(LinDesc));
    std::memcpy(host.data() + off_terms, terms.data(), terms.size() * sizeof(LinTerm));
    std::memcpy(host.data() + off_lut, lut_idx.data(), lut_idx.size() * 4);
    std::memcpy(host.data() + off_out, out_ptrs.data(), out_ptrs.size() * sizeof(uint64_t *));
    // the previous flush may still be reading the old plan buffer: order through the stream
    hipError_t e = plan_buf_.cap >= total ? hipSuccess : hipStreamSynchronize(ctx.stream);
    if (e == hipSuccess) e = plan_buf_.reserve(total);
    if (e != hipSuccess) return ctx.hip_fail(e, "plan buffer");
    if (int rc = upload_plan(plan_buf_.ptr, host.data(), total)) return rc;
    if (batch_in_.cap < max_width * BIG_CT * 8) {
        e = hipStreamSynchronize(ctx.stream);
        if (e == hipSuccess) e = batch_in_.reserve(max_width * BIG_CT * 8);
        if (e != hipSuccess) return ctx.hip_fail(e, "batch buffer");
    }
    if (ctx.ks_buf.cap < max_width * SMALL_CT * 8) {
        e = hipStreamSynchronize(ctx.stream);
        if (e == hipSuccess) e = ctx.ks_buf.reserve(max_width * SMALL_CT * 8);
        if (e != hipSuccess) return ctx.hip_fail(e, "ks buffer");
    }
    // pageable host memory: hipMemcpyAsync has consumed `host` when it returns

    plan_.off_desc = off_desc;
    plan_.off_terms = off_terms;
    plan_.off_lut = off_lut;
    plan_.off_out = off_out;
    plan_.max_width = max_width;
    return 0;
}

// Runs items [lo, hi) of level k.  dense_out == nullptr: results go straight to the nodes' blocks;
// otherwise they are written as rows 0..hi-lo of dense_out (distributed mode, before the all-gather).
int Engine::exec_level(size_t k, size_t lo, size_t hi, uint64_t *dense_out) {
    if (k >= plan_.levels.size()) return ctx.fail(-1, "level index out of range");
    const LevelPlan &lp = plan_.levels[k];
    if (lo > hi || hi > lp.count) return ctx.fail(-1, "slice out of range");
    const size_t cnt = hi - lo;
    if (cnt == 0) return 0;
    if (stats.group_rows.size() < (1u << 20)) stats.group_rows.push_back((uint32_t)cnt);
    if (planner) {
        stats.pbs_executed += cnt;
        if (lo == 0 || dense_out) {
            stats.levels += 1;
            if (stats.level_widths.size() < (1u << 20)) stats.level_widths.push_back((uint32_t)lp.count);
            stats.max_level_width = std::max<uint64_t>(stats.max_level_width, lp.count);
        }
        return 0;
    }
    if (hipSetDevice(ctx.device) != hipSuccess) return ctx.fail(-2, "hipSetDevice failed");
    const uint8_t *dp = plan_buf_.as<uint8_t>();
    const LinDesc *d_desc = reinterpret_cast<const LinDesc *>(dp + plan_.off_desc) + lp.first + lo;
    const LinTerm *d_terms = reinterpret_cast<const LinTerm *>(dp + plan_.off_terms);
    const uint32_t *d_lut = reinterpret_cast<const uint32_t *>(dp + plan_.off_lut) + lp.first + lo;
    uint64_t *const *d_out = reinterpret_cast<uint64_t *const *>(dp + plan_.off_out) + lp.first + lo;
    hipError_t e = launch_lincomb(d_desc, d_terms, batch_in_.as<uint64_t>(), (int)cnt, ctx.stream);
    if (e != hipSuccess) return ctx.hip_fail(e, "lincomb launch");
    if (capture_max_rows && plan_.recs.size() >= lp.first + hi) {
        // debug only: a strided sample of this level's PBS inputs goes to the host (synchronous copies)
        const size_t stride = (cnt + capture_max_rows - 1) / capture_max_rows;
        for (size_t i = 0; i < cnt; i += stride) {
            const size_t at = capture_rows.size();
            capture_rows.resize(at + BIG_CT);
            e = hipMemcpyAsync(capture_rows.data() + at, batch_in_.as<uint64_t>() + i * BIG_CT, BIG_CT * 8,
                               hipMemcpyDeviceToHost, ctx.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx.stream);
            if (e != hipSuccess) return ctx.hip_fail(e, "capture download");
            capture_recs.push_back(plan_.recs[lp.first + lo + i]);
        }
    }
    if (int rc = ctx.keyswitch(batch_in_.as<uint64_t>(), cnt, ctx.stream)) return rc;
    if (int rc = ctx.blind_rotate(ctx.ks_buf.as<uint64_t>(), d_lut, d_luts_, dense_out, dense_out ? nullptr : d_out, cnt,
                                  ctx.stream))
        return rc;
    stats.pbs_executed += cnt;
    if (lo == 0 || dense_out) {
        stats.levels += 1;
        if (stats.level_widths.size() < (1u << 20)) stats.level_widths.push_back((uint32_t)lp.count);
        stats.max_level_width = std::max<uint64_t>(stats.max_level_width, lp.count);
    }
    return 0;
}

// Distributed mode: rows 0..count of d_all (the gathered level, item order) -> the nodes' blocks.
int Engine::commit_level(size_t k, const uint64_t *d_all) {
    if (k >= plan_.levels.size()) return ctx.fail(-1, "level index out of range");
    const LevelPlan &lp = plan_.levels[k];
    if (planner) {
        if (k + 1 == plan_.levels.size()) plan_.levels.clear();
        return 0;
    }
    if (hipSetDevice(ctx.device) != hipSuccess) return ctx.fail(-2, "hipSetDevice failed");
    uint64_t *const *d_out =
        reinterpret_cast<uint64_t *const *>(plan_buf_.as<uint8_t>() + plan_.off_out) + lp.first;
    hipError_t e = launch_scatter_blocks(d_all, d_out, (int)lp.count, ctx.stream);
    if (e != hipSuccess) return ctx.hip_fail(e, "scatter launch");
    if (k + 1 == plan_.levels.size()) plan_.levels.clear();
    return 0;
}

int Engine::materialize_lin(Bid b) {
    BlockNode &n = nodes_[b];
    if (n.kind != BlockNode::LIN) return 0;
    if (planner) return ctx.fail(-3, "planner context: nothing is computed");
    std::vector<LinTerm> terms;
    for (const Term &t : n.terms) {
        const BlockNode &tb = nodes_[t.blk];
        if (tb.kind != BlockNode::MAT) return ctx.fail(-3, "internal: lincomb term pending after flush");
        terms.push_back({tb.dev, t.coef});
    }
    LinDesc d{0, (uint32_t)terms.size(), (uint64_t)(n.konst & 31) << DELTA_LOG};
    const size_t total = sizeof(LinDesc) + terms.size() * sizeof(LinTerm);
    std::vector<uint8_t> host(total);
    std::memcpy(host.data(), &d, sizeof(d));
    std::memcpy(host.data() + sizeof(d), terms.data(), terms.size() * sizeof(LinTerm));
    hipError_t e = hipStreamSynchronize(ctx.stream);   // plan_buf_ may be in use by queued launches
    if (e == hipSuccess) e = plan_buf_.reserve(total);
    if (e == hipSuccess) e = hipMemcpyAsync(plan_buf_.ptr, host.data(), total, hipMemcpyHostToDevice, ctx.stream);
    if (e != hipSuccess) return ctx.hip_fail(e, "materialize upload");
    uint64_t *o = alloc_block();
    if (!o) return ctx.fail(-2, "device block pool exhausted");
    e = launch_lincomb(plan_buf_.as<LinDesc>(),
                       reinterpret_cast<const LinTerm *>(plan_buf_.as<uint8_t>() + sizeof(LinDesc)), o, 1,
                       ctx.stream);
    if (e != hipSuccess) return ctx.hip_fail(e, "lincomb launch");
    // the materialised block IS the linear combination: it keeps its noise (a download followed by further use of the
    // same handle must not look like a fresh bootstrap output to the bookkeeping)
    const int64_t v = sum_c2(b);
    std::vector<Term> old;
    old.swap(n.terms);
    n.kind = BlockNode::MAT;
    n.var = (uint16_t)std::min<int64_t>(std::max<int64_t>(v, 1), 65535);
    n.dev = o;
    n.level = 0;
    for (const Term &t : old) release(t.blk);
    return 0;
}

int Engine::read_block(Bid b, uint64_t *host_out) {
    if (planner) return ctx.fail(-3, "planner context: nothing is computed, there is nothing to download");
    (void)hipSetDevice(ctx.device);
    int rc = flush();
    if (rc) return rc;
    if (nodes_[b].kind == BlockNode::TRIV) {
        std::memset(host_out, 0, BIG_CT * 8);
        host_out[BIG_N] = (uint64_t)nodes_[b].triv << DELTA_LOG;
        return 0;
    }
    if (nodes_[b].kind == BlockNode::LIN && (rc = materialize_lin(b))) return rc;
    if (nodes_[b].kind != BlockNode::MAT) return ctx.fail(-3, "internal: block not materialised");
    hipError_t e = hipMemcpyAsync(host_out, nodes_[b].dev, BIG_CT * 8, hipMemcpyDeviceToHost, ctx.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx.stream);
    if (e != hipSuccess) return ctx.hip_fail(e, "download");
    return 0;
}

int Engine::read_many(const Bid *b, size_t count, uint64_t *host_out) {
    if (planner) return ctx.fail(-3, "planner context: nothing is computed, there is nothing to download");
    if (count == 0) return 0;
    (void)hipSetDevice(ctx.device);
    if (int rc = flush()) return rc;
    for (size_t i = 0; i < count; i++)                               // linear combinations become blocks of their own first
        if (nodes_[b[i]].kind == BlockNode::LIN)
            if (int rc = materialize_lin(b[i])) return rc;
    constexpr size_t MAX_BATCH = 2048;
    for (size_t done = 0; done < count;) {
        const size_t n = std::min(MAX_BATCH, count - done);
        if (count < 4 || !ensure_staging(n)) {
            // too few blocks to matter, or no staging memory: block by block
            for (size_t i = done; i < done + n; i++)
                if (int rc = read_block(b[i], host_out + i * BIG_CT)) return rc;
            done += n;
            continue;
        }
        if (upload_done_) (void)hipEventSynchronize(upload_done_);   // the staging buffer is shared with the uploads
        size_t n_dev = 0;
        for (size_t k = 0; k < n; k++) {
            const BlockNode &nd = nodes_[b[done + k]];
            if (nd.kind == BlockNode::TRIV) { upload_pin_[n * BIG_CT + k] = 0; continue; }
            if (nd.kind != BlockNode::MAT) return ctx.fail(-3, "internal: block not materialised");
            upload_pin_[n * BIG_CT + k] = (uint64_t)(uintptr_t)nd.dev;
            n_dev++;
        }
        // trivial blocks have no device row: point them at the first real block (their rows are overwritten on the host)
        uint64_t any = 0;
        for (size_t k = 0; k < n && !any; k++) any = upload_pin_[n * BIG_CT + k];
        if (n_dev) {
            for (size_t k = 0; k < n; k++)
                if (!upload_pin_[n * BIG_CT + k]) upload_pin_[n * BIG_CT + k] = any;
            hipError_t e = hipMemcpyAsync(upload_dev_ + n * BIG_CT, upload_pin_ + n * BIG_CT, n * 8, hipMemcpyHostToDevice, ctx.stream);
            if (e == hipSuccess)
                e = launch_gather_rows(reinterpret_cast<const uint64_t *const *>(upload_dev_ + n * BIG_CT), upload_dev_, (int)n, ctx.stream);
            if (e == hipSuccess) e = hipMemcpyAsync(upload_pin_, upload_dev_, n * BIG_CT * 8, hipMemcpyDeviceToHost, ctx.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx.stream);
            if (e != hipSuccess) return ctx.hip_fail(e, "download");
            std::memcpy(host_out + done * BIG_CT, upload_pin_, n * BIG_CT * 8);
        }
        for (size_t k = 0; k < n; k++) {
            const BlockNode &nd = nodes_[b[done + k]];
            if (nd.kind != BlockNode::TRIV) continue;
            uint64_t *row = host_out + (done + k) * BIG_CT;
            std::memset(row, 0, BIG_CT * 8);
            row[BIG_N] = (uint64_t)nd.triv << DELTA_LOG;
        }
        done += n;
    }
    return 0;
}

int Engine::copy_block_to_device(Bid b, uint64_t *d_out, bool wait, bool do_flush) {
    if (planner) return ctx.fail(-3, "planner context: nothing is computed");
    (void)hipSetDevice(ctx.device);
    int rc = do_flush ? flush() : 0;
    if (rc) return rc;
    hipError_t e;
    if (nodes_[b].kind == BlockNode::TRIV) {
        // body = triv << 59: only its high word is non-zero, written by a 32-bit memset (no host buffer in flight)
        const uint64_t body = (uint64_t)nodes_[b].triv << DELTA_LOG;
        e = hipMemsetAsync(d_out, 0, BIG_CT * 8, ctx.stream);
        if (e == hipSuccess)
            e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(reinterpret_cast<uint32_t *>(d_out + BIG_N) + 1),
                                  (int)(uint32_t)(body >> 32), 1, ctx.stream);
    } else {
        if (nodes_[b].kind == BlockNode::LIN && (rc = materialize_lin(b))) return rc;
        if (nodes_[b].kind != BlockNode::MAT) return ctx.fail(-3, "internal: block not materialised");
        e = hipMemcpyAsync(d_out, nodes_[b].dev, BIG_CT * 8, hipMemcpyDeviceToDevice, ctx.stream);
    }
    if (e == hipSuccess && wait) e = hipStreamSynchronize(ctx.stream);
    if (e != hipSuccess) return ctx.hip_fail(e, "export");
    return 0;
}

uint64_t Engine::new_char(const Bid b[4]) {
    uint64_t h;
    if (!free_chars_.empty()) {
        h = free_chars_.back();
        free_chars_.pop_back();
    } else {
        chars_.emplace_back();
        h = chars_.size();
    }
    CharRec &c = chars_[h - 1];
    for (int i = 0; i < 4; i++) c.b[i] = b[i];
    c.used = true;
    return h;
}

void Engine::free_char(uint64_t h) {
    CharRec &c = chars_[h - 1];
    for (int i = 0; i < 4; i++) release(c.b[i]);
    c.used = false;
    free_chars_.push_back(h);
}

}  // namespace fhs
