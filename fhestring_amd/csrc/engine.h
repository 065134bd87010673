// Lazy DAG engine: records shortint-block operations, folds constants, levelises the pending
// programmable bootstraps and launches each dependency level as one wide batch.
//
// The reference evaluates every FheAsciiChar op eagerly and blocking; behind that
// API a GPU would only ever see 4-8 PBS at a time.  Here an op only creates nodes; flush() plans
// all levels on the host, uploads the plan once and enqueues lincomb -> keyswitch -> blind-rotate
// per level on one stream with no host round trip in between.
#pragma once
#include <cstdint>
#include <map>
#include <unordered_map>
#include <utility>
#include <vector>

#include "context.h"
#include "luts.h"

namespace fhs {

using Bid = uint32_t;   // block id; 0 is never valid

struct Term {
    int64_t coef;
    Bid blk;
};

struct BlockNode {
    enum Kind : uint8_t { FREE, TRIV, MAT, LIN, PBS };
    Kind kind = FREE;
    uint8_t triv = 0;        // TRIV: value mod 32
    uint16_t lut = 0;        // PBS
    uint16_t var = 1;        // MAT: noise variance in units of one bootstrap output's (1 = a bootstrap output or a fresh
                             // upload; more for a linear combination that was materialised for a download / an export,
                             // or an upload declared noisier with fhs_char_set_noise)
    int32_t konst = 0;       // LIN: constant (mod 32)
    uint32_t level = 0;      // PBS depth since the last flush
    uint32_t refs = 0;
    uint32_t gen = 0;        // bumped every time the slot is handed out again (common-subexpression table keys)
    uint64_t nk = 0;         // PBS (fused mode): hash of (table, terms) WITHOUT the trivial constant -- rows with equal nk will
                             // share a blind rotation (plan_job); 0 = counts as a rotation of its own
    uint32_t rot = 0;        // MAT: rotation group (non-zero for the leader and the followers of ONE shared blind rotation):
                             // their noises are correlated, the bookkeeping adds their coefficients before squaring
    uint64_t ready_tick = 0; // MAT produced by a scheduled (not yet enqueued) job level: the tick that writes it
    Bid src = 0;             // PBS: input block
    uint64_t *dev = nullptr; // MAT: device ciphertext (2049 u64)
    std::vector<Term> terms; // LIN
};

struct EngineStats {
    uint64_t pbs_executed = 0, pbs_folded = 0, levels = 0, max_level_width = 0;
    uint64_t pbs_shared = 0;         // bootstraps not run because an identical one (same input combination, same LUT) exists
    uint64_t max_input_sum_c2 = 0;   // largest sum of squared coefficients of any executed bootstrap's input
    std::vector<uint32_t> level_widths;   // width of every dependency level executed since the last reset (capped)
    uint64_t pbs_extracted = 0;      // results obtained as a further sample extraction of a shared blind rotation (not in pbs_executed)
    std::vector<uint32_t> group_rows;     // rows THIS rank ran in every launch group (lincomb -> keyswitch -> blind rotation) since the last reset (capped)
};

class Engine {
  public:
    Context ctx;
    EngineStats stats;
    int mode = 0;   // 0 as written, 1 fused (string layer)
    // Planner context (fhs_ctx_create_planner): records DAGs, levelises them and keeps the statistics (PBS count, level
    // widths, noise bookkeeping) but owns no device and executes nothing; downloads fail.  Host logic only.
    bool planner = false;

    int on_key_loaded();
    void shutdown();

    // ---- block graph (all returned ids carry one reference owned by the caller) ----
    Bid triv(int v);
    Bid from_host(const uint64_t *ct);          // uploads 2049 words
    // `count` ciphertexts of 2049 words each, back to back on the host: one copy through a pinned staging buffer and one
    // scatter launch into the pool blocks instead of one pageable copy (~10 us) per block
    int from_host_many(const uint64_t *cts, size_t count, Bid *out);
    Bid from_device(const uint64_t *d_ct);      // D2D copy
    Bid lin(const Term *terms, size_t n, int konst);
    Bid pbs(Bid x, int lut);
    void retain(Bid b);
    void release(Bid b);
    const BlockNode &node(Bid b) const { return nodes_[b]; }
    bool is_triv(Bid b) const { return nodes_[b].kind == BlockNode::TRIV; }
    int triv_val(Bid b) const { return nodes_[b].triv; }
    // sum of squared coefficients of a block as a combination of bootstrap outputs / uploads (its noise variance in
    // units of one bootstrap output's): 0 trivial, 1 materialised or pending bootstrap, sum c^2 for a linear combination
    int64_t sum_c2(Bid b) const;
    // ... of a flattened combination.  The leader and the followers of one shared blind rotation are extractions of one
    // accumulator: their errors are correlated (measured: rho falls linearly with the constant difference from +0.45 to
    // -0.45, profiles/r05_rotation_sharing_rho.txt), so a group contributes sum c^2 + 1/2 ((sum |c|)^2 - sum c^2)
    int64_t lin_c2(const std::vector<Term> &terms) const;
    int64_t term_var(Bid b) const { return nodes_[b].kind == BlockNode::MAT ? nodes_[b].var : 1; }   // of a flattened term
    int set_var(Bid b, uint64_t v, bool check_only = false);                         // MAT blocks only (fhs_char_set_noise)

    int flush();
    // ---- level-skewed batching of independent jobs (fhs_submit / fhs_pump) ----------------------------------------
    // submit() plans the pending PBS as a JOB and schedules its dependency levels on consecutive TICKS starting at the
    // next one (later if an input is produced by an earlier job that has not run yet); pump(n) enqueues the next n ticks,
    // each as ONE launch group (lincomb -> keyswitch -> blind rotation) over the union of every job's level scheduled for
    // it.  With one submit + one pump per request the narrow tail levels of request k ride in the wide launch of
    // request k + 1 instead of paying one bootstrap latency each on an almost empty GPU.  flush() drains all ticks.
    // Automatic partial flush: once this many pending bootstraps have all their inputs available (depth 1), THAT level is
    // planned and enqueued while the caller keeps building the rest of the DAG; deeper nodes stay pending (their depth
    // drops by one), so the narrow later levels still merge across the whole operation.  A 1024-character replace records
    // 256 k bootstraps in ~0.25 s of host time: without this the GPU idles through all of it, and with 8 GPUs that is as
    // long as the computation.  0 = off.  Not used while jobs are scheduled by hand (submit / pump) or inputs are captured.
    size_t auto_flush_pending = 8192;
    // fhs_submit scheduling: a job level that would leave the launch group of its tick with a partly filled last round
    // of the persistent blind-rotation kernel (width not a multiple of the resident workgroup slots) sends the excess
    // rows to the next tick instead (their consumers move with them).  0 = off.
    size_t balance_slots = 0;
    int submit() { manual_jobs_ = true; return plan_job(false); }
    int pump(size_t n_ticks);
    bool has_scheduled() const { return !sched_.empty(); }
    // Level-parallel execution INSIDE the library (fhs_dist_level_parallel): every rank holds the same ciphertexts and
    // records the same DAG; flush() then runs slice [rank*cap, (rank+1)*cap) of every level, all-gathers the slices on
    // the context's stream (ctx.dist: RCCL, no host wait between levels) and installs the gathered level.
    bool level_parallel = false;
    // Rotation sharing (fused mode): rows of one level that differ only in the trivial constant of their input share one
    // keyswitch + blind rotation (plan_job).  fhs_set_rotation_sharing(ctx, 0) switches it off (A/B measurements, tests).
    bool share_rotations = true;
    // Gathers `n` blocks per rank: local[k] of every rank -> out[r * n + k] (fresh MAT blocks owned by the caller).
    // Flushes the local DAG first; everything is enqueued on the context's stream.
    int gather_blocks(const Bid *local, size_t n, std::vector<Bid> &out);
    // distributed execution (one process per GPU, identical DAGs on every rank): plan once, then per
    // level every rank runs its slice into a dense buffer, the caller all-gathers, commit scatters
    int dist_rank = 0, dist_world = 1;
    struct LevelPlan { size_t first, count; };
    int plan_flush();
    size_t planned_levels() const { return plan_.levels.size(); }
    size_t level_width(size_t k) const { return plan_.levels[k].count; }
    size_t planned_max_width() const { return plan_.max_width; }
    int exec_level(size_t k, size_t lo, size_t hi, uint64_t *dense_out);
    int commit_level(size_t k, const uint64_t *d_all);
    int read_block(Bid b, uint64_t *host_out);        // flushes if needed
    // `count` blocks into consecutive 2049-word rows on the host: one gather launch and one copy through the pinned staging
    // buffer per 2048 blocks instead of one synchronous copy per block (the twin of from_host_many)
    int read_many(const Bid *b, size_t count, uint64_t *host_out);
    // flushes if needed (do_flush = false: the caller has made sure the block's tick is enqueued); wait=false: enqueue only
    int copy_block_to_device(Bid b, uint64_t *d_out, bool wait = true, bool do_flush = true);
    uint64_t blocks_live() const { return live_dev_blocks_; }

    // ---- debug: capture of PBS inputs (the linear-combination results entering keyswitch) ----
    // Noise-margin tests download a sample of every level's inputs and measure their phase error with the client
    // key; nothing here sees a secret.  capture_max_rows == 0: off.
    struct CaptureRec { uint32_t level, index, lut, n_terms; int64_t sum_c2; int32_t konst; uint32_t width; };
    size_t capture_max_rows = 0;
    // capture_live: sample inside the ordinary execution path (plan_job / run_tick: rotation sharing, round alignment and
    // tick scheduling as in production) instead of the all-at-once plan of plan_flush (which never shares rotations)
    bool capture_live = false;
    std::vector<uint64_t> capture_rows;      // [n][2049]
    std::vector<CaptureRec> capture_recs;

    // ---- debug: plan trace (planner contexts; tests and bench.py's CPU-baseline leg) ----
    // While on, a planner context writes what it WOULD run as a flat list of 64-bit words (block tokens are the planner's
    // unique fake pointers): uploads in order, every launch group row by row (output token, LUT id, constant, terms), the
    // shared extractions, a group end.  A host executor replays it with any PBS implementation -- the CPU oracle runs the
    // product's fused DAGs that way (the checker's plan_exec.py); nothing in the product reads it back.
    enum : uint64_t { TR_UPLOAD = 1, TR_ROW = 2, TR_EXT = 3, TR_GROUP_END = 4 };
    bool trace_plan = false;
    std::vector<uint64_t> trace_;
    // [kind 0 TRIV / 1 MAT / 2 LIN / 3 pending, value or constant, n_terms, (token, coef)...] of one block
    void describe_block(Bid b, std::vector<uint64_t> &out) const;

    // ---- char handles (fhs_char_t) ----
    struct CharRec { Bid b[4]; bool used; };
    uint64_t new_char(const Bid b[4]);   // takes over the 4 references
    bool valid_char(uint64_t h) const { return h >= 1 && h <= chars_.size() && chars_[h - 1].used; }
    const Bid *char_blocks(uint64_t h) const { return chars_[h - 1].b; }
    void free_char(uint64_t h);

  private:
    std::vector<BlockNode> nodes_{1};   // slot 0 reserved
    std::vector<Bid> free_nodes_;
    struct Pend { Bid id; uint32_t gen; };                    // gen: a released pending node's slot may be reused before
    std::vector<Pend> pending_;                               // the flush; the stale entry must not stand for the new node
    std::vector<CharRec> chars_;
    std::vector<uint64_t> free_chars_;

    // device block pool
    std::vector<void *> chunks_;
    uint64_t *upload_pin_ = nullptr, *upload_dev_ = nullptr;   // staging of from_host_many: pinned host side, device side
    size_t upload_words_ = 0;
    hipEvent_t upload_done_ = nullptr;           // the last copy out of upload_pin_
    std::vector<uint64_t *> free_blocks_;
    uint64_t live_dev_blocks_ = 0;
    uint64_t planner_tokens_ = 0;
    bool ensure_staging(size_t n_blocks);
    uint64_t *alloc_block();
    void free_block(uint64_t *p);

    // LUT table on device (catalogue)
    uint64_t *d_luts_ = nullptr;
    DevBuf plan_buf_, batch_in_;

    struct FlushPlan {
        std::vector<LevelPlan> levels;
        std::vector<CaptureRec> recs;        // per planned PBS (only filled while capturing)
        size_t off_desc = 0, off_terms = 0, off_lut = 0, off_out = 0, max_width = 0;
    } plan_;

    // rotation sharing (plan_job): a follower row is a further sample extraction of its leader's blind rotation
    struct ShareRow {
        uint32_t lead_row;                // position of the leader among the rotation rows of the same TickLevel
        uint32_t K;                       // negacyclic coefficient to extract: 128 x (constant difference mod 32)
        uint64_t *out;                    // the follower's own block
    };
    struct TickLevel {
        std::vector<LinDesc> descs;       // first_term relative to `terms`
        std::vector<LinTerm> terms;
        std::vector<uint32_t> lut;
        std::vector<uint64_t *> out;
        std::vector<uint64_t *> body;     // empty, or per rotation row: where the accumulator's body polynomial goes (leaders)
        std::vector<ShareRow> ext;        // followers of this level's leaders
        std::vector<CaptureRec> recs;     // per row, only while capturing live (capture_live)
        uint64_t job = 0;                 // rows of one job that land on the same tick share one TickLevel
    };
    std::map<uint64_t, std::vector<TickLevel>> sched_;            // tick -> job levels to run in that launch group
    std::map<uint64_t, std::vector<uint64_t *>> free_after_;      // blocks reusable once that tick has been enqueued
    uint64_t next_tick_ = 1, last_sched_tick_ = 0;
    bool manual_jobs_ = false;           // the caller schedules jobs itself (fhs_submit): no automatic partial flushes
    bool in_auto_flush_ = false;
    int auto_flush_rc_ = 0;              // first error of an automatic partial flush, reported by the next flush()
    std::string auto_flush_err_;
    DevBuf tick_buf_;
    // sharded: level-parallel mode -- this rank runs slice [rank*cap, (rank+1)*cap) of the group into the exchange buffer,
    // the slices are all-gathered on the stream and scattered into the nodes' blocks
    int run_tick(std::vector<TickLevel> &levels, bool sharded = false);
    // plans the pending PBS level by level; run_now: every level is enqueued as soon as it is planned (the host plans level
    // k + 1 while the GPU runs level k), otherwise the levels are scheduled on ticks (submit)
    // stream_pump (scheduled path only): enqueue every tick as soon as no later level can add rows to it
    int plan_job(bool run_now, bool first_level_only = false, bool stream_pump = false);
    uint64_t job_counter_ = 0;
    uint32_t rot_counter_ = 0;           // rotation groups handed out (BlockNode::rot)
    size_t n_depth1_ = 0;                // pending bootstraps whose inputs are all available
    // ... and how many blind ROTATIONS they are once rows that differ only in a trivial constant share one: what the
    // automatic partial flush counts in (a round of the persistent kernel is 1024 rotations, not 1024 results)
    std::unordered_map<uint64_t, uint32_t> depth1_keys_;
    size_t n_depth1_solo_ = 0;           // depth-1 nodes without a share key
    size_t depth1_rotations() const { return n_depth1_solo_ + depth1_keys_.size(); }
    void depth1_add(const BlockNode &n) { if (n.nk) depth1_keys_[n.nk]++; else n_depth1_solo_++; }
    size_t peel_limit_ = 0;              // automatic partial flush of an idle GPU: take this many ready rows (0 = all)
    hipEvent_t last_group_done_ = nullptr;   // recorded behind every launch group: tells whether the GPU has run dry
    uint32_t idle_poll_ = 0;
    // Pinned staging for plan uploads: a hipMemcpyAsync from PAGEABLE memory blocks the host until the stream reaches
    // the copy, i.e. until the previous launch group has finished -- the host could never plan ahead of the GPU.  A small
    // ring of pinned buffers, each guarded by an event recorded after its copy, keeps the upload asynchronous.
    struct Staging { void *p = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool busy = false; };
    std::vector<Staging> staging_;
    int upload_plan(void *d_dst, const void *src, size_t bytes);

    // Common-subexpression table of the fused string layer: (LUT, constant, [(block, generation, coefficient)...]) ->
    // the bootstrap node that already computes it.  Nodes are immutable, so an entry stays valid while its node lives.
    // Keys are 128-bit order-independent hashes of the terms (two independent 64-bit mixes: a false match needs both to
    // collide, ~2^-128 per pair), so a lookup allocates nothing.
    struct CseEntry { uint64_t h2; Bid id; uint32_t gen; };
    std::unordered_map<uint64_t, CseEntry> cse_;

    Bid new_node();
    int materialize_lin(Bid b);
};

// RAII reference to a block
class Ref {
  public:
    Ref() = default;
    Ref(Engine *e, Bid id) : e_(e), id_(id) {}   // adopts one reference
    Ref(const Ref &o) : e_(o.e_), id_(o.id_) { if (id_) e_->retain(id_); }
    Ref(Ref &&o) noexcept : e_(o.e_), id_(o.id_) { o.id_ = 0; }
    Ref &operator=(Ref o) noexcept { std::swap(e_, o.e_); std::swap(id_, o.id_); return *this; }
    ~Ref() { if (id_) e_->release(id_); }
    Bid id() const { return id_; }
    Engine *engine() const { return e_; }
    Bid detach() { Bid t = id_; id_ = 0; return t; }   // caller takes the reference
    explicit operator bool() const { return id_ != 0; }

  private:
    Engine *e_ = nullptr;
    Bid id_ = 0;
};

}  // namespace fhs
