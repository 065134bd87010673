// extern "C" boundary, part 1: context, server key, raw batched PBS (include/fhestring_hip.h).
#include <algorithm>
#include <new>

#include "capi_internal.h"
#include "luts.h"

extern "C" {

int fhs_ctx_create(int device_id, fhs_ctx **out) {
    if (!out) return FHS_ERR_ARG;
    *out = nullptr;
    fhs_ctx *c = new (std::nothrow) fhs_ctx();
    if (!c) return FHS_ERR_STATE;
    int rc = c->eng.ctx.init(device_id);
    if (rc) {
        // keep the object alive so the caller can read the error text
        *out = c;
        return rc;
    }
    *out = c;
    return FHS_OK;
}

int fhs_ctx_create_planner(fhs_ctx **out) {
    if (!out) return FHS_ERR_ARG;
    fhs_ctx *c = new (std::nothrow) fhs_ctx();
    if (!c) return FHS_ERR_STATE;
    c->eng.planner = true;
    *out = c;
    return FHS_OK;
}

void fhs_ctx_destroy(fhs_ctx *ctx) {
    if (!ctx) return;
    ctx->eng.shutdown();
    delete ctx;
}

const char *fhs_last_error(const fhs_ctx *ctx) { return ctx ? ctx->eng.ctx.err.c_str() : "null context"; }

int fhs_load_server_key(fhs_ctx *ctx, const uint64_t *bsk, const uint64_t *ksk) {
    if (!ctx) return FHS_ERR_ARG;
    int rc = ctx->eng.ctx.load_server_key(bsk, ksk);
    if (rc) return rc;
    return ctx->eng.on_key_loaded();
}

int fhs_load_multibit_key(fhs_ctx *ctx, const uint64_t *bsk_mb2) {
    if (!ctx) return FHS_ERR_ARG;
    if (ctx->eng.planner) return FHS_OK;
    if (int rc = ctx->eng.flush()) return rc;
    return ctx->eng.ctx.load_multibit_key(bsk_mb2);
}

int fhs_set_arithmetic(fhs_ctx *ctx, int arith) {
    if (!ctx) return FHS_ERR_ARG;
    if (int rc = ctx->eng.flush()) return rc;   // pending work runs in the arithmetic it was built under
    return ctx->eng.ctx.set_arithmetic(arith);
}
int fhs_get_arithmetic(const fhs_ctx *ctx) { return ctx ? ctx->eng.ctx.arith : FHS_ERR_ARG; }
int fhs_set_launch_chunk(fhs_ctx *ctx, int arith, size_t n_ciphertexts) {
    if (!ctx || arith < 0 || arith > 3) return FHS_ERR_ARG;
    ctx->eng.ctx.launch_chunk[arith] = n_ciphertexts;
    return FHS_OK;
}
int fhs_set_fft4_max_batch(fhs_ctx *ctx, int max_batch) {
    if (!ctx || max_batch < 0) return FHS_ERR_ARG;
    if (int rc = ctx->eng.flush()) return rc;
    ctx->eng.ctx.fft4_max_batch = max_batch;
    return FHS_OK;
}
void fhs_fft_tables(double *w_re, double *w_im, double *u_re, double *u_im) {
    fhs::HostFftTables t;
    fhs::build_fft_tables(t);
    std::copy(t.w_re.begin(), t.w_re.end(), w_re);
    std::copy(t.w_im.begin(), t.w_im.end(), w_im);
    std::fill(u_re, u_re + 16, 0.0);
    std::fill(u_im, u_im + 16, 0.0);
    std::copy(t.u_re.begin(), t.u_re.end(), u_re);
    std::copy(t.u_im.begin(), t.u_im.end(), u_im);
}

void fhs_fft_mono_table(double *mono /*[4096][2]*/) {
    fhs::HostFftTables t;
    fhs::build_fft_tables(t);
    std::copy(t.mono.begin(), t.mono.end(), mono);
}

int fhs_read_server_key_file(const char *path, std::vector<uint64_t> &bsk, std::vector<uint64_t> &ksk);

int fhs_read_multibit_key_file(const char *path, std::vector<uint64_t> &mb);

int fhs_load_multibit_key_file(fhs_ctx *ctx, const char *path) {
    if (!ctx || !path) return FHS_ERR_ARG;
    std::vector<uint64_t> mb;
    if (fhs_read_multibit_key_file(path, mb) != FHS_OK)
        return ctx->eng.ctx.fail(FHS_ERR_STATE, "cannot read pair key file (missing, truncated or wrong parameters)");
    return fhs_load_multibit_key(ctx, mb.data());
}

int fhs_load_server_key_file(fhs_ctx *ctx, const char *path) {
    if (!ctx || !path) return FHS_ERR_ARG;
    std::vector<uint64_t> bsk, ksk;
    if (fhs_read_server_key_file(path, bsk, ksk) != FHS_OK)
        return ctx->eng.ctx.fail(FHS_ERR_STATE, "cannot read key file (missing, truncated or wrong parameters)");
    return fhs_load_server_key(ctx, bsk.data(), ksk.data());
}

int fhs_pbs_batch(fhs_ctx *ctx, const uint64_t *in, const uint32_t *lut_idx, const uint64_t *luts,
                  size_t n_luts, uint64_t *out, size_t B) {
    if (!ctx) return FHS_ERR_ARG;
    return ctx->eng.ctx.pbs_batch_host(in, lut_idx, luts, n_luts, out, B);
}

int fhs_pbs_batch_shifted(fhs_ctx *ctx, const uint64_t *in, const uint32_t *lut_idx, const uint64_t *luts, size_t n_luts,
                          const uint32_t *shifts, size_t n_shifts, uint64_t *out, size_t B) {
    if (!ctx) return FHS_ERR_ARG;
    if (ctx->eng.planner) return ctx->eng.ctx.fail(FHS_ERR_STATE, "planner context: nothing is computed");
    return ctx->eng.ctx.pbs_batch_shifted_host(in, lut_idx, luts, n_luts, shifts, n_shifts, out, B);
}

int fhs_debug_blind_rotate_batch(fhs_ctx *ctx, const uint64_t *ks, const uint32_t *lut_idx, const uint64_t *luts,
                                 size_t n_luts, uint64_t *out, size_t B) {
    if (!ctx) return FHS_ERR_ARG;
    if (ctx->eng.planner) return ctx->eng.ctx.fail(FHS_ERR_STATE, "planner context: nothing is computed");
    return ctx->eng.ctx.blind_rotate_host(ks, lut_idx, luts, n_luts, out, B);
}

int fhs_keyswitch_modswitch_batch(fhs_ctx *ctx, const uint64_t *in, uint32_t *ms_out, size_t B) {
    if (!ctx) return FHS_ERR_ARG;
    return ctx->eng.ctx.ks_ms_batch_host(in, ms_out, B);
}

int fhs_pbs_batch_device(fhs_ctx *ctx, const uint64_t *d_in, const uint32_t *d_lut_idx, const uint64_t *d_luts,
                         uint64_t *d_out, size_t B, void *hip_stream) {
    if (!ctx) return FHS_ERR_ARG;
    if (!d_in || !d_lut_idx || !d_luts || !d_out) return ctx->eng.ctx.fail(FHS_ERR_ARG, "null device pointer");
    return ctx->eng.ctx.pbs_batch_device(d_in, d_lut_idx, d_luts, d_out, B,
                                         hip_stream ? (hipStream_t)hip_stream : ctx->eng.ctx.stream);
}

int fhs_debug_capture_pbs_inputs(fhs_ctx *ctx, size_t max_rows_per_level) {
    if (!ctx) return FHS_ERR_ARG;
    if (int rc = ctx->eng.flush()) return rc;
    ctx->eng.capture_max_rows = max_rows_per_level;
    if (!max_rows_per_level) {
        ctx->eng.capture_rows.clear(); ctx->eng.capture_rows.shrink_to_fit();
        ctx->eng.capture_recs.clear();
    }
    return FHS_OK;
}
int fhs_debug_capture_live(fhs_ctx *ctx, int on) {
    if (!ctx) return FHS_ERR_ARG;
    if (int rc = ctx->eng.flush()) return rc;
    ctx->eng.capture_live = on != 0;
    return FHS_OK;
}

int fhs_debug_plan_trace(fhs_ctx *ctx, int on) {
    if (!ctx) return FHS_ERR_ARG;
    if (!ctx->eng.planner) return ctx->eng.ctx.fail(FHS_ERR_STATE, "fhs_debug_plan_trace: planner contexts only (fhs_ctx_create_planner)");
    if (int rc = ctx->eng.flush()) return rc;
    ctx->eng.trace_plan = on != 0;
    ctx->eng.trace_.clear();
    return FHS_OK;
}

int fhs_debug_plan_read(fhs_ctx *ctx, uint64_t *out, size_t cap, size_t *n) {
    if (!ctx || !n) return FHS_ERR_ARG;
    if (int rc = ctx->eng.flush()) return rc;
    const std::vector<uint64_t> &t = ctx->eng.trace_;
    *n = t.size();
    if (!out) return FHS_OK;
    if (cap < t.size()) return ctx->eng.ctx.fail(FHS_ERR_ARG, "fhs_debug_plan_read: buffer too small");
    std::copy(t.begin(), t.end(), out);
    ctx->eng.trace_.clear();
    return FHS_OK;
}

int fhs_debug_char_terms(fhs_ctx *ctx, fhs_char_t h, uint64_t *out, size_t cap, size_t *n) {
    if (!ctx || !n || !ctx->eng.valid_char(h)) return ctx ? ctx->eng.ctx.fail(FHS_ERR_ARG, "invalid character handle") : FHS_ERR_ARG;
    if (int rc = ctx->eng.flush()) return rc;
    std::vector<uint64_t> v;
    const fhs::Bid *b = ctx->eng.char_blocks(h);
    for (int k = 0; k < 4; k++) ctx->eng.describe_block(b[k], v);
    *n = v.size();
    if (!out) return FHS_OK;
    if (cap < v.size()) return ctx->eng.ctx.fail(FHS_ERR_ARG, "fhs_debug_char_terms: buffer too small");
    std::copy(v.begin(), v.end(), out);
    return FHS_OK;
}

int fhs_debug_lut_poly(int lut_id, uint64_t *out) {
    if (!out || lut_id < 0 || lut_id >= fhs::LUT_COUNT) return FHS_ERR_ARG;
    fhs::make_lut_poly(lut_id, out);
    return FHS_OK;
}

int fhs_debug_capture_read(fhs_ctx *ctx, uint64_t *rows, fhs_capture_rec *recs, size_t cap, size_t *n) {
    if (!ctx || !n) return FHS_ERR_ARG;
    auto &e = ctx->eng;
    static_assert(sizeof(fhs_capture_rec) == sizeof(fhs::Engine::CaptureRec), "record layout");
    const size_t have = e.capture_recs.size();
    if (!rows || !recs) { *n = have; return FHS_OK; }
    const size_t k = std::min(cap, have);
    std::copy(e.capture_rows.begin(), e.capture_rows.begin() + k * fhs::BIG_CT, rows);
    std::copy(e.capture_recs.begin(), e.capture_recs.begin() + k,
              reinterpret_cast<fhs::Engine::CaptureRec *>(recs));
    *n = k;
    e.capture_rows.clear();
    e.capture_recs.clear();
    return FHS_OK;
}

int fhs_kernel_timing(fhs_ctx *ctx, int reset, double *blind_rotate_ms, double *keyswitch_ms,
                      uint64_t *n_blind_rotate, uint64_t *n_keyswitch, uint64_t *pbs_in_launches) {
    if (!ctx) return FHS_ERR_ARG;
    auto &t = ctx->eng.ctx.timer;
    t.resolve();
    if (blind_rotate_ms) *blind_rotate_ms = t.n[0] ? t.ms[0] / (double)t.n[0] : 0.0;
    if (keyswitch_ms) *keyswitch_ms = t.n[1] ? t.ms[1] / (double)t.n[1] : 0.0;
    if (n_blind_rotate) *n_blind_rotate = t.n[0];
    if (n_keyswitch) *n_keyswitch = t.n[1];
    if (pbs_in_launches) *pbs_in_launches = t.units[0];
    if (reset) t.reset();
    return FHS_OK;
}

int fhs_kernel_timing_kind(fhs_ctx *ctx, int kind, double *avg_ms, uint64_t *launches, uint64_t *pbs_in_launches) {
    if (!ctx || kind < 0 || kind > 2) return FHS_ERR_ARG;
    auto &t = ctx->eng.ctx.timer;
    t.resolve();
    if (avg_ms) *avg_ms = t.n[kind] ? t.ms[kind] / (double)t.n[kind] : 0.0;
    if (launches) *launches = t.n[kind];
    if (pbs_in_launches) *pbs_in_launches = t.units[kind];
    return FHS_OK;
}

}  // extern "C"
