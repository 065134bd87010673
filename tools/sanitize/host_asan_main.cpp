// ASan/UBSan driver of the product's HOST-ONLY code (no device is touched): client keygen / encrypt / decrypt /
// key files (client.cpp), the twiddle and key transforms (ntt_tables.cpp, fft_tables.cpp), and the whole DAG layer --
// engine.cpp graph logic, radix.cpp, strings.cpp, capi_*.cpp -- through a planner context (fhs_ctx_create_planner),
// which records and levelises every string op of the C ABI without executing anything.
// Built by `make -C fhestring_amd/csrc asan`; run by tests/test_sanitizers.py.  CPU only.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fhestring_hip.h"
#include "../../fhestring_amd/csrc/fft_tables.h"
#include "../../fhestring_amd/csrc/ntt_tables.h"

static int fails = 0;
#define CHECK(x) do { if (!(x)) { std::printf("CHECK failed: %s (line %d)\n", #x, __LINE__); fails++; } } while (0)

static std::vector<fhs_char_t> dummy(fhs_ctx *c, size_t n) {
    std::vector<uint64_t> z((size_t)FHS_CHAR_WORDS, 0);
    std::vector<fhs_char_t> v;
    for (size_t i = 0; i < n; i++) v.push_back(fhs_upload(c, z.data()));
    return v;
}

int main() {
    // ---- client -----------------------------------------------------------------------------------------------
    fhs_client *ck = nullptr;
    CHECK(fhs_client_create_insecure_seeded(42, &ck) == FHS_OK);
    std::vector<uint64_t> ct((size_t)12 * FHS_CHAR_WORDS);
    CHECK(fhs_client_encrypt_str(ck, "sanitizers", 10, 2, ct.data()) == FHS_OK);
    char buf[16];
    size_t n = 0;
    CHECK(fhs_client_decrypt_str(ck, ct.data(), 12, buf, &n) == FHS_OK && n == 10 && !std::memcmp(buf, "sanitizers", 10));
    CHECK(fhs_client_encrypt_str(ck, "bad\0x", 5, 0, ct.data()) == FHS_ERR_ARG);
    const std::string path = "/tmp/fhs_asan_key.bin";
    CHECK(fhs_client_save(ck, path.c_str(), 0) == FHS_OK);
    fhs_client *ck2 = nullptr;
    CHECK(fhs_client_load(path.c_str(), &ck2) == FHS_OK);
    uint8_t v = 0;
    CHECK(fhs_client_decrypt_char(ck2, ct.data(), &v) == FHS_OK);
    fhs_client *os = nullptr;
    CHECK(fhs_client_create(&os) == FHS_OK);           // OS-entropy path
    fhs_client_destroy(os);
    std::remove(path.c_str());

    // ---- host transforms of the key ---------------------------------------------------------------------------
    {
        fhs::HostNttTables ht;
        fhs::build_ntt_tables(ht);
        fhs::HostFftTables ft;
        fhs::build_fft_tables(ft);
        CHECK(ht.fwd_uni.size() == 64 && ft.w_re.size() == 1024 && ft.mono.size() == 2 * 4096 && ft.r16.size() == 32);
        const uint64_t *mb = fhs_client_bsk_mb2(ck);              // pair key of FHS_ARITH_F64_FFT_MB2 (generated here)
        CHECK(mb != nullptr && (mb[0] & 63) == 0 && fhs_client_bsk_mb2(ck) == mb);
        std::vector<double> out((size_t)4 * 2 * 2 * 2048);       // one GGSW
        std::vector<uint64_t> one(fhs_client_bsk(ck), fhs_client_bsk(ck) + 4 * 2048);
        // convert_bsk_to_ntt walks all 742 GGSWs: give it the real key (reads only)
        std::vector<double> all((size_t)742 * 4 * 2 * 2048);
        fhs::convert_bsk_to_ntt(fhs_client_bsk(ck), all.data(), 4);
    }

    // ---- DAG layer through the planner ------------------------------------------------------------------------
    fhs_ctx *c = nullptr;
    CHECK(fhs_ctx_create_planner(&c) == FHS_OK);
    // strings whose characters are mostly PLAINTEXT (trivial) with a few ciphertexts in between: sums fold, trees see
    // constants, noise-driven groupings see zero-variance terms -- the shapes on which fixed-size term buffers overflowed
    auto mixed = [&](size_t n, size_t every) {
        std::vector<uint64_t> z((size_t)FHS_CHAR_WORDS, 0);
        std::vector<fhs_char_t> v;
        for (size_t i = 0; i < n; i++) v.push_back(i % every == every / 2 ? fhs_upload(c, z.data()) : fhs_trivial(c, (uint8_t)("ab c"[i % 4])));
        return v;
    };
    for (int variant = 0; variant < 5; variant++)
    for (int mode = 0; mode < 2; mode++) {
        if (variant >= 3 && mode == 0) continue;                 // long strings: the re-associated DAGs only (as written is O(n^2) nodes)
        CHECK(fhs_set_mode(c, mode) == FHS_OK);
        auto s = variant == 0 ? dummy(c, 14) : variant == 1 ? mixed(14, 5) : variant == 2 ? mixed(40, 9) : variant == 3 ? mixed(200, 50) : mixed(254, 300);
        auto p = variant == 0 ? dummy(c, 3) : mixed(3, variant == 1 ? 2 : 7), to = variant == 0 ? dummy(c, 5) : mixed(5, 3);
        auto o = variant == 0 ? dummy(c, 14) : mixed(s.size() - (variant & 1), 11);
        fhs_char_t r = 0, f = 0;
        CHECK(fhs_str_contains(c, s.data(), s.size(), p.data(), p.size(), &r) == FHS_OK);
        CHECK(fhs_str_contains_clear(c, s.data(), s.size(), "abc", 3, &r) == FHS_OK);
        CHECK(fhs_str_starts_with(c, s.data(), s.size(), p.data(), p.size(), &r) == FHS_OK);
        CHECK(fhs_str_ends_with(c, s.data(), s.size(), p.data(), p.size(), &r) == FHS_OK);
        CHECK(fhs_str_find(c, s.data(), s.size(), p.data(), p.size(), &r) == FHS_OK);
        CHECK(fhs_str_rfind(c, s.data(), s.size(), p.data(), p.size(), &r) == FHS_OK);
        CHECK(fhs_str_is_empty(c, s.data(), s.size(), &r) == FHS_OK);
        CHECK(fhs_str_len(c, s.data(), s.size(), &r) == FHS_OK);
        CHECK(fhs_str_eq(c, s.data(), s.size(), o.data(), o.size(), &r) == FHS_OK);
        CHECK(fhs_str_ne(c, s.data(), s.size(), o.data(), o.size(), &r) == FHS_OK);
        CHECK(fhs_str_eq_ignore_case(c, s.data(), s.size(), o.data(), o.size(), &r) == FHS_OK);
        for (int cmp = 0; cmp < 4; cmp++) CHECK(fhs_str_compare(c, s.data(), s.size(), o.data(), o.size(), cmp, &r) == FHS_OK);
        std::vector<fhs_char_t> out(64 * 16 + 2 * s.size() + o.size(), 0);
        CHECK(fhs_str_to_upper(c, s.data(), s.size(), out.data()) == FHS_OK);
        CHECK(fhs_str_to_lower(c, s.data(), s.size(), out.data()) == FHS_OK);
        CHECK(fhs_str_trim(c, s.data(), s.size(), out.data()) == FHS_OK);
        CHECK(fhs_str_trim_start(c, s.data(), s.size(), out.data()) == FHS_OK);
        CHECK(fhs_str_trim_end(c, s.data(), s.size(), out.data()) == FHS_OK);
        CHECK(fhs_str_strip_prefix(c, s.data(), s.size(), p.data(), p.size(), out.data(), &f) == FHS_OK);
        CHECK(fhs_str_strip_suffix(c, s.data(), s.size(), p.data(), p.size(), out.data(), &f) == FHS_OK);
        size_t len = 0;
        std::vector<fhs_char_t> rep(fhs_str_replace_len(s.size(), p.size(), to.size()) + 8);
        CHECK(fhs_str_replace(c, s.data(), s.size(), p.data(), p.size(), to.data(), to.size(), rep.data(), rep.size(), &len) == FHS_OK);
        CHECK(fhs_str_replace(c, s.data(), s.size(), to.data(), to.size(), p.data(), p.size(), rep.data(), rep.size(), &len) == FHS_OK);
        CHECK(fhs_str_concatenate(c, s.data(), s.size(), o.data(), o.size(), out.data()) == FHS_OK);
        if (mode == 1 && variant < 3) {                           // (the split family allocates n x n handles)
            for (int kind = 0; kind < 9; kind++) {
                const size_t d = fhs_str_split_dim(kind, s.size());
                std::vector<fhs_char_t> sp(d * d);
                size_t dim = 0;
                fhs_char_t cnt = (kind == 3 || kind == 6) ? fhs_trivial(c, 2) : 0;
                CHECK(fhs_str_split(c, kind, s.data(), s.size(), p.data(), kind == 8 ? 0 : p.size(), cnt, sp.data(), sp.size(), &dim, &f) == FHS_OK);
            }
        }
        CHECK(fhs_flush(c) == FHS_OK);
        uint64_t blocks[FHS_CHAR_WORDS];
        CHECK(fhs_download(c, r, blocks) == FHS_ERR_STATE);      // a planner computes nothing
    }
    {   // level-skewed batching: jobs on ticks, dependent jobs, handles released while their levels are still scheduled
        CHECK(fhs_set_mode(c, 1) == FHS_OK);
        CHECK(fhs_resident_slots(c) == 512 && fhs_set_tick_balance(c, 64) == FHS_OK);   // levels get split across ticks
        CHECK(fhs_set_launch_chunk(c, 0, 100) == FHS_OK && fhs_set_launch_chunk(c, 7, 1) == FHS_ERR_ARG);
        std::vector<fhs_char_t> keep;
        for (int k = 0; k < 6; k++) {
            auto s = dummy(c, 30);
            fhs_char_t r = 0, f = 0;
            CHECK(fhs_str_contains_clear(c, s.data(), s.size(), "abc", 3, &r) == FHS_OK);
            CHECK(fhs_str_find_clear(c, s.data(), s.size(), "abc", 3, &f) == FHS_OK);
            CHECK(fhs_submit(c) == FHS_OK);
            for (fhs_char_t h : s) CHECK(fhs_release(c, h) == FHS_OK);      // inputs dropped before their ticks run
            if (k & 1) CHECK(fhs_release(c, f) == FHS_OK); else keep.push_back(f);
            if (!keep.empty() && k == 3) {                                   // a job consuming an unfinished job
                fhs_char_t e = fhs_eq(c, keep[0], r);
                CHECK(e != 0 && fhs_submit(c) == FHS_OK);
                keep.push_back(e);
            }
            keep.push_back(r);
            CHECK(fhs_pump(c, 1) == FHS_OK);
        }
        CHECK(fhs_flush(c) == FHS_OK);
        size_t nl = 0;
        CHECK(fhs_level_widths(c, nullptr, 0, &nl) == FHS_OK && nl > 20);
        for (fhs_char_t h : keep) CHECK(fhs_release(c, h) == FHS_OK);
        CHECK(fhs_set_tick_balance(c, 0) == FHS_OK);
    }
    {   // all-trivial inputs: every bootstrap folds at recording time (the CPU truth-table tests live on this path)
        CHECK(fhs_set_mode(c, 1) == FHS_OK);
        std::vector<fhs_char_t> s, p;
        for (int i = 0; i < 257; i++) s.push_back(fhs_trivial(c, i == 256 ? 0 : (uint8_t)('a' + i % 3)));
        for (char ch : std::string("wxyz")) p.push_back(fhs_trivial(c, (uint8_t)ch));
        fhs_char_t r = 0;
        int triv = 0;
        uint8_t val = 0;
        CHECK(fhs_str_find(c, s.data(), s.size(), p.data(), p.size(), &r) == FHS_OK);
        CHECK(fhs_trivial_value(c, r, &triv, &val) == FHS_OK && triv == 1 && val == 255);
        CHECK(fhs_str_rfind(c, s.data(), s.size(), p.data(), p.size(), &r) == FHS_OK);
        CHECK(fhs_trivial_value(c, r, &triv, &val) == FHS_OK && triv == 1 && val == 255);
        std::vector<fhs_char_t> out(s.size());
        CHECK(fhs_bubble_zeroes_right(c, s.data(), s.size(), out.data()) == FHS_OK);
        CHECK(fhs_str_compare(c, s.data(), s.size(), s.data(), s.size() - 3, 1, &r) == FHS_OK);   // le: the longer buffer has a non-NUL tail
        CHECK(fhs_trivial_value(c, r, &triv, &val) == FHS_OK && triv == 1 && val == 0);
        CHECK(fhs_flush(c) == FHS_OK);
    }
    fhs_stats st;
    // (mostly plaintext strings: counts over repeated flags may pass the budget slightly, tests/test_planner.py)
    CHECK(fhs_get_stats(c, &st) == FHS_OK && st.pbs_executed > 1000 && st.max_input_sum_c2 <= 160);
    size_t w0, w1, c0, c1;
    fhs_dist_plan_windows(257, 4, 8, 7, &w0, &w1, &c0, &c1);
    CHECK(w1 == 254 && c1 == 257);
    CHECK(fhs_str_find(c, nullptr, 3, nullptr, 0, nullptr) != FHS_OK);   // argument errors do not crash
    fhs_ctx_destroy(c);
    fhs_client_destroy(ck);
    fhs_client_destroy(ck2);
    std::printf(fails ? "FAILED\n" : "host sanitizer run ok\n");
    return fails ? 1 : 0;
}
