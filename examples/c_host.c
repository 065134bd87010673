/* A compiled host over the C ABI, in plain C99 -- what a Rust `extern "C"` block would bind,
 * exercised the way the reference's main.rs does it (src/main.rs:34-116): client key, encrypt a string with padding,
 * run MyServerKey methods, decrypt, compare with the clear result, print `Test Passed: OK, Result: ...` like
 * src/utils.rs:114-120.  No Python, no C++ on this side of the boundary.
 *
 *     cc -std=c99 -Iinclude examples/c_host.c -Lfhestring_amd -lfhestring_hip -Wl,-rpath,$PWD/fhestring_amd -o examples/c_host
 *     examples/c_host "the quick brown fox" "brown"                (needs an MI355X: there is no CPU fallback)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fhestring_hip.h"

#define CHAR_WORDS (4 * 2049)                /* FheAsciiChar: 4 radix blocks of 2048 + 1 u64 (fheasciichar.rs:8-10) */

static fhs_ctx *ctx;
static fhs_client *ck;

static void die(const char *what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, ctx ? fhs_last_error(ctx) : "no context");
    exit(2);
}
#define TRY(call) do { int rc_ = (call); if (rc_ != FHS_OK) die(#call, rc_); } while (0)

/* MyClientKey::encrypt (client_key.rs:45-65) + upload: n = len + padding handles */
static fhs_char_t *encrypt_upload(const char *s, size_t padding, size_t *n_out) {
    const size_t len = strlen(s), n = len + padding;
    uint64_t *raw = malloc(n * CHAR_WORDS * sizeof(uint64_t));
    fhs_char_t *h = malloc((n ? n : 1) * sizeof(fhs_char_t));
    if (!raw || !h) die("malloc", -1);
    TRY(fhs_client_encrypt_str(ck, s, len, padding, raw));
    TRY(fhs_upload_string(ctx, raw, n, h));
    free(raw);
    *n_out = n;
    return h;
}

static uint8_t decrypt_char(fhs_char_t h) {
    uint64_t blocks[CHAR_WORDS];
    uint8_t v = 0;
    TRY(fhs_download(ctx, h, blocks));                        /* flushes the lazy DAG */
    TRY(fhs_client_decrypt_char(ck, blocks, &v));
    return v;
}

static char *decrypt_str(const fhs_char_t *h, size_t n) {
    uint64_t *raw = malloc((n ? n : 1) * CHAR_WORDS * sizeof(uint64_t));
    char *out = calloc(n + 1, 1);
    size_t len = 0;
    if (!raw || !out) die("malloc", -1);
    TRY(fhs_download_string(ctx, h, n, raw));                 /* one gather + one copy for the whole string */
    TRY(fhs_client_decrypt_str(ck, raw, n, out, &len));
    out[len] = 0;
    free(raw);
    return out;
}

static int report_u8(const char *method, unsigned got, unsigned want) {
    printf("%s: %s, Result: %u, Expected: %u\n", method, got == want ? "Test Passed: OK" : "Test FAILED", got, want);
    return got == want;
}
static int report_str(const char *method, const char *got, const char *want) {
    const int ok = strcmp(got, want) == 0;
    printf("%s: %s, Result: \"%s\", Expected: \"%s\"\n", method, ok ? "Test Passed: OK" : "Test FAILED", got, want);
    return ok;
}

int main(int argc, char **argv) {
    const char *text = argc > 1 ? argv[1] : "the quick brown fox jumps over the lazy dog";
    const char *pat = argc > 2 ? argv[2] : "brown";
    const size_t n_text = strlen(text), m = strlen(pat);
    int ok = 1;

    TRY(fhs_client_create_insecure_seeded(0xF5E57121ull, &ck));   /* reproducible demo keys; fhs_client_create for real ones */
    TRY(fhs_ctx_create(0, &ctx));
    TRY(fhs_set_arithmetic(ctx, FHS_ARITH_F64_FFT));               /* before or after the key load: either order works */
    TRY(fhs_load_server_key(ctx, fhs_client_bsk(ck), fhs_client_ksk(ck)));
    TRY(fhs_set_mode(ctx, FHS_MODE_FUSED));

    size_t n = 0, np = 0;
    fhs_char_t *s = encrypt_upload(text, 1, &n);                   /* STRING_PADDING = 1 (main.rs:12) */
    fhs_char_t *p = encrypt_upload(pat, 0, &np);                   /* patterns carry no padding (client_key.rs:67-79) */
    fhs_char_t r = 0;

    /* contains_clear / contains (mod.rs:198, :151) */
    TRY(fhs_str_contains_clear(ctx, s, n, pat, m, &r));
    ok &= report_u8("contains_clear", decrypt_char(r), strstr(text, pat) != NULL);
    TRY(fhs_release(ctx, r));
    TRY(fhs_str_contains(ctx, s, n, p, np, &r));
    ok &= report_u8("contains", decrypt_char(r), strstr(text, pat) != NULL);
    TRY(fhs_release(ctx, r));

    /* find (mod.rs:1010): index of the first match, 255 if none */
    const char *at = strstr(text, pat);
    TRY(fhs_str_find(ctx, s, n, p, np, &r));
    ok &= report_u8("find", decrypt_char(r), at ? (unsigned)(at - text) : 255u);
    TRY(fhs_release(ctx, r));

    /* len (mod.rs:478) */
    TRY(fhs_str_len(ctx, s, n, &r));
    ok &= report_u8("len", decrypt_char(r), (unsigned)(n_text & 255));
    TRY(fhs_release(ctx, r));

    /* to_upper (mod.rs:65) */
    {
        fhs_char_t *up = malloc(n * sizeof(fhs_char_t));
        char *want = malloc(n_text + 1);
        if (!up || !want) die("malloc", -1);
        for (size_t i = 0; i <= n_text; i++) want[i] = (char)((text[i] >= 'a' && text[i] <= 'z') ? text[i] - 32 : text[i]);
        TRY(fhs_str_to_upper(ctx, s, n, up));
        char *got = decrypt_str(up, n);
        ok &= report_str("to_upper", got, want);
        for (size_t i = 0; i < n; i++) TRY(fhs_release(ctx, up[i]));
        free(got); free(want); free(up);
    }

    /* replace with encrypted from / to (mod.rs:624): the pattern becomes "<>" */
    {
        size_t nt = 0, out_len = 0;
        fhs_char_t *to = encrypt_upload("<>", 0, &nt);
        const size_t cap = fhs_str_replace_len(n, np, nt);
        fhs_char_t *out = malloc((cap ? cap : 1) * sizeof(fhs_char_t));
        if (!out) die("malloc", -1);
        TRY(fhs_str_replace(ctx, s, n, p, np, to, nt, out, cap, &out_len));
        char *got = decrypt_str(out, out_len);
        /* the clear model.  |from| >= |to| (handle_longer_from, mod.rs:828-882): EVERY window of the ORIGINAL text that
         * matches is overwritten with `to` padded with NULs, later windows over earlier ones, then the NULs are bubbled out
         * -- "aaaa".replace("aa", "<>") is "<<<>" in the reference, not str::replace's "<><>".  |from| < |to|
         * (handle_shorter_from, :885-980): the greedy non-overlapping matches, left to right, like str::replace. */
        char *want = calloc(n_text * 2 + 3, 1);
        if (!want) die("malloc", -1);
        if (m >= 2) {
            char *work = malloc(n_text + 1);
            size_t w = 0;
            if (!work) die("malloc", -1);
            memcpy(work, text, n_text + 1);
            for (size_t i = 0; i + m <= n_text; i++)
                if (strncmp(text + i, pat, m) == 0)
                    for (size_t k = 0; k < m; k++) work[i + k] = k < 2 ? "<>"[k] : '\0';
            for (size_t i = 0; i < n_text; i++)
                if (work[i]) want[w++] = work[i];
            free(work);
        } else {
            for (const char *q = text; *q;) {
                if (m && strncmp(q, pat, m) == 0) { strcat(want, "<>"); q += m; }
                else { strncat(want, q, 1); q++; }
            }
        }
        ok &= report_str("replace", got, want);
        for (size_t i = 0; i < out_len; i++) TRY(fhs_release(ctx, out[i]));
        for (size_t i = 0; i < nt; i++) TRY(fhs_release(ctx, to[i]));
        free(got); free(want); free(out); free(to);
    }

    fhs_stats st;
    TRY(fhs_get_stats(ctx, &st));
    printf("PBS executed: %llu in %llu launch groups (largest bootstrap input: %llu output variances, budget %d)\n",
           (unsigned long long)st.pbs_executed, (unsigned long long)st.levels, (unsigned long long)st.max_input_sum_c2,
           FHS_NOISE_BUDGET_SUM_C2);
    ok &= st.max_input_sum_c2 <= FHS_NOISE_BUDGET_SUM_C2;

    for (size_t i = 0; i < n; i++) TRY(fhs_release(ctx, s[i]));
    for (size_t i = 0; i < np; i++) TRY(fhs_release(ctx, p[i]));
    free(s); free(p);
    fhs_ctx_destroy(ctx);
    fhs_client_destroy(ck);
    return ok ? 0 : 1;
}
