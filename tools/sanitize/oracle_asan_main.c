/* ASan/UBSan driver of the CPU oracle (oracle/tfhe_oracle.c): keygen, one PBS in every arithmetic, keyswitch, the
 * NTT-vs-schoolbook identity and a threaded batch.  Built by `make -C oracle asan`; run by tests/test_sanitizers.py.
 * CPU only (GPU sanitizers are not available on this pool). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef uint64_t u64;
typedef uint32_t u32;
typedef int64_t i64;
typedef struct orc_server_key orc_server_key;
void orc_keygen(u64 seed, u64 *lwe_sk, u64 *glwe_sk, u64 *bsk, u64 *ksk);
orc_server_key *orc_server_key_new(const u64 *bsk, const u64 *ksk);
void orc_server_key_free(orc_server_key *k);
void orc_encrypt_block(const u64 *glwe_sk, u64 m, u64 *rng_state, u64 *ct);
u64 orc_decrypt_block(const u64 *glwe_sk, const u64 *ct);
void orc_make_lut(const u64 *f_table, u64 *lut);
void orc_keyswitch_modswitch(const orc_server_key *k, const u64 *in, u32 *out);
void orc_pbs(const orc_server_key *k, const u64 *in, const u64 *lut, u64 *out, int mode);
void orc_pbs_batch(const orc_server_key *k, const u64 *in, const u32 *lut_idx, const u64 *luts, u64 *out, u64 B,
                   int nthreads, int mode);
void orc_negacyclic_schoolbook(const i64 *d, const u64 *b, u64 *res);
void orc_negacyclic_ntt(const i64 *d, const u64 *b_quantised, u64 *res);
void orc_bsk_quantize(u64 *bsk, u64 n);
void orc_keygen_mb2(u64 seed, const u64 *lwe_sk, const u64 *glwe_sk, u64 *bsk_mb2);
void orc_server_key_set_mb2(orc_server_key *k, const u64 *bsk_mb2);
u64 orc_bsk_words(void);
u64 orc_ksk_words(void);

int main(void) {
    const u64 nb = orc_bsk_words(), nk = orc_ksk_words();
    u64 *lwe = calloc(742, 8), *glwe = calloc(2048, 8), *bsk = calloc(nb, 8), *ksk = calloc(nk, 8);
    orc_keygen(12345, lwe, glwe, bsk, ksk);
    orc_server_key *K = orc_server_key_new(bsk, ksk);
    u64 rng = 99, tab[16], lut[2048], in[4 * 2049], out[4 * 2049];
    for (int x = 0; x < 16; x++) tab[x] = (u64)((x * 3 + 1) & 3);
    orc_make_lut(tab, lut);
    int bad = 0;
    for (int mode = 0; mode <= 3; mode += (mode == 0 ? 2 : 1)) {        /* 0 exact NTT, 2 f64 FFT, 3 GPU mirror */
        orc_encrypt_block(glwe, 7, &rng, in);
        orc_pbs(K, in, lut, out, mode);
        if (orc_decrypt_block(glwe, out) != tab[7]) { printf("mode %d wrong\n", mode); bad = 1; }
    }
    {   /* mode 4: two key bits per external product, with its pair key */
        u64 *mb = calloc((size_t)371 * 3 * 4 * 2048, 8);
        orc_keygen_mb2(12345, lwe, glwe, mb);
        orc_server_key_set_mb2(K, mb);
        orc_encrypt_block(glwe, 11, &rng, in);
        orc_pbs(K, in, lut, out, 4);
        if (orc_decrypt_block(glwe, out) != tab[11]) { printf("mode 4 wrong\n"); bad = 1; }
        free(mb);
    }
    u32 ms[743], idx[4] = {0, 0, 0, 0};
    orc_keyswitch_modswitch(K, in, ms);
    for (int b = 0; b < 4; b++) orc_encrypt_block(glwe, (u64)(b * 5 % 16), &rng, in + b * 2049);
    orc_pbs_batch(K, in, idx, lut, out, 4, 4, 3);
    for (int b = 0; b < 4; b++)
        if (orc_decrypt_block(glwe, out + b * 2049) != tab[b * 5 % 16]) { printf("batch %d wrong\n", b); bad = 1; }
    i64 *d = calloc(2048, 8);
    u64 *q = calloc(2048, 8), *r1 = calloc(2048, 8), *r2 = calloc(2048, 8);
    for (int i = 0; i < 2048; i++) { d[i] = (i64)((i * 2654435761u) % 8388608u) - 4194304; q[i] = 0x9E3779B97F4A7C15ull * (u64)(i + 1); }
    orc_bsk_quantize(q, 2048);
    orc_negacyclic_schoolbook(d, q, r1);
    orc_negacyclic_ntt(d, q, r2);
    if (memcmp(r1, r2, 2048 * 8)) { printf("ntt != schoolbook\n"); bad = 1; }
    orc_server_key_free(K);
    free(lwe); free(glwe); free(bsk); free(ksk); free(d); free(q); free(r1); free(r2);
    printf(bad ? "FAILED\n" : "oracle sanitizer run ok\n");
    return bad;
}
