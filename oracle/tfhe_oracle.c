/*
 * oracle/tfhe_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * CPU restatement of the TFHE programmable-bootstrap path that the reference
 * (MakisChristou/fhestring) reaches through `tfhe 0.5.2` (Cargo.lock:416-433)
 * from src/ciphertext/fheasciichar.rs:23-102 and src/client_key.rs:31-99 with
 * PARAM_MESSAGE_2_CARRY_2_KS_PBS (src/main.rs:3,43).  The arithmetic crate is
 * NOT on disk, so this file restates its published algorithm (SURVEY.md
 * Appendix A): LWE keyswitch -> modulus switch -> blind rotation (CMUX chain of
 * GGSW x GLWE external products over Z_{2^64}[X]/(X^2048+1)) -> sample extract.
 *
 * PARITY STATUS: ciphertext-level parity with tfhe-rs is UNPINNED (the
 * reference holds no ciphertext-level known-answer vectors, draws keys from OS
 * entropy and multiplies polynomials with an f64 FFT).  What IS pinned:
 *   - decrypt-level results against the reference's own test literals
 *     (tests/golden/ref_tests.json, from src/main.rs:138-1153);
 *   - the exact negacyclic product used here (Goldilocks NTT, 2 x 29-bit key
 *     limbs) is checked bit-for-bit against a schoolbook product mod 2^64.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (fhestring_amd/) never does.
 *
 * Everything here is exact integer arithmetic (wrapping u64 torus).
 */
#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;
typedef __uint128_t u128;

/* ---- PARAM_MESSAGE_2_CARRY_2_KS_PBS -------------- */
#define LWE_N 742       /* small LWE dimension                               */
#define POLY_N 2048     /* GLWE polynomial size, k = 1                        */
#define BIG_N 2048      /* big LWE dimension k*N                              */
#define PBS_BASE_LOG 23 /* 1 level                                            */
#define KS_BASE_LOG 3
#define KS_LEVEL 5
#define DELTA_LOG 59    /* 2 msg + 2 carry + 1 padding bit                    */
#define LWE_NOISE 7.069849454709433e-6
#define GLWE_NOISE 2.9403601535432533e-16
#define BSK_QUANT_BITS 6 /* BSK coefficients are multiples of 2^6 (58-bit torus) */

#define BSK_POLYS (LWE_N * 4)            /* [i][row][col] polys of POLY_N    */
#define BSK_WORDS ((size_t)BSK_POLYS * POLY_N)
#define KSK_WORDS ((size_t)BIG_N * KS_LEVEL * (LWE_N + 1))
#define BIG_CT (BIG_N + 1)
#define SMALL_CT (LWE_N + 1)

/* ---- deterministic RNG (SplitMix64 + Box-Muller) ------------------------ */
typedef struct { u64 s; } orc_rng;

static inline u64 rng_u64(orc_rng *r) {
    u64 z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double rng_unit(orc_rng *r) { /* (0,1] */
    return ((double)(rng_u64(r) >> 11) + 1.0) * (1.0 / 9007199254740992.0);
}
static inline u64 rng_noise(orc_rng *r, double std_frac) {
    double u1 = rng_unit(r), u2 = rng_unit(r);
    double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
    double v = g * std_frac * 18446744073709551616.0;
    return (u64)(i64)llround(v);
}

/* ---- Goldilocks field p = 2^64 - 2^32 + 1 ------------------------------- */
#define GP 0xFFFFFFFF00000001ull
#define GEPS 0xFFFFFFFFull

static inline u64 g_reduce128(u128 x) {
    u64 lo = (u64)x, hi = (u64)(x >> 64);
    u64 hh = hi >> 32, hl = hi & GEPS;
    u64 t0 = lo - hh;
    t0 -= GEPS & (0 - (u64)(lo < hh)); /* borrow: add p == subtract eps (mod 2^64) */
    u64 t1 = hl * GEPS;
    u64 res = t0 + t1;
    res += GEPS & (0 - (u64)(res < t1));
    u64 c = res - GP;
    return res >= GP ? c : res;
}
static inline u64 g_mul(u64 a, u64 b) { return g_reduce128((u128)a * b); }
static inline u64 g_add(u64 a, u64 b) { /* a,b < p */
    u64 s = a + b;
    u64 t = s - GP;
    return ((s < a) | (s >= GP)) ? t : s;
}
static inline u64 g_sub(u64 a, u64 b) { return a - b + (GP & (0 - (u64)(a < b))); }
static u64 g_pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = g_mul(r, b); b = g_mul(b, b); e >>= 1; }
    return r;
}
static u64 g_inv(u64 a) { return g_pow(a, GP - 2); }

/* negacyclic NTT tables (Cooley-Tukey forward with merged psi powers,
 * Gentleman-Sande inverse; bit-reversed twiddle order) */
static u64 g_psi_br[POLY_N], g_ipsi_br[POLY_N], g_ninv;
static int g_tables_ready = 0;
static pthread_mutex_t g_tab_mu = PTHREAD_MUTEX_INITIALIZER;

static unsigned bitrev11(unsigned x) {
    unsigned r = 0;
    for (int i = 0; i < 11; i++) r |= ((x >> i) & 1u) << (10 - i);
    return r;
}
static void g_init_tables(void) {
    pthread_mutex_lock(&g_tab_mu);
    if (!g_tables_ready) {
        u64 psi = g_pow(7, (GP - 1) / (2 * POLY_N)); /* primitive 4096-th root */
        u64 ipsi = g_inv(psi);
        u64 a = 1, b = 1;
        for (unsigned i = 0; i < POLY_N; i++) {
            g_psi_br[bitrev11(i)] = a;
            g_ipsi_br[bitrev11(i)] = b;
            a = g_mul(a, psi);
            b = g_mul(b, ipsi);
        }
        g_ninv = g_inv(POLY_N);
        g_tables_ready = 1;
    }
    pthread_mutex_unlock(&g_tab_mu);
}
static void g_ntt_fwd(u64 *a) {
    unsigned t = POLY_N;
    for (unsigned m = 1; m < POLY_N; m <<= 1) {
        t >>= 1;
        for (unsigned i = 0; i < m; i++) {
            u64 w = g_psi_br[m + i];
            u64 *x = a + 2 * i * t, *y = x + t;
            for (unsigned j = 0; j < t; j++) {
                u64 u = x[j], v = g_mul(y[j], w);
                x[j] = g_add(u, v);
                y[j] = g_sub(u, v);
            }
        }
    }
}
static void g_ntt_inv(u64 *a) {
    unsigned t = 1;
    for (unsigned m = POLY_N; m > 1; m >>= 1) {
        unsigned h = m >> 1;
        for (unsigned i = 0; i < h; i++) {
            u64 w = g_ipsi_br[h + i];
            u64 *x = a + 2 * i * t, *y = x + t;
            for (unsigned j = 0; j < t; j++) {
                u64 u = x[j], v = y[j];
                x[j] = g_add(u, v);
                y[j] = g_mul(g_sub(u, v), w);
            }
        }
        t <<= 1;
    }
    for (unsigned j = 0; j < POLY_N; j++) a[j] = g_mul(a[j], g_ninv);
}


/* ---- f64 FFT variant of the external product (mode 2) ---------------------
 * The algorithm CLASS the reference's tfhe/concrete-fft uses (Cargo.lock:168-179): a negacyclic
 * product through a 1024-point complex FFT of the folded polynomial z[n] = (x[n] + i x[n+N/2]) * zeta^n.
 * It is approximate (53-bit mantissa): used ONLY as the faster CPU baseline in bench.py and checked
 * at decrypt level against the exact path; it is never the parity oracle. */
#define FFT_N (POLY_N / 2)
static double fft_tw_re[FFT_N], fft_tw_im[FFT_N];     /* zeta^n, zeta = exp(i pi / N)         */
static double fft_w_re[FFT_N / 2], fft_w_im[FFT_N / 2]; /* exp(-2 pi i k / FFT_N)             */
static unsigned fft_rev[FFT_N];
static int fft_ready = 0;
static void fft_init(void) {
    pthread_mutex_lock(&g_tab_mu);
    if (!fft_ready) {
        const double PI = 3.14159265358979323846;
        for (int n = 0; n < FFT_N; n++) {
            fft_tw_re[n] = cos(PI * n / POLY_N);
            fft_tw_im[n] = sin(PI * n / POLY_N);
            unsigned r = 0;
            for (int b = 0; b < 10; b++) r |= ((n >> b) & 1u) << (9 - b);
            fft_rev[n] = r;
        }
        for (int k = 0; k < FFT_N / 2; k++) {
            fft_w_re[k] = cos(-2.0 * PI * k / FFT_N);
            fft_w_im[k] = sin(-2.0 * PI * k / FFT_N);
        }
        fft_ready = 1;
    }
    pthread_mutex_unlock(&g_tab_mu);
}
/* in-place radix-2 DIT on bit-reversed input; inverse = conjugated twiddles, unscaled */
static void fft_core(double *re, double *im, int inverse) {
    for (int len = 2; len <= FFT_N; len <<= 1) {
        const int half = len >> 1, step = FFT_N / len;
        for (int i = 0; i < FFT_N; i += len)
            for (int k = 0; k < half; k++) {
                const double wr = fft_w_re[k * step], wi = inverse ? -fft_w_im[k * step] : fft_w_im[k * step];
                const int a = i + k, b = a + half;
                const double tr = re[b] * wr - im[b] * wi, ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr; im[b] = im[a] - ti;
                re[a] += tr; im[a] += ti;
            }
    }
}
/* forward transform of a real polynomial given as doubles: out[k] (bit-reversed permuted input) */
static void fft_forward_poly(const double *x, double *re, double *im) {
    for (int n = 0; n < FFT_N; n++) {
        const double a = x[n], b = x[n + FFT_N];
        const unsigned r = fft_rev[n];
        re[r] = a * fft_tw_re[n] - b * fft_tw_im[n];
        im[r] = a * fft_tw_im[n] + b * fft_tw_re[n];
    }
    fft_core(re, im, 0);
}
/* inverse: spectrum -> polynomial coefficients (as doubles, exact value up to rounding) */
static void fft_inverse_poly(double *re, double *im, double *x) {
    static __thread double tr[FFT_N], ti[FFT_N];
    for (int k = 0; k < FFT_N; k++) { tr[fft_rev[k]] = re[k]; ti[fft_rev[k]] = im[k]; }
    fft_core(tr, ti, 1);
    const double sc = 1.0 / FFT_N;
    for (int n = 0; n < FFT_N; n++) {
        const double zr = tr[n] * sc, zi = ti[n] * sc;          /* undo the twist: multiply by conj(zeta^n) */
        x[n] = zr * fft_tw_re[n] + zi * fft_tw_im[n];
        x[n + FFT_N] = zi * fft_tw_re[n] - zr * fft_tw_im[n];
    }
}
static inline u64 f64_to_torus(double v) {                         /* v mod 2^64, round to nearest */
    const double two64 = 18446744073709551616.0;
    v -= two64 * nearbyint(v / two64);                              /* now |v| <= 2^63 */
    return (u64)(i64)llrint(v);
}

/* ---- mode 3: lane-for-lane mirror of the product's f64-FFT blind-rotation kernel ------------------
 * (fhestring_amd/csrc/fft_kernels.hip).  Folded negacyclic transform: z[n] = x[n] + i x[n+1024],
 * merged-twist Cooley-Tukey over 1024 complex points, twiddle table
 *   W[m+i] = exp(i*pi/2048 * (1024/(2m)) * (4*bitrev_log2m(i) + 1)),
 * W[m+i+1] = i*W[m+i] (i even) applied as an exact rotation.  The 10 radix-2 stages run in three
 * register layouts (lane, reg) <-> point n, as on the GPU:
 *   A  n = lane + 64 reg                          stages t = 512..64 (lane-uniform twiddles)
 *   B  n = 64 (lane >> 2) + 4 reg + (lane & 3)    stages t = 32..4   (per-lane base * U)
 *   C  n = 16 lane + reg                          stages t = 2, 1    (per-lane base * U)
 * The loops below walk (lane, register) exactly like one wavefront does, with the same IEEE-754
 * operation order (explicit fma, no contraction), so GPU and CPU results are bit-identical.
 * This is the reference's algorithm CLASS (tfhe + concrete-fft, Cargo.lock:168-179); approximate
 * w.r.t. the exact modes 0/1 (differences far below the noise), exact w.r.t. the GPU FFT path. */
#define FM 1024               /* complex points */
static double fW_re[FM], fW_im[FM];          /* W[1..1023] */
static double fU_re[3], fU_im[3];            /* exp(i*pi/4), exp(i*pi/8), exp(3i*pi/8) */
static double fMono_re[4096], fMono_im[4096];   /* exp(i*pi*k/2048): monomial evaluation table of mode 4 */
static int fmirror_ready = 0;
static unsigned brev_bits(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}
static void fmirror_init(void) {
    pthread_mutex_lock(&g_tab_mu);
    if (!fmirror_ready) {
        const double PI = 3.14159265358979323846;
        int d = 0;
        for (unsigned m = 1; m < FM; m <<= 1, d++)
            for (unsigned i = 0; i < m; i++) {
                const double e = (double)((FM / (2 * m)) * (4 * brev_bits(i, d) + 1));   /* exact integer */
                fW_re[m + i] = cos(PI * e / 2048.0);
                fW_im[m + i] = sin(PI * e / 2048.0);
            }
        fU_re[0] = cos(PI * 1.0 / 4.0); fU_im[0] = sin(PI * 1.0 / 4.0);
        fU_re[1] = cos(PI * 1.0 / 8.0); fU_im[1] = sin(PI * 1.0 / 8.0);
        fU_re[2] = cos(PI * 3.0 / 8.0); fU_im[2] = sin(PI * 3.0 / 8.0);
        /* through volatile pointers so that the pair is not fused into one sincos() call (whose results differ from
         * cos() / sin() in the last place for a few arguments); fhestring_amd/csrc/fft_tables.cpp does the same */
        static double (*volatile p_cos)(double) = cos;
        static double (*volatile p_sin)(double) = sin;
        for (int k = 0; k < 4096; k++) {
            fMono_re[k] = p_cos(PI * (double)k / 2048.0);
            fMono_im[k] = p_sin(PI * (double)k / 2048.0);
        }
        fmirror_ready = 1;
    }
    pthread_mutex_unlock(&g_tab_mu);
}
void orc_fft_tables(double *w_re, double *w_im, double *u_re, double *u_im) {
    fmirror_init();
    memcpy(w_re, fW_re, sizeof(fW_re)); memcpy(w_im, fW_im, sizeof(fW_im));
    memset(u_re, 0, 16 * sizeof(double)); memset(u_im, 0, 16 * sizeof(double));
    memcpy(u_re, fU_re, sizeof(fU_re)); memcpy(u_im, fU_im, sizeof(fU_im));
}
void orc_fft_mono_table(double *mono /*[4096][2]*/) {
    fmirror_init();
    for (int k = 0; k < 4096; k++) { mono[2 * k] = fMono_re[k]; mono[2 * k + 1] = fMono_im[k]; }
}
typedef struct { double r, i; } fcplx;
static inline fcplx fcmul(fcplx a, double wr, double wi) {
    fcplx t;
    t.r = fma(-a.i, wi, a.r * wr);
    t.i = fma(a.i, wr, a.r * wi);
    return t;
}
/* forward butterfly (a, b) <- (a + w' b, a - w' b), w' = w or i*w (rot): sum accumulated onto a with two
 * fused operations per component, difference = 2a - sum */
static inline void fbf_fwd(fcplx *a, fcplx *b, double wr, double wi, int rot) {
    const fcplx u = *a, v = *b;
    if (!rot) {
        a->r = fma(-v.i, wi, fma(v.r, wr, u.r));
        a->i = fma(v.i, wr, fma(v.r, wi, u.i));
    } else {
        a->r = fma(-v.i, wr, fma(-v.r, wi, u.r));
        a->i = fma(-v.i, wi, fma(v.r, wr, u.i));
    }
    b->r = fma(2.0, u.r, -a->r);
    b->i = fma(2.0, u.i, -a->i);
}
/* inverse butterfly (a, b) <- (a + b, (a - b) conj(w')) */
static inline void fbf_inv(fcplx *a, fcplx *b, double wr, double wi, int rot) {
    const fcplx u = *a, v = *b;
    fcplx d, q;
    a->r = u.r + v.r; a->i = u.i + v.i;
    d.r = u.r - v.r; d.i = u.i - v.i;
    q = fcmul(d, wr, -wi);
    if (!rot) *b = q;
    else { b->r = q.i; b->i = -q.r; }
}
static void fstages_uniform(fcplx *z, int inv) {
    for (int s = 0; s < 4; s++) {
        const int T = inv ? (1 << s) : (8 >> s), m = 8 / T;
        for (int i = 0; i < m; i++) {
            const int k = m == 1 ? 1 : ((m + i) & ~1);
            const double wr = fW_re[k], wi = fW_im[k];
            for (int r = 2 * i * T; r < 2 * i * T + T; r++) {
                if (inv) fbf_inv(&z[r], &z[r + T], wr, wi, i & 1); else fbf_fwd(&z[r], &z[r + T], wr, wi, i & 1);
            }
        }
    }
}
/* one in-lane stage: register distance tau, G = 8/tau groups, twiddle = base * U_G[g] (g even), rotated for odd g */
static void fstage_lane(fcplx *z, int inv, int tau, double br, double bi) {
    const int G = 8 / tau;
    for (int g = 0; g < G; g += 2) {
        double wr = br, wi = bi;
        if (g) {
            const int u = (G == 8 && g == 4) ? 1 : (G == 8 && g == 6) ? 2 : 0;
            fcplx b = {br, bi};
            const fcplx w = fcmul(b, fU_re[u], fU_im[u]);
            wr = w.r; wi = w.i;
        }
        for (int c = 2 * g * tau; c < 2 * g * tau + tau; c++) {
            if (inv) fbf_inv(&z[c], &z[c + tau], wr, wi, 0); else fbf_fwd(&z[c], &z[c + tau], wr, wi, 0);
        }
        if (G > 1)
            for (int c = 2 * (g + 1) * tau; c < 2 * (g + 1) * tau + tau; c++) {
                if (inv) fbf_inv(&z[c], &z[c + tau], wr, wi, 1); else fbf_fwd(&z[c], &z[c + tau], wr, wi, 1);
            }
    }
}
#define N_A(l, r) ((l) + 64 * (r))
#define N_B(l, r) (64 * ((l) >> 2) + 4 * (r) + ((l) & 3))
#define N_C(l, r) (16 * (l) + (r))
/* forward: x[2048] real coefficients (as doubles) -> F[lane][c] complex at array index 16*lane + c */
static void fmirror_forward(const double *x, fcplx (*F)[16]) {
    static __thread fcplx buf[FM];
    fcplx z[16];
    for (int l = 0; l < 64; l++) {
        for (int r = 0; r < 16; r++) { z[r].r = x[N_A(l, r)]; z[r].i = x[N_A(l, r) + 1024]; }
        fstages_uniform(z, 0);
        for (int r = 0; r < 16; r++) buf[N_A(l, r)] = z[r];
    }
    for (int l = 0; l < 64; l++) {
        const int hi = l >> 2;
        for (int r = 0; r < 16; r++) z[r] = buf[N_B(l, r)];
        for (int tau = 8, G = 1; tau >= 1; tau >>= 1, G <<= 1)
            fstage_lane(z, 0, tau, fW_re[16 * G + G * hi], fW_im[16 * G + G * hi]);
        for (int r = 0; r < 16; r++) F[l][r] = z[r];            /* parked in layout B */
    }
    for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) buf[N_B(l, r)] = F[l][r];
    for (int l = 0; l < 64; l++) {
        for (int c = 0; c < 16; c++) z[c] = buf[N_C(l, c)];
        fstage_lane(z, 0, 2, fW_re[256 + 4 * l], fW_im[256 + 4 * l]);
        fstage_lane(z, 0, 1, fW_re[512 + 8 * l], fW_im[512 + 8 * l]);
        for (int c = 0; c < 16; c++) F[l][c] = z[c];
    }
}
/* inverse (unscaled: 1/1024 is folded into the key): T[lane][c] -> x[2048] real coefficients */
static void fmirror_inverse(fcplx (*Tq)[16], double *x) {
    static __thread fcplx buf[FM];
    fcplx z[16];
    for (int l = 0; l < 64; l++) {
        for (int c = 0; c < 16; c++) z[c] = Tq[l][c];
        fstage_lane(z, 1, 1, fW_re[512 + 8 * l], fW_im[512 + 8 * l]);
        fstage_lane(z, 1, 2, fW_re[256 + 4 * l], fW_im[256 + 4 * l]);
        for (int c = 0; c < 16; c++) buf[N_C(l, c)] = z[c];
    }
    for (int l = 0; l < 64; l++) {
        const int hi = l >> 2;
        for (int r = 0; r < 16; r++) z[r] = buf[N_B(l, r)];
        for (int tau = 1, G = 8; tau <= 8; tau <<= 1, G >>= 1)
            fstage_lane(z, 1, tau, fW_re[16 * G + G * hi], fW_im[16 * G + G * hi]);
        for (int r = 0; r < 16; r++) Tq[l][r] = z[r];           /* parked in layout B */
    }
    for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) buf[N_B(l, r)] = Tq[l][r];
    for (int l = 0; l < 64; l++) {
        for (int r = 0; r < 16; r++) z[r] = buf[N_A(l, r)];
        fstages_uniform(z, 1);
        for (int r = 0; r < 16; r++) { x[N_A(l, r)] = z[r].r; x[N_A(l, r) + 1024] = z[r].i; }
    }
}
/* v_fract_f64: x - floor(x), clamped below 1 */
static inline double ffract(double x) {
    const double r = x - floor(x);
    return r < 1.0 ? r : 0x1.fffffffffffffp-1;
}
/* torus value (mod 2^64) of t * 2^64: the Fourier-domain key carries 2^-64, so the increment is the fractional
 * part of the inverse transform's output; 1 + f moves it into the 52 mantissa bits (one rounding at 2^-52) */
static inline u64 fmirror_to_torus(double t) {
    const double g = 1.0 + ffract(t);
    u64 b;
    memcpy(&b, &g, 8);
    return b << 12;
}
/* BSK polynomial -> Fourier domain in the kernel's layout [c][lane] (re, im), pre-scaled by 2^-74 (1/1024 and 2^-64) */
static void fmirror_bsk_poly(const u64 *src, double *dst /* [16][64][2] */) {
    static __thread double x[POLY_N];
    static __thread fcplx F[64][16];
    for (int n = 0; n < POLY_N; n++)   /* 58 significant bits: both halves convert exactly, one rounding in the add */
        x[n] = (double)(int32_t)(src[n] >> 32) * 0x1p32 + (double)(uint32_t)src[n];
    fmirror_forward(x, F);
    for (int L = 0; L < 64; L++)
        for (int c = 0; c < 16; c++) {
            dst[(c * 64 + L) * 2 + 0] = F[L][c].r * 0x1p-74;
            dst[(c * 64 + L) * 2 + 1] = F[L][c].i * 0x1p-74;
        }
}
void orc_fft_bsk_convert(const u64 *bsk_quantised, double *out /* [742*4][16][64][2] */) {
    fmirror_init();
    for (size_t p = 0; p < BSK_POLYS; p++) fmirror_bsk_poly(bsk_quantised + p * POLY_N, out + p * 2 * FM);
}

/* ---- mode 6: the CPU BASELINE of bench.py -- f64 FFT external product, AVX2 + FMA ----------------------------------
 * The algorithm class of the reference's engine (tfhe 0.5.2 + concrete-fft 0.4.0, Cargo.lock:168-179: a folded
 * 1024-point complex f64 FFT, SIMD through `pulp`), written for 4-wide vectors over split re / im arrays: merged-twist
 * Cooley-Tukey forward (natural order in, bit-reversed out, no separate twist pass and no permutation), Gentleman-Sande
 * inverse, the same twiddle table W as mode 3; the two narrowest stages use in-register permutes.  Torus conversion,
 * rotation + decomposition and the pointwise products are vector loops as well.  It exists so that the CPU figure
 * beside the GPU number is an honest one (the scalar port was ~2x slower per core than published tfhe-rs);
 * it is decrypt-checked against the exact modes (tests/test_oracle_pbs.py) and is NEVER the parity oracle. */
static double vW1_re[FM / 2 * 2], vW1_im[FM / 2 * 2];   /* twiddles of the distance-1 stage in unpack order */
static int vec_ready = 0;
static void vec_init(void) {
    fmirror_init();
    pthread_mutex_lock(&g_tab_mu);
    if (!vec_ready) {
        /* stage t = 1: m = 512 groups of 2; 8 consecutive points = pairs (0,1)(2,3)(4,5)(6,7) = groups g..g+3; after
         * unpacklo / unpackhi of [x0..x3] and [x4..x7] the lanes hold pairs in the order g, g+2, g+1, g+3 */
        for (int g = 0; g < 512; g += 4) {
            const int o[4] = {g, g + 2, g + 1, g + 3};
            for (int k = 0; k < 4; k++) { vW1_re[g + k] = fW_re[512 + o[k]]; vW1_im[g + k] = fW_im[512 + o[k]]; }
        }
        vec_ready = 1;
    }
    pthread_mutex_unlock(&g_tab_mu);
}
#define VLD(p) _mm256_loadu_pd(p)
#define VST(p, v) _mm256_storeu_pd(p, v)
/* forward butterfly on vectors: (a, b) <- (a + w b, a - w b) */
#define VBF_FWD(ar, ai, br, bi, wr, wi) do { \
        const __m256d tr_ = _mm256_fnmadd_pd(bi, wi, _mm256_mul_pd(br, wr)); \
        const __m256d ti_ = _mm256_fmadd_pd(bi, wr, _mm256_mul_pd(br, wi)); \
        br = _mm256_sub_pd(ar, tr_); bi = _mm256_sub_pd(ai, ti_); \
        ar = _mm256_add_pd(ar, tr_); ai = _mm256_add_pd(ai, ti_); } while (0)
/* inverse butterfly: (a, b) <- (a + b, (a - b) conj(w)) */
#define VBF_INV(ar, ai, br, bi, wr, wi) do { \
        const __m256d dr_ = _mm256_sub_pd(ar, br), di_ = _mm256_sub_pd(ai, bi); \
        ar = _mm256_add_pd(ar, br); ai = _mm256_add_pd(ai, bi); \
        br = _mm256_fmadd_pd(di_, wi, _mm256_mul_pd(dr_, wr)); \
        bi = _mm256_fnmadd_pd(dr_, wi, _mm256_mul_pd(di_, wr)); } while (0)
static void vec_forward(double *restrict re, double *restrict im) {
    /* stages t = 512 .. 4 two at a time (one pass over memory per pair): distance t with W[m + i], then distance t/2
     * with W[2m + 2i] (lower half of the block) and W[2m + 2i + 1] (upper half) */
    for (int m = 1, t = FM / 2; t >= 8; m <<= 2, t >>= 2)
        for (int i = 0; i < m; i++) {
            const __m256d w1r = _mm256_set1_pd(fW_re[m + i]), w1i = _mm256_set1_pd(fW_im[m + i]);
            const __m256d w2r = _mm256_set1_pd(fW_re[2 * m + 2 * i]), w2i = _mm256_set1_pd(fW_im[2 * m + 2 * i]);
            const __m256d w3r = _mm256_set1_pd(fW_re[2 * m + 2 * i + 1]), w3i = _mm256_set1_pd(fW_im[2 * m + 2 * i + 1]);
            double *pr = re + 2 * i * t, *pi = im + 2 * i * t;
            const int h = t / 2;
            for (int j = 0; j < h; j += 4) {
                __m256d x0r = VLD(pr + j), x0i = VLD(pi + j), x1r = VLD(pr + j + h), x1i = VLD(pi + j + h);
                __m256d x2r = VLD(pr + j + t), x2i = VLD(pi + j + t), x3r = VLD(pr + j + t + h), x3i = VLD(pi + j + t + h);
                VBF_FWD(x0r, x0i, x2r, x2i, w1r, w1i);
                VBF_FWD(x1r, x1i, x3r, x3i, w1r, w1i);
                VBF_FWD(x0r, x0i, x1r, x1i, w2r, w2i);
                VBF_FWD(x2r, x2i, x3r, x3i, w3r, w3i);
                VST(pr + j, x0r); VST(pi + j, x0i); VST(pr + j + h, x1r); VST(pi + j + h, x1i);
                VST(pr + j + t, x2r); VST(pi + j + t, x2i); VST(pr + j + t + h, x3r); VST(pi + j + t + h, x3i);
            }
        }
    for (int g = 0; g < 256; g += 2) {      /* t = 2: groups of 4 points, two groups per pass */
        double *pr = re + 4 * g, *pi = im + 4 * g;
        const __m256d v0r = VLD(pr), v1r = VLD(pr + 4), v0i = VLD(pi), v1i = VLD(pi + 4);
        __m256d ar = _mm256_permute2f128_pd(v0r, v1r, 0x20), br = _mm256_permute2f128_pd(v0r, v1r, 0x31);
        __m256d ai = _mm256_permute2f128_pd(v0i, v1i, 0x20), bi = _mm256_permute2f128_pd(v0i, v1i, 0x31);
        const __m256d wr = _mm256_setr_pd(fW_re[256 + g], fW_re[256 + g], fW_re[257 + g], fW_re[257 + g]);
        const __m256d wi = _mm256_setr_pd(fW_im[256 + g], fW_im[256 + g], fW_im[257 + g], fW_im[257 + g]);
        VBF_FWD(ar, ai, br, bi, wr, wi);
        VST(pr, _mm256_permute2f128_pd(ar, br, 0x20)); VST(pr + 4, _mm256_permute2f128_pd(ar, br, 0x31));
        VST(pi, _mm256_permute2f128_pd(ai, bi, 0x20)); VST(pi + 4, _mm256_permute2f128_pd(ai, bi, 0x31));
    }
    for (int g = 0; g < 512; g += 4) {      /* t = 1: pairs of neighbours, four groups per pass */
        double *pr = re + 2 * g, *pi = im + 2 * g;
        const __m256d v0r = VLD(pr), v1r = VLD(pr + 4), v0i = VLD(pi), v1i = VLD(pi + 4);
        __m256d ar = _mm256_unpacklo_pd(v0r, v1r), br = _mm256_unpackhi_pd(v0r, v1r);
        __m256d ai = _mm256_unpacklo_pd(v0i, v1i), bi = _mm256_unpackhi_pd(v0i, v1i);
        const __m256d wr = VLD(vW1_re + g), wi = VLD(vW1_im + g);
        VBF_FWD(ar, ai, br, bi, wr, wi);
        VST(pr, _mm256_unpacklo_pd(ar, br)); VST(pr + 4, _mm256_unpackhi_pd(ar, br));
        VST(pi, _mm256_unpacklo_pd(ai, bi)); VST(pi + 4, _mm256_unpackhi_pd(ai, bi));
    }
}
static void vec_inverse(double *restrict re, double *restrict im) {
    for (int g = 0; g < 512; g += 4) {
        double *pr = re + 2 * g, *pi = im + 2 * g;
        const __m256d v0r = VLD(pr), v1r = VLD(pr + 4), v0i = VLD(pi), v1i = VLD(pi + 4);
        __m256d ar = _mm256_unpacklo_pd(v0r, v1r), br = _mm256_unpackhi_pd(v0r, v1r);
        __m256d ai = _mm256_unpacklo_pd(v0i, v1i), bi = _mm256_unpackhi_pd(v0i, v1i);
        const __m256d wr = VLD(vW1_re + g), wi = VLD(vW1_im + g);
        VBF_INV(ar, ai, br, bi, wr, wi);
        VST(pr, _mm256_unpacklo_pd(ar, br)); VST(pr + 4, _mm256_unpackhi_pd(ar, br));
        VST(pi, _mm256_unpacklo_pd(ai, bi)); VST(pi + 4, _mm256_unpackhi_pd(ai, bi));
    }
    for (int g = 0; g < 256; g += 2) {
        double *pr = re + 4 * g, *pi = im + 4 * g;
        const __m256d v0r = VLD(pr), v1r = VLD(pr + 4), v0i = VLD(pi), v1i = VLD(pi + 4);
        __m256d ar = _mm256_permute2f128_pd(v0r, v1r, 0x20), br = _mm256_permute2f128_pd(v0r, v1r, 0x31);
        __m256d ai = _mm256_permute2f128_pd(v0i, v1i, 0x20), bi = _mm256_permute2f128_pd(v0i, v1i, 0x31);
        const __m256d wr = _mm256_setr_pd(fW_re[256 + g], fW_re[256 + g], fW_re[257 + g], fW_re[257 + g]);
        const __m256d wi = _mm256_setr_pd(fW_im[256 + g], fW_im[256 + g], fW_im[257 + g], fW_im[257 + g]);
        VBF_INV(ar, ai, br, bi, wr, wi);
        VST(pr, _mm256_permute2f128_pd(ar, br, 0x20)); VST(pr + 4, _mm256_permute2f128_pd(ar, br, 0x31));
        VST(pi, _mm256_permute2f128_pd(ai, bi, 0x20)); VST(pi + 4, _mm256_permute2f128_pd(ai, bi, 0x31));
    }
    /* stages t = 4 .. 512 two at a time: distance h = t/2 first (groups 2i, 2i + 1 of the finer level), then distance t */
    for (int t = 8, m = FM / 16; t < FM; t <<= 2, m >>= 2)
        for (int i = 0; i < m; i++) {
            const __m256d w1r = _mm256_set1_pd(fW_re[m + i]), w1i = _mm256_set1_pd(fW_im[m + i]);
            const __m256d w2r = _mm256_set1_pd(fW_re[2 * m + 2 * i]), w2i = _mm256_set1_pd(fW_im[2 * m + 2 * i]);
            const __m256d w3r = _mm256_set1_pd(fW_re[2 * m + 2 * i + 1]), w3i = _mm256_set1_pd(fW_im[2 * m + 2 * i + 1]);
            double *pr = re + 2 * i * t, *pi = im + 2 * i * t;
            const int h = t / 2;
            for (int j = 0; j < h; j += 4) {
                __m256d x0r = VLD(pr + j), x0i = VLD(pi + j), x1r = VLD(pr + j + h), x1i = VLD(pi + j + h);
                __m256d x2r = VLD(pr + j + t), x2i = VLD(pi + j + t), x3r = VLD(pr + j + t + h), x3i = VLD(pi + j + t + h);
                VBF_INV(x0r, x0i, x1r, x1i, w2r, w2i);
                VBF_INV(x2r, x2i, x3r, x3i, w3r, w3i);
                VBF_INV(x0r, x0i, x2r, x2i, w1r, w1i);
                VBF_INV(x1r, x1i, x3r, x3i, w1r, w1i);
                VST(pr + j, x0r); VST(pi + j, x0i); VST(pr + j + h, x1r); VST(pi + j + h, x1i);
                VST(pr + j + t, x2r); VST(pi + j + t, x2i); VST(pr + j + t + h, x3r); VST(pi + j + t + h, x3i);
            }
        }
}
/* key polynomial -> [re 1024 | im 1024] in the forward transform's output order, pre-scaled by 2^-74 (1/1024, 2^-64) */
static void vec_bsk_poly(const u64 *src, double *dst) {
    double *re = dst, *im = dst + FM;
    for (int n = 0; n < FM; n++) {
        re[n] = (double)(int32_t)(src[n] >> 32) * 0x1p32 + (double)(uint32_t)src[n];
        im[n] = (double)(int32_t)(src[n + FM] >> 32) * 0x1p32 + (double)(uint32_t)src[n + FM];
    }
    vec_forward(re, im);
    for (int n = 0; n < 2 * FM; n++) dst[n] *= 0x1p-74;
}
/* acc += torus(t), t = increment / 2^64 (fractional part; one rounding at 2^-52 like the GPU kernel's conversion) */
static inline void vec_acc_add(u64 *restrict acc, const double *restrict tv) {
    const __m256d one = _mm256_set1_pd(1.0);
    for (int n = 0; n < FM; n += 4) {
        const __m256d x = VLD(tv + n);
        const __m256d g = _mm256_add_pd(one, _mm256_sub_pd(x, _mm256_floor_pd(x)));
        const __m256i b = _mm256_slli_epi64(_mm256_castpd_si256(g), 12);
        _mm256_storeu_si256((__m256i *)(acc + n), _mm256_add_epi64(_mm256_loadu_si256((const __m256i *)(acc + n)), b));
    }
}
static void poly_rotate(const u64 *in, unsigned a, u64 *out);
static void blind_rotate_vec(const double *restrict bsk_vec, const u32 *ms, const u64 *lut, u64 *restrict acc /* [2][N] */) {
    static __thread double F[2][2 * FM] __attribute__((aligned(32)));
    static __thread double T[2 * FM] __attribute__((aligned(32)));
    static __thread int32_t dig32[POLY_N] __attribute__((aligned(32)));
    memset(acc, 0, POLY_N * sizeof(u64));
    poly_rotate(lut, (2 * POLY_N - ms[LWE_N]) & (2 * POLY_N - 1), acc + POLY_N);
    for (int i = 0; i < LWE_N; i++) {
        const unsigned a = ms[i];
        if (a == 0) continue;
        const unsigned s = a & (POLY_N - 1);
        const u64 sg = (a >> 11) & 1 ? ~0ull : 0ull;      /* X^a = -X^(a - N) for a >= N: two's complement through (v ^ sg) - sg */
        for (int c = 0; c < 2; c++) {
            const u64 *restrict ac = acc + c * POLY_N;
            double *restrict re = F[c];                  /* [re 1024 | im 1024] = coefficients 0..2047 */
            /* digit of X^a acc - acc: closest multiple of 2^41 as a signed 23-bit digit; two branch-free ranges */
            for (unsigned n = 0; n < s; n++) {
                const u64 v = (((u64)0 - ac[n + POLY_N - s]) ^ sg) - sg;
                dig32[n] = (int32_t)((i64)(v - ac[n] + (1ull << 40)) >> 41);
            }
            for (unsigned n = s; n < POLY_N; n++) {
                const u64 v = (ac[n - s] ^ sg) - sg;
                dig32[n] = (int32_t)((i64)(v - ac[n] + (1ull << 40)) >> 41);
            }
            for (int n = 0; n < POLY_N; n++) re[n] = (double)dig32[n];
            vec_forward(re, re + FM);
        }
        for (int col = 0; col < 2; col++) {
            const double *b0 = bsk_vec + ((((size_t)i * 2 + 0) * 2 + col)) * 2 * FM;
            const double *b1 = bsk_vec + ((((size_t)i * 2 + 1) * 2 + col)) * 2 * FM;
            for (int q = 0; q < FM; q += 4) {
                const __m256d f0r = VLD(F[0] + q), f0i = VLD(F[0] + FM + q), f1r = VLD(F[1] + q), f1i = VLD(F[1] + FM + q);
                const __m256d k0r = VLD(b0 + q), k0i = VLD(b0 + FM + q), k1r = VLD(b1 + q), k1i = VLD(b1 + FM + q);
                __m256d rr = _mm256_mul_pd(f0r, k0r), ii = _mm256_mul_pd(f0r, k0i);
                rr = _mm256_fnmadd_pd(f0i, k0i, rr); ii = _mm256_fmadd_pd(f0i, k0r, ii);
                rr = _mm256_fmadd_pd(f1r, k1r, rr);  ii = _mm256_fmadd_pd(f1r, k1i, ii);
                rr = _mm256_fnmadd_pd(f1i, k1i, rr); ii = _mm256_fmadd_pd(f1i, k1r, ii);
                VST(T + q, rr); VST(T + FM + q, ii);
            }
            vec_inverse(T, T + FM);
            vec_acc_add(acc + col * POLY_N, T);
            vec_acc_add(acc + col * POLY_N + FM, T + FM);
        }
    }
}

/* ---- server key --------------------------------------------------------- */
typedef struct {
    u64 *bsk;      /* [742][2 rows][2 cols][2048] std domain, quantised to 2^6 */
    u64 *ksk;      /* [2048][5][743]                                          */
    u64 *bsk_ntt;  /* [742][2 rows][2 cols][2 limbs][2048] Goldilocks NTT     */
    double *bsk_fft; /* [742][2 rows][2 cols][1024] complex (re,im): f64-FFT variant (mode 2) */
    double *bsk_fm;  /* [742][2 rows][2 cols][16][64][2]: mirror of the GPU FFT kernel (mode 3), lazy */
    double *bsk_vec; /* [742][2 rows][2 cols][re 1024 | im 1024]: Fourier-domain key of the vectorised CPU baseline (mode 6), lazy */
    double *bsk_mb;  /* [371][K1,K2,K3][2 rows][2 cols][16][64][2]: pair key of mode 4 (orc_server_key_set_mb2) */
    u64 *bsk_mb_q7;  /* [371][K1,K2,K3][2 rows][2 cols][2048] std domain, rounded to multiples of 2^7: mode 5 */
} orc_server_key;

u64 orc_bsk_words(void) { return BSK_WORDS; }
u64 orc_ksk_words(void) { return KSK_WORDS; }

/* Round every BSK coefficient to the nearest multiple of 2^6.  The on-device
 * key format is a 58-bit torus (see DESIGN.md "BSK precision"); keys generated
 * by orc_keygen are already on that grid so this is then the identity. */
void orc_bsk_quantize(u64 *bsk, u64 n) {
    const u64 half = 1ull << (BSK_QUANT_BITS - 1);
    const u64 mask = ~((1ull << BSK_QUANT_BITS) - 1);
    for (u64 i = 0; i < n; i++) bsk[i] = (bsk[i] + half) & mask;
}

/* negacyclic a (*) S for binary S, wrapping u64 */
static void negacyclic_mul_binary(const u64 *a, const u64 *s_bits, u64 *out) {
    memset(out, 0, POLY_N * sizeof(u64));
    for (unsigned j = 0; j < POLY_N; j++) {
        if (!s_bits[j]) continue;
        for (unsigned k = 0; k < j; k++) out[k] -= a[k + POLY_N - j];
        for (unsigned k = j; k < POLY_N; k++) out[k] += a[k - j];
    }
}

/* Key generation (restates what tfhe::integer::gen_keys_radix produces for the
 * reference at src/client_key.rs:31; own seeded RNG).
 *   lwe_sk[742], glwe_sk[2048]: binary secret keys (one u64 per bit)
 *   bsk: GGSW_i encrypts lwe_sk[i]; row 0 = GLWE(-s_i*S(X)*2^41), row 1 = GLWE(s_i*2^41)
 *   ksk[i][l] = LWE_small(glwe_sk[i] * 2^(64-3(l+1)))                      */
void orc_keygen(u64 seed, u64 *lwe_sk, u64 *glwe_sk, u64 *bsk, u64 *ksk) {
    orc_rng r = { seed };
    for (int i = 0; i < LWE_N; i++) lwe_sk[i] = rng_u64(&r) >> 63;
    for (int i = 0; i < POLY_N; i++) glwe_sk[i] = rng_u64(&r) >> 63;
    const u64 qmask = ~((1ull << BSK_QUANT_BITS) - 1);
    const u64 qhalf = 1ull << (BSK_QUANT_BITS - 1);
    u64 *prod = (u64 *)malloc(POLY_N * sizeof(u64));
    for (int i = 0; i < LWE_N; i++) {
        for (int row = 0; row < 2; row++) {
            u64 *mask = bsk + (((size_t)i * 2 + row) * 2 + 0) * POLY_N;
            u64 *body = bsk + (((size_t)i * 2 + row) * 2 + 1) * POLY_N;
            for (int n = 0; n < POLY_N; n++) mask[n] = rng_u64(&r) & qmask;
            negacyclic_mul_binary(mask, glwe_sk, prod);
            for (int n = 0; n < POLY_N; n++) {
                u64 m;
                if (row == 0) m = (u64)0 - (lwe_sk[i] * glwe_sk[n] << (64 - PBS_BASE_LOG));
                else m = (n == 0) ? (lwe_sk[i] << (64 - PBS_BASE_LOG)) : 0;
                u64 e = rng_noise(&r, GLWE_NOISE);
                body[n] = (prod[n] + e + m + qhalf) & qmask;
            }
        }
    }
    free(prod);
    for (int i = 0; i < BIG_N; i++) {
        for (int l = 0; l < KS_LEVEL; l++) {
            u64 *ct = ksk + ((size_t)i * KS_LEVEL + l) * SMALL_CT;
            u64 acc = 0;
            for (int j = 0; j < LWE_N; j++) {
                ct[j] = rng_u64(&r);
                acc += ct[j] * lwe_sk[j];
            }
            ct[LWE_N] = acc + rng_noise(&r, LWE_NOISE) +
                        (glwe_sk[i] << (64 - KS_BASE_LOG * (l + 1)));
        }
    }
}

/* Pair key of mode 4: GGSWs of s(1-s'), (1-s)s', s s' for (s, s') = (lwe_sk[2p], lwe_sk[2p+1]); same format, noise and
 * grid as the bootstrapping key above.  bsk_mb2: [371][3][2 rows][2 cols][2048]. */
void orc_keygen_mb2(u64 seed, const u64 *lwe_sk, const u64 *glwe_sk, u64 *bsk_mb2) {
    orc_rng r = { seed ^ 0x6d62325f6b657973ull };
    const u64 qmask = ~((1ull << BSK_QUANT_BITS) - 1);
    const u64 qhalf = 1ull << (BSK_QUANT_BITS - 1);
    u64 *prod = (u64 *)malloc(POLY_N * sizeof(u64));
    for (int g = 0; g < (LWE_N / 2) * 3; g++) {
        const int p = g / 3, t = g % 3;
        const u64 s1 = lwe_sk[2 * p], s2 = lwe_sk[2 * p + 1];
        const u64 msg = t == 0 ? (s1 & (1 - s2)) : t == 1 ? ((1 - s1) & s2) : (s1 & s2);
        for (int row = 0; row < 2; row++) {
            u64 *mask = bsk_mb2 + (((size_t)g * 2 + row) * 2 + 0) * POLY_N;
            u64 *body = bsk_mb2 + (((size_t)g * 2 + row) * 2 + 1) * POLY_N;
            for (int n = 0; n < POLY_N; n++) mask[n] = rng_u64(&r) & qmask;
            negacyclic_mul_binary(mask, glwe_sk, prod);
            for (int n = 0; n < POLY_N; n++) {
                u64 m;
                if (row == 0) m = (u64)0 - (msg * glwe_sk[n] << (64 - PBS_BASE_LOG));
                else m = (n == 0) ? (msg << (64 - PBS_BASE_LOG)) : 0;
                u64 e = rng_noise(&r, GLWE_NOISE);
                body[n] = (prod[n] + e + m + qhalf) & qmask;
            }
        }
    }
    free(prod);
}

orc_server_key *orc_server_key_new(const u64 *bsk, const u64 *ksk) {
    g_init_tables();
    orc_server_key *k = (orc_server_key *)calloc(1, sizeof(*k));
    k->bsk = (u64 *)malloc(BSK_WORDS * sizeof(u64));
    k->ksk = (u64 *)malloc(KSK_WORDS * sizeof(u64));
    k->bsk_ntt = (u64 *)malloc(2 * BSK_WORDS * sizeof(u64));
    memcpy(k->bsk, bsk, BSK_WORDS * sizeof(u64));
    memcpy(k->ksk, ksk, KSK_WORDS * sizeof(u64));
    orc_bsk_quantize(k->bsk, BSK_WORDS);
    const u64 lm = (1ull << 29) - 1;
    for (size_t p = 0; p < BSK_POLYS; p++) {
        const u64 *src = k->bsk + p * POLY_N;
        u64 *l0 = k->bsk_ntt + (2 * p) * POLY_N, *l1 = l0 + POLY_N;
        for (int n = 0; n < POLY_N; n++) {
            u64 v = src[n] >> BSK_QUANT_BITS;
            l0[n] = v & lm;
            l1[n] = v >> 29;
        }
        g_ntt_fwd(l0);
        g_ntt_fwd(l1);
    }
    fft_init();
    k->bsk_fft = (double *)malloc(BSK_POLYS * 2 * FFT_N * sizeof(double));
    {
        double *tmp = (double *)malloc(POLY_N * sizeof(double));
        for (size_t p = 0; p < BSK_POLYS; p++) {
            const u64 *src = k->bsk + p * POLY_N;
            for (int n = 0; n < POLY_N; n++) tmp[n] = (double)(i64)src[n];   /* signed torus, 53-bit rounding */
            fft_forward_poly(tmp, k->bsk_fft + p * 2 * FFT_N, k->bsk_fft + p * 2 * FFT_N + FFT_N);
        }
        free(tmp);
    }
    return k;
}
/* Pair key of mode 4 (the product's FHS_ARITH_F64_FFT_MB2, fhestring_amd/csrc/fftmb_kernels.hip): for each pair of LWE
 * key bits (s, s') GGSWs K1, K2, K3 of s(1-s'), (1-s)s', s s' in the bootstrapping key's format
 * [371][3][2 rows][2 cols][2048]; transformed like the device does (same forward transform, same scaling). */
#define MB2_POLYS ((size_t)(LWE_N / 2) * 3 * 4)
void orc_server_key_set_mb2(orc_server_key *k, const u64 *bsk_mb2) {
    fmirror_init();
    u64 *q = (u64 *)malloc(MB2_POLYS * POLY_N * sizeof(u64));
    memcpy(q, bsk_mb2, MB2_POLYS * POLY_N * sizeof(u64));
    orc_bsk_quantize(q, MB2_POLYS * POLY_N);
    free(k->bsk_mb);
    k->bsk_mb = (double *)malloc(MB2_POLYS * 2 * FM * sizeof(double));
    for (size_t p = 0; p < MB2_POLYS; p++) fmirror_bsk_poly(q + p * POLY_N, k->bsk_mb + p * 2 * FM);
    /* mode 5 (exact arithmetic): the 57-bit torus grid, so that the integer result stays inside the product's CRT range */
    for (size_t i = 0; i < MB2_POLYS * POLY_N; i++) q[i] = (bsk_mb2[i] + (1ull << 6)) & ~((1ull << 7) - 1);
    free(k->bsk_mb_q7);
    k->bsk_mb_q7 = q;
}

void orc_server_key_free(orc_server_key *k) {
    if (!k) return;
    free(k->bsk); free(k->ksk); free(k->bsk_ntt); free(k->bsk_fft); free(k->bsk_fm); free(k->bsk_vec); free(k->bsk_mb); free(k->bsk_mb_q7); free(k);
}

/* ---- client side (src/client_key.rs:85-106 via RadixClientKey) ---------- */
/* encrypt one shortint block value m in [0,32) (padding bit included) under
 * the big key */
void orc_encrypt_block(const u64 *glwe_sk, u64 m, u64 *rng_state, u64 *ct) {
    orc_rng r = { *rng_state };
    u64 acc = 0;
    for (int j = 0; j < BIG_N; j++) {
        ct[j] = rng_u64(&r);
        acc += ct[j] * glwe_sk[j];
    }
    ct[BIG_N] = acc + rng_noise(&r, GLWE_NOISE) + (m << DELTA_LOG);
    *rng_state = r.s;
}
u64 orc_phase(const u64 *glwe_sk, const u64 *ct) {
    u64 acc = 0;
    for (int j = 0; j < BIG_N; j++) acc += ct[j] * glwe_sk[j];
    return ct[BIG_N] - acc;
}
/* decrypt to [0,32): 4 message+carry bits and the padding bit */
u64 orc_decrypt_block(const u64 *glwe_sk, const u64 *ct) {
    u64 ph = orc_phase(glwe_sk, ct);
    return ((ph + (1ull << (DELTA_LOG - 1))) >> DELTA_LOG) & 31;
}
/* u8 -> 4 blocks of 2 bits, little endian (src/ciphertext/fheasciichar.rs:27-29) */
void orc_encrypt_char(const u64 *glwe_sk, unsigned v, u64 *rng_state, u64 *ct4) {
    for (int b = 0; b < 4; b++)
        orc_encrypt_block(glwe_sk, (v >> (2 * b)) & 3, rng_state, ct4 + (size_t)b * BIG_CT);
}
/* decrypt::<u8>: sum block_i * 4^i (mod 256), carries included like tfhe's radix decrypt */
unsigned orc_decrypt_char(const u64 *glwe_sk, const u64 *ct4) {
    unsigned v = 0;
    for (int b = 0; b < 4; b++)
        v += (unsigned)(orc_decrypt_block(glwe_sk, ct4 + (size_t)b * BIG_CT) & 15) << (2 * b);
    return v & 255;
}

/* ---- LUT generation (SURVEY.md Appendix A "LUT generation") ------------- */
/* f_table[16]: output block values (mod 32 allowed, e.g. 31 for -1) */
void orc_make_lut(const u64 *f_table, u64 *lut) {
    const int box = POLY_N / 16, half = box / 2;
    u64 tmp[POLY_N];
    for (int x = 0; x < 16; x++)
        for (int t = 0; t < box; t++) tmp[x * box + t] = f_table[x] << DELTA_LOG;
    for (int i = 0; i < POLY_N - half; i++) lut[i] = tmp[i + half];
    for (int t = 0; t < half; t++) lut[POLY_N - half + t] = (u64)0 - tmp[t];
}

/* acc[j] -= d * row[j] for a balanced base-8 digit d in [-4, 3], d != 0: shifts and adds on 4-wide vectors (AVX2 has no
 * 64-bit multiply; the compiler's emulation through 32-bit products is ~2x slower).  Used by the CPU-baseline keyswitch
 * below; the result is the same wrapping u64 arithmetic as the plain loop. */
static inline void ks_row_mac(u64 *restrict acc, const u64 *restrict row, int d) {
    const int n4 = SMALL_CT & ~3;
    const int neg = d < 0, ad = neg ? -d : d;            /* |d| in 1..4 */
    for (int j = 0; j < n4; j += 4) {
        const __m256i x = _mm256_loadu_si256((const __m256i *)(row + j));
        __m256i m = ad == 1 ? x : ad == 2 ? _mm256_slli_epi64(x, 1) : ad == 3 ? _mm256_add_epi64(_mm256_slli_epi64(x, 1), x)
                                                                             : _mm256_slli_epi64(x, 2);
        __m256i a = _mm256_loadu_si256((__m256i *)(acc + j));
        a = neg ? _mm256_add_epi64(a, m) : _mm256_sub_epi64(a, m);
        _mm256_storeu_si256((__m256i *)(acc + j), a);
    }
    for (int j = n4; j < SMALL_CT; j++) acc[j] -= (u64)(i64)d * row[j];
}

/* ---- keyswitch + modulus switch ------------------------------------------ */
/* out[743]: a~_0..a~_741, b~ in [0,4096) */
void orc_keyswitch_modswitch(const orc_server_key *k, const u64 *in, u32 *out) {
    u64 acc[SMALL_CT];
    memset(acc, 0, sizeof(acc));
    acc[LWE_N] = in[BIG_N];
    const int tot = KS_BASE_LOG * KS_LEVEL; /* 15 */
    for (int i = 0; i < BIG_N; i++) {
        /* closest representable on 15 bits, then balanced base-8 digits */
        u64 v = (in[i] + (1ull << (63 - tot))) >> (64 - tot);
        for (int l = KS_LEVEL - 1; l >= 0; l--) { /* least significant level first */
            i64 d = (i64)(v & 7);
            v >>= 3;
            if (d >= 4) { d -= 8; v += 1; }
            if (d == 0) continue;
            ks_row_mac(acc, k->ksk + ((size_t)i * KS_LEVEL + l) * SMALL_CT, (int)d);
        }
    }
    for (int j = 0; j < SMALL_CT; j++)
        out[j] = (u32)(((acc[j] + (1ull << 51)) >> 52) & 4095);
}
/* raw keyswitch output (before mod switch), for kernel-level tests */
void orc_keyswitch(const orc_server_key *k, const u64 *in, u64 *out) {
    memset(out, 0, SMALL_CT * sizeof(u64));
    out[LWE_N] = in[BIG_N];
    const int tot = KS_BASE_LOG * KS_LEVEL;
    for (int i = 0; i < BIG_N; i++) {
        u64 v = (in[i] + (1ull << (63 - tot))) >> (64 - tot);
        for (int l = KS_LEVEL - 1; l >= 0; l--) {
            i64 d = (i64)(v & 7);
            v >>= 3;
            if (d >= 4) { d -= 8; v += 1; }
            if (d == 0) continue;
            const u64 *row = k->ksk + ((size_t)i * KS_LEVEL + l) * SMALL_CT;
            u64 du = (u64)d;
            for (int j = 0; j < SMALL_CT; j++) out[j] -= du * row[j];
        }
    }
}

/* ---- blind rotation ------------------------------------------------------- */
/* out = X^a * in, a in [0, 2N) */
static void poly_rotate(const u64 *in, unsigned a, u64 *out) {
    unsigned s = a & (POLY_N - 1);
    int neg = (a >> 11) & 1;
    for (unsigned n = 0; n < POLY_N; n++) {
        u64 v = (n >= s) ? in[n - s] : (u64)0 - in[n + POLY_N - s];
        out[n] = neg ? (u64)0 - v : v;
    }
}
static inline i64 pbs_digit(u64 x) { /* closest multiple of 2^41, as signed 23-bit digit */
    u64 v = ((x + (1ull << 40)) >> 41) & ((1ull << 23) - 1);
    return (v >= (1ull << 22)) ? (i64)v - (1ll << 23) : (i64)v;
}

/* schoolbook ground truth: res += d (*) b  mod (X^N+1, 2^64) */
static void negacyclic_mac_schoolbook(const i64 *d, const u64 *b, u64 *res) {
    for (unsigned i = 0; i < POLY_N; i++) {
        u64 di = (u64)d[i];
        if (!di) continue;
        for (unsigned j = 0; j < POLY_N - i; j++) res[i + j] += di * b[j];
        for (unsigned j = POLY_N - i; j < POLY_N; j++) res[i + j - POLY_N] -= di * b[j];
    }
}

/* mode 4: mirror of blind_rotate_mb2_kernel -- two key bits per external product:
 *   ACC <- ACC + [ K1 (X^e1 - 1) + K2 (X^e2 - 1) + K3 (X^(e1+e2) - 1) ] (.) ACC,
 * the bracket formed pointwise in the Fourier domain.  The point held by (lane L, register c) is the evaluation at
 * rho = w^(4 j + 1), w = exp(i pi / 2048), j = 64 rev4(c) + rev6(L); rho^e = mono[(4 rev6(L) + 1) e] * exp(i pi r e / 8),
 * r = rev4(c) mod 8, times (-1)^e for odd c -- factored exactly like the kernel does. */
static inline fcplx fcmac(fcplx w, double kr, double ki, fcplx acc) {   /* acc + w * k */
    fcplx t;
    t.r = fma(-w.i, ki, fma(w.r, kr, acc.r));
    t.i = fma(w.i, kr, fma(w.r, ki, acc.i));
    return t;
}
static void blind_rotate_mb2(const orc_server_key *k, const u32 *ms, const u64 *lut, u64 *acc /* [2][N] */) {
    static __thread double xx[POLY_N], xo[POLY_N];
    static __thread fcplx F[2][64][16], Tq[64][16];
    memset(acc, 0, POLY_N * sizeof(u64));
    poly_rotate(lut, (2 * POLY_N - ms[LWE_N]) & (2 * POLY_N - 1), acc + POLY_N);
    for (int p = 0; p < LWE_N / 2; p++) {
        const unsigned e1 = ms[2 * p], e2 = ms[2 * p + 1];
        if ((e1 | e2) == 0) continue;
        for (int c = 0; c < 2; c++) {
            for (int n = 0; n < POLY_N; n++) xx[n] = (double)pbs_digit(acc[c * POLY_N + n]);
            fmirror_forward(xx, F[c]);
        }
        for (int col = 0; col < 2; col++) {
            for (int L = 0; L < 64; L++) {
                const unsigned lane_root = 4u * brev_bits((unsigned)L, 6) + 1u;
                const unsigned ia = (lane_root * e1) & 4095u, ib = (lane_root * e2) & 4095u;
                const fcplx la = { fMono_re[ia], fMono_im[ia] }, lb = { fMono_re[ib], fMono_im[ib] };
                fcplx a = { 0, 0 }, b = { 0, 0 };
                for (int c = 0; c < 16; c++) {
                    if ((c & 1) == 0) {
                        const unsigned r3 = brev_bits((unsigned)(c >> 1), 3);
                        const unsigned ua = 256u * ((r3 * e1) & 15u), ub = 256u * ((r3 * e2) & 15u);
                        a = fcmul(la, fMono_re[ua], fMono_im[ua]);
                        b = fcmul(lb, fMono_re[ub], fMono_im[ub]);
                    } else {
                        if (e1 & 1) { a.r = -a.r; a.i = -a.i; }
                        if (e2 & 1) { b.r = -b.r; b.i = -b.i; }
                    }
                    fcplx a1 = a, b1 = b, ab1 = fcmul(a, b.r, b.i);
                    a1.r = a.r - 1.0; b1.r = b.r - 1.0; ab1.r = ab1.r - 1.0;
                    double rr = 0, ii = 0;
                    for (int h = 0; h < 2; h++) {            /* own row (= col) first, then the partner's */
                        const int row = h == 0 ? col : 1 - col;
                        const double *K[3];
                        for (int t = 0; t < 3; t++)
                            K[t] = k->bsk_mb + (((((size_t)p * 3 + t) * 2 + row) * 2 + col)) * 2 * FM + (size_t)(c * 64 + L) * 2;
                        fcplx y = { K[0][0], K[0][1] };
                        fcplx A = fcmul(y, a1.r, a1.i);
                        A = fcmac(b1, K[1][0], K[1][1], A);
                        A = fcmac(ab1, K[2][0], K[2][1], A);
                        const fcplx T = F[row][L][c];
                        if (h == 0) {
                            rr = T.r * A.r; rr = fma(-T.i, A.i, rr);
                            ii = T.r * A.i; ii = fma(T.i, A.r, ii);
                        } else {
                            rr = fma(T.r, A.r, rr); rr = fma(-T.i, A.i, rr);
                            ii = fma(T.r, A.i, ii); ii = fma(T.i, A.r, ii);
                        }
                    }
                    Tq[L][c].r = rr; Tq[L][c].i = ii;
                }
            }
            fmirror_inverse(Tq, xo);
            for (int n = 0; n < POLY_N; n++) acc[col * POLY_N + n] += fmirror_to_torus(xo[n]);
        }
    }
}

/* mode 5: the product's FHS_ARITH_EXACT_NTT_MB2 (fhestring_amd/csrc/nttmb_kernels.hip) restated with a different exact
 * algorithm.  The kernel forms K' = K1 (X^e1 - 1) + K2 (X^e2 - 1) + K3 (X^(e1+e2) - 1) pointwise in the domain of its
 * two-prime NTT; here K' is formed in the COEFFICIENT domain (rotate and subtract, wrapping u64 = exact, |K'| < 2^60) and
 * multiplied with the digits of ACC through the Goldilocks NTT on two key limbs like mode 0.  Both are exact integer
 * arithmetic mod 2^64, so the outputs must be equal bit for bit. */
static void blind_rotate_mb2_exact(const orc_server_key *k, const u32 *ms, const u64 *lut, u64 *acc /* [2][N] */) {
    u64 *dn = (u64 *)malloc(2 * POLY_N * sizeof(u64));
    u64 *kc = (u64 *)malloc(POLY_N * sizeof(u64));
    u64 *rt = (u64 *)malloc(POLY_N * sizeof(u64));
    u64 *lim = (u64 *)malloc(2 * POLY_N * sizeof(u64));
    u64 *tt = (u64 *)malloc(2 * POLY_N * sizeof(u64));
    const u64 lm = (1ull << 29) - 1;
    memset(acc, 0, POLY_N * sizeof(u64));
    poly_rotate(lut, (2 * POLY_N - ms[LWE_N]) & (2 * POLY_N - 1), acc + POLY_N);
    for (int p = 0; p < LWE_N / 2; p++) {
        const unsigned e[3] = { ms[2 * p], ms[2 * p + 1], (ms[2 * p] + ms[2 * p + 1]) & (2 * POLY_N - 1) };
        if ((e[0] | e[1]) == 0) continue;
        for (int c = 0; c < 2; c++) {
            for (int n = 0; n < POLY_N; n++) {
                i64 d = pbs_digit(acc[c * POLY_N + n]);
                dn[c * POLY_N + n] = d < 0 ? GP - (u64)(-d) : (u64)d;
            }
            g_ntt_fwd(dn + c * POLY_N);
        }
        for (int col = 0; col < 2; col++) {
            memset(tt, 0, 2 * POLY_N * sizeof(u64));
            for (int row = 0; row < 2; row++) {
                memset(kc, 0, POLY_N * sizeof(u64));
                for (int t = 0; t < 3; t++) {
                    const u64 *K = k->bsk_mb_q7 + ((((size_t)p * 3 + t) * 2 + row) * 2 + col) * POLY_N;
                    poly_rotate(K, e[t], rt);
                    for (int n = 0; n < POLY_N; n++) kc[n] += rt[n] - K[n];
                }
                for (int n = 0; n < POLY_N; n++) {           /* signed value / 2^7 -> limb 0 in [0, 2^29), signed limb 1 */
                    const i64 v = (i64)kc[n] >> 7;
                    const u64 l0 = (u64)v & lm;
                    const i64 l1 = (v - (i64)l0) >> 29;
                    lim[n] = l0;
                    lim[POLY_N + n] = l1 < 0 ? GP - (u64)(-l1) : (u64)l1;
                }
                g_ntt_fwd(lim);
                g_ntt_fwd(lim + POLY_N);
                for (int limb = 0; limb < 2; limb++)
                    for (int n = 0; n < POLY_N; n++)
                        tt[limb * POLY_N + n] = g_add(tt[limb * POLY_N + n], g_mul(dn[row * POLY_N + n], lim[limb * POLY_N + n]));
            }
            g_ntt_inv(tt);
            g_ntt_inv(tt + POLY_N);
            for (int n = 0; n < POLY_N; n++) {
                u64 r0 = tt[n], r1 = tt[POLY_N + n];
                u64 s0 = (r0 > GP / 2) ? r0 - GP : r0;
                u64 s1 = (r1 > GP / 2) ? r1 - GP : r1;
                acc[col * POLY_N + n] += (s0 + (s1 << 29)) << 7;
            }
        }
    }
    free(dn); free(kc); free(rt); free(lim); free(tt);
}

/* mode 0: Goldilocks NTT (2 x 29-bit key limbs, exact); mode 1: schoolbook (exact);
 * mode 2: f64 FFT (approximate, CPU-baseline only); mode 3: mirror of the GPU f64-FFT kernel */
static void blind_rotate(const orc_server_key *k, const u32 *ms, const u64 *lut,
                         u64 *acc /* [2][N] */, int mode) {
    if (mode == 4) { blind_rotate_mb2(k, ms, lut, acc); return; }
    if (mode == 5) { blind_rotate_mb2_exact(k, ms, lut, acc); return; }
    if (mode == 6) { blind_rotate_vec(k->bsk_vec, ms, lut, acc); return; }
    u64 *rot = (u64 *)malloc(POLY_N * sizeof(u64));
    i64 *dig = (i64 *)malloc(2 * POLY_N * sizeof(i64));
    u64 *dn = (u64 *)malloc(2 * POLY_N * sizeof(u64));
    u64 *tt = (u64 *)malloc(2 * POLY_N * sizeof(u64));
    u64 *res = (u64 *)malloc(2 * POLY_N * sizeof(u64));
    memset(acc, 0, POLY_N * sizeof(u64));
    poly_rotate(lut, (2 * POLY_N - ms[LWE_N]) & (2 * POLY_N - 1), acc + POLY_N);
    for (int i = 0; i < LWE_N; i++) {
        unsigned a = ms[i];
        if (a == 0) continue; /* X^0*acc - acc == 0: exact no-op */
        for (int c = 0; c < 2; c++) {
            poly_rotate(acc + c * POLY_N, a, rot);
            for (int n = 0; n < POLY_N; n++) dig[c * POLY_N + n] = pbs_digit(rot[n] - acc[c * POLY_N + n]);
        }
        if (mode == 3) {
            static __thread double xx[POLY_N], xo[POLY_N];
            static __thread fcplx F[2][64][16], Tq[64][16];
            for (int c = 0; c < 2; c++) {
                for (int n = 0; n < POLY_N; n++) xx[n] = (double)dig[c * POLY_N + n];
                fmirror_forward(xx, F[c]);
            }
            for (int col = 0; col < 2; col++) {
                /* wave `col` multiplies its own transform first (row = col), then the partner's */
                const double *bo = k->bsk_fm + ((((size_t)i * 2 + col) * 2 + col)) * 2 * FM;
                const double *bp = k->bsk_fm + ((((size_t)i * 2 + (1 - col)) * 2 + col)) * 2 * FM;
                for (int L = 0; L < 64; L++)
                    for (int c = 0; c < 16; c++) {
                        const double fr = F[col][L][c].r, fi = F[col][L][c].i, gr = F[1 - col][L][c].r, gi = F[1 - col][L][c].i;
                        const double b0r = bo[(c * 64 + L) * 2], b0i = bo[(c * 64 + L) * 2 + 1];
                        const double b1r = bp[(c * 64 + L) * 2], b1i = bp[(c * 64 + L) * 2 + 1];
                        double rr = fr * b0r; rr = fma(-fi, b0i, rr); rr = fma(gr, b1r, rr); rr = fma(-gi, b1i, rr);
                        double ii = fr * b0i; ii = fma(fi, b0r, ii); ii = fma(gr, b1i, ii); ii = fma(gi, b1r, ii);
                        Tq[L][c].r = rr; Tq[L][c].i = ii;
                    }
                fmirror_inverse(Tq, xo);
                for (int n = 0; n < POLY_N; n++) acc[col * POLY_N + n] += fmirror_to_torus(xo[n]);
            }
        } else if (mode == 2) {
            double dd[POLY_N], dre[2][FFT_N], dim[2][FFT_N], ore[FFT_N], oim[FFT_N], xo[POLY_N];
            for (int c = 0; c < 2; c++) {
                for (int n = 0; n < POLY_N; n++) dd[n] = (double)dig[c * POLY_N + n];
                fft_forward_poly(dd, dre[c], dim[c]);
            }
            for (int col = 0; col < 2; col++) {
                const double *b0 = k->bsk_fft + ((((size_t)i * 2 + 0) * 2 + col)) * 2 * FFT_N;
                const double *b1 = k->bsk_fft + ((((size_t)i * 2 + 1) * 2 + col)) * 2 * FFT_N;
                for (int q = 0; q < FFT_N; q++) {
                    ore[q] = dre[0][q] * b0[q] - dim[0][q] * b0[FFT_N + q] + dre[1][q] * b1[q] - dim[1][q] * b1[FFT_N + q];
                    oim[q] = dre[0][q] * b0[FFT_N + q] + dim[0][q] * b0[q] + dre[1][q] * b1[FFT_N + q] + dim[1][q] * b1[q];
                }
                fft_inverse_poly(ore, oim, xo);
                for (int n = 0; n < POLY_N; n++) acc[col * POLY_N + n] += f64_to_torus(xo[n]);
            }
        } else if (mode == 1) {
            memset(res, 0, 2 * POLY_N * sizeof(u64));
            for (int row = 0; row < 2; row++)
                for (int col = 0; col < 2; col++)
                    negacyclic_mac_schoolbook(dig + row * POLY_N,
                                              k->bsk + (((size_t)i * 2 + row) * 2 + col) * POLY_N,
                                              res + col * POLY_N);
            for (int n = 0; n < 2 * POLY_N; n++) acc[n] += res[n];
        } else {
            for (int c = 0; c < 2; c++) {
                for (int n = 0; n < POLY_N; n++) {
                    i64 d = dig[c * POLY_N + n];
                    dn[c * POLY_N + n] = d < 0 ? GP - (u64)(-d) : (u64)d;
                }
                g_ntt_fwd(dn + c * POLY_N);
            }
            for (int col = 0; col < 2; col++) {
                for (int limb = 0; limb < 2; limb++) {
                    const u64 *b0 = k->bsk_ntt + ((((size_t)i * 2 + 0) * 2 + col) * 2 + limb) * POLY_N;
                    const u64 *b1 = k->bsk_ntt + ((((size_t)i * 2 + 1) * 2 + col) * 2 + limb) * POLY_N;
                    u64 *t = tt + limb * POLY_N;
                    for (int n = 0; n < POLY_N; n++)
                        t[n] = g_add(g_mul(dn[n], b0[n]), g_mul(dn[POLY_N + n], b1[n]));
                    g_ntt_inv(t);
                }
                for (int n = 0; n < POLY_N; n++) {
                    /* lift to signed (|R| < 2^63 < p/2), recombine limbs mod 2^64 */
                    u64 r0 = tt[n], r1 = tt[POLY_N + n];
                    u64 s0 = (r0 > GP / 2) ? r0 - GP : r0; /* wraps to the signed value mod 2^64 */
                    u64 s1 = (r1 > GP / 2) ? r1 - GP : r1;
                    acc[col * POLY_N + n] += (s0 + (s1 << 29)) << BSK_QUANT_BITS;
                }
            }
        }
    }
    free(rot); free(dig); free(dn); free(tt); free(res);
}

static void ensure_mode(const orc_server_key *kc, int mode) {
    orc_server_key *k = (orc_server_key *)kc;
    if ((mode == 4 || mode == 5) && !k->bsk_mb) {
        fprintf(stderr, "oracle: modes 4 and 5 need the pair key (orc_server_key_set_mb2)\n");
        abort();
    }
    if (mode == 6) {
        vec_init();
        pthread_mutex_lock(&g_tab_mu);
        const int need6 = k->bsk_vec == 0;
        pthread_mutex_unlock(&g_tab_mu);
        if (need6) {
            double *m = (double *)aligned_alloc(32, BSK_POLYS * 2 * FM * sizeof(double));
            for (size_t p = 0; p < BSK_POLYS; p++) vec_bsk_poly(k->bsk + p * POLY_N, m + p * 2 * FM);
            pthread_mutex_lock(&g_tab_mu);
            if (!k->bsk_vec) k->bsk_vec = m; else free(m);
            pthread_mutex_unlock(&g_tab_mu);
        }
        return;
    }
    if (mode != 3) return;
    pthread_mutex_lock(&g_tab_mu);
    const int need = k->bsk_fm == 0;
    pthread_mutex_unlock(&g_tab_mu);
    if (need) {
        double *m = (double *)malloc(BSK_POLYS * 2 * FM * sizeof(double));
        orc_fft_bsk_convert(k->bsk, m);
        pthread_mutex_lock(&g_tab_mu);
        if (!k->bsk_fm) k->bsk_fm = m; else free(m);
        pthread_mutex_unlock(&g_tab_mu);
    }
}

/* one PBS: in big LWE [2049], lut [2048] -> out big LWE [2049] */
void orc_pbs(const orc_server_key *k, const u64 *in, const u64 *lut, u64 *out, int mode) {
    ensure_mode(k, mode);
    u32 ms[SMALL_CT];
    orc_keyswitch_modswitch(k, in, ms);
    u64 *acc = (u64 *)malloc(2 * POLY_N * sizeof(u64));
    blind_rotate(k, ms, lut, acc, mode);
    /* sample extract, coefficient 0 */
    out[0] = acc[0];
    for (int j = 1; j < POLY_N; j++) out[j] = (u64)0 - acc[POLY_N - j];
    out[BIG_N] = acc[POLY_N];
    free(acc);
}
/* ONE blind rotation, several sample extractions (the product's shift sharing, fhestring_amd/csrc/engine.cpp
 * "rotation sharing"): out[s] is what a bootstrap of (in + shift[s] * Delta) with the same look-up table yields, shift in
 * message units 0..31 -- adding c * Delta = c * 2^59 to the body moves the modulus-switched body by exactly 128 c
 * (2^59 * 4096 / 2^64), i.e. rotates the accumulator by X^(128 c), so that bootstrap's constant coefficient is
 * coefficient K = 128 c (negacyclic index in [0, 4096)) of THIS accumulator: extract coefficient K mod 2048, negate the
 * whole LWE when K >= 2048.  shift 0 is orc_pbs itself. */
void orc_pbs_shifted(const orc_server_key *k, const u64 *in, const u64 *lut, const u32 *shifts, u64 n, u64 *out, int mode) {
    ensure_mode(k, mode);
    u32 ms[SMALL_CT];
    orc_keyswitch_modswitch(k, in, ms);
    u64 *acc = (u64 *)malloc(2 * POLY_N * sizeof(u64));
    blind_rotate(k, ms, lut, acc, mode);
    for (u64 s = 0; s < n; s++) {
        const u32 K = (128u * (shifts[s] & 31u)) & (2 * POLY_N - 1);
        const u32 h = K & (POLY_N - 1);
        const u64 neg = K >= POLY_N ? ~(u64)0 : 0;          /* two's complement negate: (x ^ neg) - neg */
        u64 *o = out + s * BIG_CT;
        for (u32 i = 0; i < POLY_N; i++) {
            const u64 v = i <= h ? acc[h - i] : (u64)0 - acc[POLY_N + h - i];
            o[i] = (v ^ neg) - neg;
        }
        o[BIG_N] = (acc[POLY_N + h] ^ neg) - neg;
    }
    free(acc);
}
/* blind rotation only, from given mod-switched values (kernel-level tests) */
void orc_blind_rotate(const orc_server_key *k, const u32 *ms, const u64 *lut, u64 *acc, int mode) {
    ensure_mode(k, mode);
    blind_rotate(k, ms, lut, acc, mode);
}

/* ---- batched PBS over host threads (also the cpu_baseline leg) ----------- */
typedef struct {
    const orc_server_key *k;
    const u64 *in; const u32 *lut_idx; const u64 *luts; u64 *out;
    u64 B; int mode; volatile u64 *next;
} pbs_job;
static void *pbs_worker(void *p) {
    pbs_job *j = (pbs_job *)p;
    for (;;) {
        u64 b = __atomic_fetch_add(j->next, 1, __ATOMIC_RELAXED);
        if (b >= j->B) break;
        orc_pbs(j->k, j->in + b * BIG_CT, j->luts + (size_t)j->lut_idx[b] * POLY_N,
                j->out + b * BIG_CT, j->mode);
    }
    return 0;
}
void orc_pbs_batch(const orc_server_key *k, const u64 *in, const u32 *lut_idx, const u64 *luts,
                   u64 *out, u64 B, int nthreads, int mode) {
    ensure_mode(k, mode);
    volatile u64 next = 0;
    pbs_job job = { k, in, lut_idx, luts, out, B, mode, &next };
    if (nthreads < 1) nthreads = 1;
    if ((u64)nthreads > B) nthreads = (int)B;
    if (nthreads <= 1) { pbs_worker(&job); return; }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], 0, pbs_worker, &job);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], 0);
    free(th);
}

/* exact negacyclic product helpers exported for NTT-vs-schoolbook tests */
void orc_negacyclic_schoolbook(const i64 *d, const u64 *b, u64 *res) {
    memset(res, 0, POLY_N * sizeof(u64));
    negacyclic_mac_schoolbook(d, b, res);
}
void orc_negacyclic_ntt(const i64 *d, const u64 *b_quantised, u64 *res) {
    g_init_tables();
    u64 dn[POLY_N], l0[POLY_N], l1[POLY_N];
    const u64 lm = (1ull << 29) - 1;
    for (int n = 0; n < POLY_N; n++) {
        dn[n] = d[n] < 0 ? GP - (u64)(-d[n]) : (u64)d[n];
        u64 v = b_quantised[n] >> BSK_QUANT_BITS;
        l0[n] = v & lm; l1[n] = v >> 29;
    }
    g_ntt_fwd(dn); g_ntt_fwd(l0); g_ntt_fwd(l1);
    for (int n = 0; n < POLY_N; n++) { l0[n] = g_mul(l0[n], dn[n]); l1[n] = g_mul(l1[n], dn[n]); }
    g_ntt_inv(l0); g_ntt_inv(l1);
    for (int n = 0; n < POLY_N; n++) {
        u64 s0 = (l0[n] > GP / 2) ? l0[n] - GP : l0[n];
        u64 s1 = (l1[n] > GP / 2) ? l1[n] - GP : l1[n];
        res[n] = (s0 + (s1 << 29)) << BSK_QUANT_BITS;
    }
}
