#include "ntt_tables.h"

#include <thread>

#include "pbs_kernels.h"

namespace fhs {
namespace {
typedef unsigned __int128 u128;

inline uint64_t mulm(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)((u128)a * b % p); }
uint64_t powm(uint64_t b, uint64_t e, uint64_t p) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = mulm(r, b, p);
        b = mulm(b, b, p);
        e >>= 1;
    }
    return r;
}
inline unsigned bitrev11(unsigned x) {
    unsigned r = 0;
    for (int i = 0; i < 11; i++) r |= ((x >> i) & 1u) << (10 - i);
    return r;
}
inline double centred(uint64_t v, uint64_t p) { return v > p / 2 ? -(double)(p - v) : (double)v; }

struct PrimeTables {
    uint64_t p;
    std::vector<uint64_t> psi_br, ipsi_br;   // Psi[k] = psi^bitrev(k), PsiInv[k] = psi^-bitrev(k)
    uint64_t ninv;
};
PrimeTables make_prime(uint64_t p, uint64_t psi) {
    PrimeTables t;
    t.p = p;
    t.psi_br.resize(POLY_N);
    t.ipsi_br.resize(POLY_N);
    const uint64_t ipsi = powm(psi, p - 2, p);
    uint64_t a = 1, b = 1;
    for (unsigned i = 0; i < (unsigned)POLY_N; i++) {
        t.psi_br[bitrev11(i)] = a;
        t.ipsi_br[bitrev11(i)] = b;
        a = mulm(a, psi, p);
        b = mulm(b, ipsi, p);
    }
    t.ninv = powm(POLY_N, p - 2, p);
    return t;
}
const PrimeTables &prime(int q) {
    static const PrimeTables t0 = make_prime(NTT_P0, NTT_PSI0);
    static const PrimeTables t1 = make_prime(NTT_P1, NTT_PSI1);
    return q ? t1 : t0;
}
// same Cooley-Tukey flow as the device kernel (natural order in, CT order out)
void ntt_forward_exact(uint64_t *a, const PrimeTables &t) {
    const uint64_t p = t.p;
    unsigned len = POLY_N;
    for (unsigned m = 1; m < (unsigned)POLY_N; m <<= 1) {
        len >>= 1;
        for (unsigned i = 0; i < m; i++) {
            const uint64_t w = t.psi_br[m + i];
            uint64_t *x = a + 2 * i * len, *y = x + len;
            for (unsigned k = 0; k < len; k++) {
                const uint64_t u = x[k], v = mulm(y[k], w, p);
                x[k] = u + v >= p ? u + v - p : u + v;
                y[k] = u >= v ? u - v : u + p - v;
            }
        }
    }
}
}  // namespace

void build_ntt_tables(HostNttTables &t) {
    t.fwd_uni.assign(2 * 32, 0.0);
    t.fwd_lane.assign(2 * 32 * 64, 0.0);
    t.inv_uni.assign(2 * 64, 0.0);
    t.inv_lane.assign(2 * 32 * 64, 0.0);
    for (int q = 0; q < 2; q++) {
        const PrimeTables &pt = prime(q);
        for (int k = 0; k < 32; k++) t.fwd_uni[q * 32 + k] = centred(pt.psi_br[k], pt.p);
        for (int k = 0; k < 64; k++) t.inv_uni[q * 64 + k] = centred(pt.ipsi_br[k], pt.p);
        for (int lane = 0; lane < 64; lane++) {
            t.fwd_lane[(q * 32 + 0) * 64 + lane] = centred(pt.psi_br[32 + lane / 2], pt.p);
            for (int e = 1; e < 32; e++) {
                int G = 1;
                while (2 * G <= e) G *= 2;
                const int g = e - G;
                const int idx = 64 * G + G * lane + g;
                t.fwd_lane[(q * 32 + e) * 64 + lane] = centred(pt.psi_br[idx], pt.p);
                t.inv_lane[(q * 32 + e) * 64 + lane] = centred(pt.ipsi_br[idx], pt.p);
            }
        }
    }
    t.crt_c = centred(powm(NTT_P0 % NTT_P1, NTT_P1 - 2, NTT_P1), NTT_P1);
    t.mono.assign(2 * 4096, 0.0);
    const uint64_t psis[2] = {NTT_PSI0, NTT_PSI1};
    for (int q = 0; q < 2; q++) {
        const PrimeTables &pt = prime(q);
        uint64_t a = 1;
        for (int k = 0; k < 4096; k++) {
            t.mono[q * 4096 + k] = centred(a, pt.p);
            a = mulm(a, psis[q], pt.p);
        }
    }
}

// The slot at array index idx of the device's forward transform holds the evaluation at psi^(2 bitrev11(idx) + 1):
// checked once per process on the monomial X (Context::load_multibit_key refuses to run otherwise).
bool ntt_slot_roots_are_bitreversed() {
    for (int q = 0; q < 2; q++) {
        const PrimeTables &pt = prime(q);
        std::vector<uint64_t> a(POLY_N, 0);
        a[1] = 1;
        ntt_forward_exact(a.data(), pt);
        const uint64_t psi = q ? NTT_PSI1 : NTT_PSI0;
        for (unsigned idx = 0; idx < (unsigned)POLY_N; idx++)
            if (a[idx] != powm(psi, 2 * bitrev11(idx) + 1, pt.p)) return false;
    }
    return true;
}

void convert_bsk_to_ntt(const uint64_t *bsk_std, double *out, int nthreads, int n_ggsw, int quant_bits) {
    const size_t n_polys = (size_t)n_ggsw * 4;
    if (nthreads < 1) nthreads = 1;
    auto work = [&](int tid) {
        std::vector<uint64_t> a(POLY_N);
        for (size_t pi = tid; pi < n_polys; pi += nthreads) {
            const uint64_t *src = bsk_std + pi * POLY_N;
            for (int q = 0; q < 2; q++) {
                const PrimeTables &pt = prime(q);
                for (int n = 0; n < POLY_N; n++) {
                    // round to the 58-bit grid, signed representative of the torus element / 2^6
                    const uint64_t r = (src[n] + (1ull << (quant_bits - 1))) & ~((1ull << quant_bits) - 1);
                    const int64_t v = (int64_t)r >> quant_bits;
                    const int64_t m = v % (int64_t)pt.p;
                    a[n] = (uint64_t)(m < 0 ? m + (int64_t)pt.p : m);
                }
                ntt_forward_exact(a.data(), pt);
                double *dst = out + (pi * 2 + q) * POLY_N;
                for (int idx = 0; idx < POLY_N; idx++) {
                    const int lane = idx >> 5, c = idx & 31;
                    dst[((c >> 1) * 64 + lane) * 2 + (c & 1)] = centred(mulm(a[idx], pt.ninv, pt.p), pt.p);
                }
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
}

}  // namespace fhs
