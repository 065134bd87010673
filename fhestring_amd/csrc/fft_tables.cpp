// Tables and bootstrapping-key conversion of the f64-FFT arithmetic mode.  The forward transform here
// walks (lane, register) exactly like fft_kernels.hip::fft_forward with the same IEEE-754 operation
// order, so a key converted on the host is what the device itself would have produced.
#include "fft_tables.h"

#include <cmath>
#include <thread>

#include "pbs_kernels.h"

namespace fhs {

#pragma clang fp contract(off)

namespace {
constexpr int FM = 1024;
unsigned brev(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}
inline int fslot(int n) { return n + (n >> 6); }
struct C { double r, i; };
inline C cmul(C a, double wr, double wi) { return C{std::fma(-a.i, wi, a.r * wr), std::fma(a.i, wr, a.r * wi)}; }

const HostFftTables &tables() {
    static const HostFftTables t = [] { HostFftTables x; build_fft_tables(x); return x; }();
    return t;
}

// x[2048] -> F[lane][c], value at array index 16*lane + c
void forward_poly(const double *x, C (*F)[16]) {
    const HostFftTables &t = tables();
    std::vector<C> lds(FM + 16);
    for (int l = 0; l < 64; l++) {
        C z[16];
        for (int r = 0; r < 16; r++) z[r] = C{x[l + 64 * r], x[l + 64 * r + 1024]};
        for (int T = 8; T >= 1; T >>= 1) {
            const int m = 8 / T;
            for (int i = 0; i < m; i++) {
                const double wr = t.w_re[m + i], wi = t.w_im[m + i];
                for (int r = 2 * i * T; r < 2 * i * T + T; r++) {
                    const C v = cmul(z[r + T], wr, wi), u = z[r];
                    z[r] = C{u.r + v.r, u.i + v.i};
                    z[r + T] = C{u.r - v.r, u.i - v.i};
                }
            }
        }
        for (int r = 0; r < 16; r++) lds[fslot(l + 64 * r)] = z[r];
    }
    for (int L = 0; L < 64; L++) {
        const int q = L & 3, gL = L >> 2;
        const double war = t.lanetab[0 * 64 + L], wai = t.lanetab[1 * 64 + L];
        const double wbr = t.lanetab[2 * 64 + L], wbi = t.lanetab[3 * 64 + L];
        const double s1 = q < 2 ? 1.0 : -1.0, s2 = (q & 1) ? -1.0 : 1.0;
        C *y = F[L];
        for (int c = 0; c < 16; c++) {
            const int b = fslot(64 * gL + c);
            const C e0 = lds[b], e1 = lds[b + 16], e2 = lds[b + 32], e3 = lds[b + 48];
            const C t2 = cmul(e2, war, wai), t3 = cmul(e3, war, wai);
            const C A{std::fma(s1, t2.r, e0.r), std::fma(s1, t2.i, e0.i)};
            const C B{std::fma(s1, t3.r, e1.r), std::fma(s1, t3.i, e1.i)};
            const C tb = cmul(B, wbr, wbi);
            y[c] = C{std::fma(s2, tb.r, A.r), std::fma(s2, tb.i, A.i)};
        }
        int lg = 0;
        for (int tt = 8; tt >= 1; tt >>= 1, lg++) {
            const int G = 8 / tt;
            const C base{t.lanetab[(4 + 2 * lg) * 64 + L], t.lanetab[(5 + 2 * lg) * 64 + L]};
            for (int g = 0; g < G; g++) {
                C w = base;
                if (g) w = cmul(base, t.u_re[G + g], t.u_im[G + g]);
                for (int c = 2 * g * tt; c < 2 * g * tt + tt; c++) {
                    const C v = cmul(y[c + tt], w.r, w.i), u = y[c];
                    y[c] = C{u.r + v.r, u.i + v.i};
                    y[c + tt] = C{u.r - v.r, u.i - v.i};
                }
            }
        }
    }
}
}  // namespace

void build_fft_tables(HostFftTables &t) {
    const double PI = 3.14159265358979323846;
    t.w_re.assign(FM, 0.0); t.w_im.assign(FM, 0.0);
    t.u_re.assign(16, 0.0); t.u_im.assign(16, 0.0);
    int d = 0;
    for (unsigned m = 1; m < (unsigned)FM; m <<= 1, d++)
        for (unsigned i = 0; i < m; i++) {
            const double e = (double)((FM / (2 * m)) * (4 * brev(i, d) + 1));   // exact integer
            t.w_re[m + i] = std::cos(PI * e / 2048.0);
            t.w_im[m + i] = std::sin(PI * e / 2048.0);
        }
    for (int G = 2, lg = 1; G <= 8; G <<= 1, lg++)
        for (int g = 0; g < G; g++) {
            t.u_re[G + g] = std::cos(PI * (double)brev(g, lg) / (double)G);
            t.u_im[G + g] = std::sin(PI * (double)brev(g, lg) / (double)G);
        }
    t.lanetab.assign(12 * 64, 0.0);
    for (int L = 0; L < 64; L++) {
        const int q = L & 3, gL = L >> 2;
        t.lanetab[0 * 64 + L] = t.w_re[16 + gL];
        t.lanetab[1 * 64 + L] = t.w_im[16 + gL];
        t.lanetab[2 * 64 + L] = t.w_re[32 + 2 * gL + (q >> 1)];
        t.lanetab[3 * 64 + L] = t.w_im[32 + 2 * gL + (q >> 1)];
        for (int lg = 0; lg < 4; lg++) {
            const int G = 1 << lg;
            t.lanetab[(4 + 2 * lg) * 64 + L] = t.w_re[64 * G + G * L];
            t.lanetab[(5 + 2 * lg) * 64 + L] = t.w_im[64 * G + G * L];
        }
    }
}

void convert_bsk_to_fft(const uint64_t *bsk_std, double *out, int nthreads) {
    const size_t n_polys = (size_t)LWE_N * 4;
    if (nthreads < 1) nthreads = 1;
    (void)tables();
    auto work = [&](int tid) {
        std::vector<double> x(POLY_N);
        std::vector<C> Fv(64 * 16);
        C(*F)[16] = reinterpret_cast<C(*)[16]>(Fv.data());
        for (size_t pi = tid; pi < n_polys; pi += nthreads) {
            const uint64_t *src = bsk_std + pi * POLY_N;
            for (int n = 0; n < POLY_N; n++) {
                const uint64_t r = (src[n] + (1ull << (BSK_QUANT_BITS - 1))) & ~((1ull << BSK_QUANT_BITS) - 1);
                x[n] = (double)(int64_t)r;
            }
            forward_poly(x.data(), F);
            double *dst = out + pi * 2 * FM;
            for (int L = 0; L < 64; L++)
                for (int c = 0; c < 16; c++) {
                    dst[(c * 64 + L) * 2 + 0] = F[L][c].r * 0.0009765625;
                    dst[(c * 64 + L) * 2 + 1] = F[L][c].i * 0.0009765625;
                }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
}

}  // namespace fhs
