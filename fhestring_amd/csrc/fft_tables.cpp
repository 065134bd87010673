// Twiddle tables of the f64-FFT arithmetic mode, derived with libm on the host.  The lane-uniform ones are
// baked into the kernel as literals (fft_consts.inc); Context::load_server_key compares both and refuses
// to run on a mismatch.  The bootstrapping key itself is transformed on the device (bsk_to_fft_kernel).
#include "fft_tables.h"

#include <cmath>

namespace fhs {

#pragma clang fp contract(off)

namespace {
constexpr int FM = 1024;
unsigned brev(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}
}  // namespace

void build_fft_tables(HostFftTables &t) {
    const double PI = 3.14159265358979323846;
    t.w_re.assign(FM, 0.0); t.w_im.assign(FM, 0.0);
    int d = 0;
    for (unsigned m = 1; m < (unsigned)FM; m <<= 1, d++)
        for (unsigned i = 0; i < m; i++) {
            const double e = (double)((FM / (2 * m)) * (4 * brev(i, d) + 1));   // exact integer
            t.w_re[m + i] = std::cos(PI * e / 2048.0);
            t.w_im[m + i] = std::sin(PI * e / 2048.0);
        }
    t.u_re = {std::cos(PI * 1.0 / 4.0), std::cos(PI * 1.0 / 8.0), std::cos(PI * 3.0 / 8.0)};
    t.u_im = {std::sin(PI * 1.0 / 4.0), std::sin(PI * 1.0 / 8.0), std::sin(PI * 3.0 / 8.0)};
    t.lanetab.assign(12 * 64, 0.0);
    for (int L = 0; L < 64; L++) {
        const int hi = L >> 2;
        for (int lg = 0; lg < 4; lg++) {
            const int G = 1 << lg;
            t.lanetab[(2 * lg) * 64 + L] = t.w_re[16 * G + G * hi];
            t.lanetab[(2 * lg + 1) * 64 + L] = t.w_im[16 * G + G * hi];
        }
        t.lanetab[8 * 64 + L] = t.w_re[256 + 4 * L];
        t.lanetab[9 * 64 + L] = t.w_im[256 + 4 * L];
        t.lanetab[10 * 64 + L] = t.w_re[512 + 8 * L];
        t.lanetab[11 * 64 + L] = t.w_im[512 + 8 * L];
    }
    // called through volatile pointers: a compiler that fuses the pair into one sincos() call gets a libm routine
    // whose results differ from cos() / sin() in the last place for a few arguments (seen: gcc vs clang, 2 of 8192
    // entries); the CPU oracle builds its copy of this table the same way
    static double (*volatile p_cos)(double) = std::cos;
    static double (*volatile p_sin)(double) = std::sin;
    t.mono.assign(2 * 4096, 0.0);
    for (int k = 0; k < 4096; k++) {
        t.mono[2 * k] = p_cos(PI * (double)k / 2048.0);
        t.mono[2 * k + 1] = p_sin(PI * (double)k / 2048.0);
    }
    t.r16.assign(2 * 16, 0.0);
    for (int k = 0; k < 16; k++) {                        // the same table values, so that rho^e is consistent
        t.r16[2 * k] = t.mono[2 * (256 * k)];
        t.r16[2 * k + 1] = t.mono[2 * (256 * k) + 1];
    }
    // effective twiddles (see fft_tables.h)
    t.weff.assign(2 * FM, 0.0);
    auto put = [&](int idx, double re, double im) { t.weff[2 * idx] = re; t.weff[2 * idx + 1] = im; };
    auto cmul = [](double ar, double ai, double wr, double wi, double &tr, double &ti) {
        tr = std::fma(-ai, wi, ar * wr);
        ti = std::fma(ai, wr, ar * wi);
    };
    put(1, t.w_re[1], t.w_im[1]);
    for (int m = 2; m < FM; m <<= 1) {
        // group size G: lane-uniform stages (m < 16) pair (even, odd = rotated); in-lane stages of layout B
        // (16 <= m < 256) use G = m / 16, of layout C (m >= 256) G = m / 64
        const int G = m < 16 ? 2 : (m < 256 ? m / 16 : m / 64);
        for (int base = m; base < 2 * m; base += G) {
            double vr = 0, vi = 0;
            for (int g = 0; g < G; g++) {
                if (g == 0) { vr = t.w_re[base]; vi = t.w_im[base]; }
                else if (g & 1) { const double r = vr; vr = -vi; vi = r; }            // i * previous
                else {
                    const int u = (G == 8 && g == 4) ? 1 : (G == 8 && g == 6) ? 2 : 0;
                    cmul(t.w_re[base], t.w_im[base], t.u_re[u], t.u_im[u], vr, vi);
                }
                put(base + g, vr, vi);
            }
        }
    }
}

}  // namespace fhs
