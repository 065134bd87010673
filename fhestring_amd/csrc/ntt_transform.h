// Device side of the exact two-prime NTT shared by pbs_kernels.hip (classic blind rotation) and nttmb_kernels.hip (two
// key bits per external product): FP64-carried modular arithmetic, the 2048-point transform of one wavefront (strided
// and contiguous register layouts, one LDS transpose), digit / residue exchange and CRT phases.  See the header
// comment of pbs_kernels.hip for the mapping.
#pragma once
#include "pbs_kernels.h"

namespace fhs {

#pragma clang fp contract(off)

#include "ntt_consts.inc"   // C_FWD_UNI, C_INV_UNI (constant address space -> scalar loads), C_CRT

// The ~96 lane-uniform twiddles of one prime live distributed over the 64 lanes of two resident
// registers (lane k holds constant k) and are broadcast with v_readlane: no memory traffic, no waits.
//   twA: lanes 0..31 = Psi[0..31],  lanes 32..63 = PsiInv[32..63]
//   twB: lanes 0..31 = PsiInv[0..31]
__device__ __forceinline__ double bcast_lane(double v, int k) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)b, k);
    const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), k);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
#define FWD_UNI(k) bcast_lane(twA, (k))
#define INV_UNI(k) ((k) >= 32 ? bcast_lane(twA, (k)) : bcast_lane(twB, (k)))

// ------------------------------------------------------------------------------------------
// exact modular arithmetic in FP64 (all values are integers with |x| < 2^52)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double mulmod(double a, double w, double p, double pinv) {
    double h = a * w;
    double l = __builtin_fma(a, w, -h);      // a*w == h + l exactly
    double q = __builtin_rint(h * pinv);
    double r = __builtin_fma(-q, p, h);      // exact: |h - q*p| < 2^52 and integral
    return r + l;                            // |result| <= p*(1/2 + |a|*2^-53)
}
__device__ __forceinline__ double reduce_once(double a, double p, double pinv) {
    double q = __builtin_rint(a * pinv);
    return __builtin_fma(-q, p, a);
}

// LDS layout used by the transposes: 8-byte slot of coefficient n, padded by 2 slots per 32 so
// that both the strided (lane = n mod 64) and the contiguous (lane = n div 32) access patterns
// are bank-conflict free with ds_read_b128 / ds_write_b128.
__device__ __forceinline__ int pad_slot(int n) { return n + 2 * (n >> 5); }
constexpr int LDS_WAVE_SLOTS = POLY_N + 2 * (POLY_N / 32);   // 2176 doubles = 17 408 B per wave

// ---- forward negacyclic NTT (Cooley-Tukey, merged psi powers, bit-reversed twiddle table) ----
// in : x[r] = coefficient (lane + 64 r)              (strided layout, natural order)
// out: x[c] = transform value at array index 32*lane + c (contiguous layout, CT output order)
// Per-lane twiddles factor as Psi[64G + G*lane + g] = Psi[64G + G*lane] * Psi[g] (disjoint bits
// under the bit reversal), so a lane keeps only 6 resident bases; everything else is lane-uniform.
__device__ __forceinline__ void ntt_forward(double (&x)[32], double *lds, int lane,
                                            const double twA,                 // lane-distributed uniform twiddles
                                            const double *__restrict__ lanetw, // [32][64] per-lane table of this prime
                                            double p, double pinv) {
    // per-lane bases Psi[32 + lane/2], Psi[64G + G*lane]: L1-resident table, loads overlap the first stages
    double base[6];
    base[0] = lanetw[lane];
#pragma unroll
    for (int k = 0; k < 5; k++) base[1 + k] = lanetw[(1 << k) * 64 + lane];
    // stages t = 1024..64: a 32-point CT on the register index, lane-uniform twiddles.  The last of them sends each
    // finished pair to LDS (transpose strided -> contiguous) while the next butterflies run: a burst of 32 stores
    // after the stage would stall the wave on the LDS store path; the scheduling barrier lets arithmetic cross
    // but pins the stores.  pad_slot(lane + 64 r) == (lane + 2*(lane>>5)) + 68 r: one address register.
    double *wr = lds + (lane + 2 * (lane >> 5));
#pragma unroll
    for (int T = 16; T >= 1; T >>= 1) {
        const int m = 16 / T;
#pragma unroll
        for (int i = 0; i < m; i++) {
            const double w = FWD_UNI(m + i);
#pragma unroll
            for (int r = 2 * i * T; r < 2 * i * T + T; r++) {
                double v = mulmod(x[r + T], w, p, pinv);
                x[r + T] = x[r] - v;
                x[r] = x[r] + v;
                if (T == 1) {
                    wr[68 * r] = x[r];
                    wr[68 * (r + 1)] = x[r + 1];
                    __builtin_amdgcn_sched_barrier(0x7);
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // stage t = 32 fused into the read: lanes (2k, 2k+1) share one 64-coefficient group
    {
        const double w = base[0];                          // Psi[32 + lane/2]
        const double sgn = (lane & 1) ? -1.0 : 1.0;
        const double *lo = lds + 34 * (lane & ~1);   // pad_slot(32 L + c) == 34 L + c
        const double *hi = lds + 34 * (lane | 1);
#pragma unroll
        for (int c = 0; c < 32; c++) {
            double v = mulmod(hi[c], w, p, pinv);
            x[c] = __builtin_fma(v, sgn, lo[c]);          // even lane: lo + v, odd lane: lo - v
        }
    }
    __builtin_amdgcn_wave_barrier();
    // stages t = 16..1: in-lane, twiddle = base[stage] * Psi[g]
    int lg = 0;
#pragma unroll
    for (int t = 16; t >= 1; t >>= 1, lg++) {
        const int G = 16 / t;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const double w = g == 0 ? base[1 + lg] : mulmod(base[1 + lg], FWD_UNI(g), p, pinv);
#pragma unroll
            for (int c = 2 * g * t; c < 2 * g * t + t; c++) {
                double v = mulmod(x[c + t], w, p, pinv);
                x[c + t] = x[c] - v;
                x[c] = x[c] + v;
            }
        }
    }
}

// ---- inverse negacyclic NTT (Gentleman-Sande), 1/N folded into the key ----
// in : x[c] at array index 32*lane + c (contiguous layout), |x| <= 1.5 p
// out: x[r] = coefficient (lane + 64 r) (strided layout), |x| <= 11.1 p, congruent mod p
__device__ __forceinline__ void ntt_inverse(double (&x)[32], double *lds, int lane,
                                            const double twA, const double twB, // lane-distributed uniform twiddles
                                            const double *__restrict__ lanetw,  // [32][64] per-lane table of this prime
                                            double p, double pinv) {
    double base[5];
#pragma unroll
    for (int k = 0; k < 5; k++) base[k] = lanetw[(1 << k) * 64 + lane];
    // Lazy ranges: magnitudes are tracked statically per register (in units of p: pointwise output
    // 1.3, a sum adds its operands, a mulmod/reduce output is 0.5 + input/60) and only the 13
    // registers that would push a mulmod input past 24 p are reset -- see DESIGN.md section 2.
    double *wr = lds + 34 * lane;
    int lg = 4;
#pragma unroll
    for (int t = 1; t <= 16; t <<= 1, lg--) {
        const int G = 16 / t;
        if (t == 16) x[0] = reduce_once(x[0], p, pinv);
#pragma unroll
        for (int g = 0; g < G; g++) {
            const double w = g == 0 ? base[lg] : mulmod(base[lg], INV_UNI(g), p, pinv);
#pragma unroll
            for (int c = 2 * g * t; c < 2 * g * t + t; c++) {
                double u = x[c], v = x[c + t];
                x[c] = u + v;
                x[c + t] = mulmod(u - v, w, p, pinv);
                if (t == 16) {   // last in-lane stage: transpose contiguous -> strided as the pairs complete
                    if (c < 4) x[c] = reduce_once(x[c], p, pinv);   // the only registers above 3 p
                    wr[c] = x[c];
                    wr[c + 16] = x[c + 16];
                    __builtin_amdgcn_sched_barrier(0x7);
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    {
        const bool upper = lane >= 32;
        const double *rd = lds + (lane & 31);      // pad_slot(l5 + 64 r) == l5 + 68 r, +34 for the partner
#pragma unroll
        for (int r = 0; r < 32; r++) {
            double a = rd[68 * r];
            double b = rd[68 * r + 34];
            double s = a + b;
            double d = mulmod(a - b, INV_UNI(32 + r), p, pinv);
            x[r] = upper ? d : s;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // stages t = 64..1024 on the register index, lane-uniform twiddles
#pragma unroll
    for (int T = 1; T <= 16; T <<= 1) {
        const int h = 16 / T;
        const unsigned reset = T == 4 ? 0x01010101u : (T >= 8 ? 0x00010001u : 0u);
#pragma unroll
        for (int r = 0; r < 32; r++)
            if ((reset >> r) & 1u) x[r] = reduce_once(x[r], p, pinv);
#pragma unroll
        for (int i = 0; i < h; i++) {
            const double w = INV_UNI(h + i);
#pragma unroll
            for (int r = 2 * i * T; r < 2 * i * T + T; r++) {
                double u = x[r], v = x[r + T];
                x[r] = u + v;
                x[r + T] = mulmod(u - v, w, p, pinv);
            }
        }
    }
}

__device__ __forceinline__ uint32_t mod_switch(uint64_t x) {      // round to Z_4096 (2N)
    return (uint32_t)(((x + (1ull << 51)) >> 52) & 4095u);
}

__device__ __forceinline__ int64_t f64_to_i64_exact(double v) {   // |v| < 2^51, integral
    const double M = 6755399441055744.0;                            // 1.5 * 2^52
    return (int64_t)(__builtin_bit_cast(uint64_t, v + M) - __builtin_bit_cast(uint64_t, M));
}


// ---- phases that depend on the ownership parity Q (compile-time, so x[] and acc[] stay in registers) ----
template <int Q>
__device__ __forceinline__ void phase_digits(double (&x)[32], const uint64_t (&acc)[16], const uint64_t *s_base,
                                             double *sib_w, int lane, uint32_t s, bool neg) {
#pragma unroll
    for (int o = 0; o < 16; o++) {
        constexpr int dummy = 0; (void)dummy;
        const int r = 2 * o + Q;
        const uint32_t n = lane + 64 * r;
        const uint32_t m = (n - s) & (POLY_N - 1);                 // source coefficient of X^s * acc
        uint64_t v = s_base[((m >> 6) & 1) * LDS_WAVE_SLOTS + m];
        if ((n < s) != neg) v = (uint64_t)0 - v;
        const uint64_t d = v - acc[o];
        const int32_t dig = (int32_t)((uint32_t)(d >> 32) + 0x100u) >> 9;
        const double dg = (double)dig;                             // identical for both primes (|dig| < p)
        x[r] = dg;
        sib_w[n] = dg;
    }
}
template <int Q>
__device__ __forceinline__ void phase_other_digits(double (&x)[32], const double *my, int lane) {
#pragma unroll
    for (int o = 0; o < 16; o++) x[2 * o + (1 - Q)] = my[lane + 64 * (2 * o + (1 - Q))];
}
template <int Q>
__device__ __forceinline__ void phase_publish_residues(const double (&x)[32], double *my, int lane) {
#pragma unroll
    for (int o = 0; o < 16; o++) my[lane + 64 * (2 * o + (1 - Q))] = x[2 * o + (1 - Q)];
}
template <int Q>
__device__ __forceinline__ void phase_crt(const double (&x)[32], uint64_t (&acc)[16], const double *sibling,
                                          uint64_t *my_u, int lane, double crt_c, double p1, double p1inv) {
#pragma unroll
    for (int o = 0; o < 16; o++) {
        const int r = 2 * o + Q;
        const double other = sibling[lane + 64 * r];
        const double r0 = Q ? other : x[r];
        const double r1 = Q ? x[r] : other;
        const double t = mulmod(r1 - r0, crt_c, p1, p1inv);
        const uint64_t v = (uint64_t)f64_to_i64_exact(r0) + NTT_P0 * (uint64_t)f64_to_i64_exact(t);
        acc[o] += v << BSK_QUANT_BITS;
        my_u[lane + 64 * r] = acc[o];
    }
}

}  // namespace fhs
