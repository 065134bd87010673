// Hand-written HIP kernels for gfx950 (MI355X): batched TFHE programmable bootstrap.
//
// Blind rotation = 742 sequential CMUXes, each a GGSW x GLWE external product over
// Z_{2^64}[X]/(X^2048+1).  The negacyclic products are computed
// EXACTLY with a number-theoretic transform over two 47-bit primes whose residues are carried
// in FP64 registers: a*b mod p = fma-split product + one rounded quotient (6 VALU ops, no carry
// chains), add/sub are single v_add_f64 with lazy ranges.  On gfx950 v_fma_f64 issues at the
// same rate as v_mad_u64_u32 / v_addc_co_u32 (tools/ubench_valu.hip), so this needs ~3x fewer
// issue cycles than a 64-bit integer (Goldilocks) NTT.  No MFMA: this is torus arithmetic.
//
// Mapping: one workgroup (4 wavefronts) per ciphertext; wavefront (j,q) owns GLWE polynomial j
// modulo prime q.  A 2048-point NTT lives in one wavefront's registers (32 coefficients/lane):
// 5 radix-2 stages in the "strided" layout (lane = n mod 64), one transpose through LDS with the
// only cross-lane stage (t=32) fused into the transposed read, 5 stages in the "contiguous"
// layout (lane = n div 32).  The accumulator stays in registers for all 742 iterations.
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

// ------------------------------------------------------------------------------------------
// blind rotation + sample extract: grid = B workgroups, block = 256 threads (4 wavefronts)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void blind_rotate_kernel(BlindRotateParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ct = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = wave >> 1;   // GLWE polynomial (0 mask, 1 body)
    const int q = wave & 1;    // prime
    double *my = reinterpret_cast<double *>(smem) + wave * LDS_WAVE_SLOTS;
    const double *partner = reinterpret_cast<double *>(smem) + (wave ^ 2) * LDS_WAVE_SLOTS;  // other poly
    const double *sibling = reinterpret_cast<double *>(smem) + (wave ^ 1) * LDS_WAVE_SLOTS;  // other prime
    uint64_t *my_u = reinterpret_cast<uint64_t *>(my);

    const double p = q ? (double)NTT_P1 : (double)NTT_P0;
    const double pinv = 1.0 / p;
    const double p1 = (double)NTT_P1, p1inv = 1.0 / p1;
    const double crt_c = C_CRT;                       // p0^-1 mod p1, centred

    const uint64_t *ks = P.ks + (size_t)ct * SMALL_CT;   // keyswitched small LWE, mod-switched on use
    const double *fwd_lane = P.tw.fwd_lane + q * 32 * 64;
    const double *inv_lane = P.tw.inv_lane + q * 32 * 64;
    const double twA = lane < 32 ? C_FWD_UNI[q][lane] : C_INV_UNI[q][lane];
    const double twB = C_INV_UNI[q][lane & 31];

    // The accumulator of polynomial j is shared by the two prime-waves (j,0), (j,1): wave (j,q) OWNS
    // the coefficients n = lane + 64 r with r = 2 o + q (o < 16).  It computes their digits and their
    // CRT; digits and residues of the other half come from the sibling through LDS.  The natural-order
    // image S_j used for the rotated reads is distributed the same way: coefficient n lives in the
    // buffer of wave (j, r & 1) at slot n.  Slots of the opposite parity in a buffer carry what the
    // sibling publishes (digits, then residues), so the four uses never overlap.
    uint64_t acc[16];
    double *sib_w = reinterpret_cast<double *>(smem) + (wave ^ 1) * LDS_WAVE_SLOTS;
    const uint64_t *s_base = reinterpret_cast<const uint64_t *>(smem) + (size_t)(wave & ~1) * LDS_WAVE_SLOTS;
    {
        const uint32_t b = mod_switch(ks[LWE_N]);
        const uint32_t a = (2 * POLY_N - b) & (2 * POLY_N - 1);     // X^{-b} = X^{2N-b}
        const uint32_t s = a & (POLY_N - 1);
        const bool neg = a >= POLY_N;
        const uint64_t *lut = P.luts + (size_t)P.lut_idx[ct] * POLY_N;
#pragma unroll
        for (int o = 0; o < 16; o++) {
            const uint32_t n = lane + 64 * (2 * o + q);
            uint64_t v = 0;
            if (j == 1) {
                v = (n >= s) ? lut[n - s] : (uint64_t)0 - lut[n - s + POLY_N];
                if (neg) v = (uint64_t)0 - v;
            }
            acc[o] = v;
            my_u[n] = v;
        }
    }
    __syncthreads();

    for (int i = 0; i < LWE_N; i++) {
        const uint32_t a = mod_switch(ks[i]);
        if (a == 0) continue;   // X^0*acc - acc == 0: exact no-op (uniform for the workgroup)
        const uint32_t s = a & (POLY_N - 1);
        const bool neg = a >= POLY_N;

        // ---- rotate, subtract, decompose (closest multiple of 2^41 -> signed 23-bit digit) ----
        double x[32];
        // wave priority: the short phases that end at a workgroup barrier run at 2 (three other waves are parked
        // until this one arrives), the long barrier-free transforms at 0 / 1 (measured -2..-4 % kernel time)
        __builtin_amdgcn_s_setprio(2);
        if (q == 0) phase_digits<0>(x, acc, s_base, sib_w, lane, s, neg);
        else phase_digits<1>(x, acc, s_base, sib_w, lane, s, neg);
        __syncthreads();
        if (q == 0) phase_other_digits<0>(x, my, lane);
        else phase_other_digits<1>(x, my, lane);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_setprio(0);

        ntt_forward(x, my, lane, twA, fwd_lane, p, pinv);
        __builtin_amdgcn_s_setprio(2);

        // ---- publish, pointwise multiply-accumulate with GGSW_i ----
        // key layout [32/2][64 lanes][2]: one 16-byte load per lane covers coefficients (c, c+1)
        typedef double __attribute__((ext_vector_type(2))) double2_t;
        const double2_t *b_own = reinterpret_cast<const double2_t *>(
            P.bsk_ntt + ((((size_t)i * 2 + j) * 2 + j) * 2 + q) * POLY_N) + lane;
        const double2_t *b_par = reinterpret_cast<const double2_t *>(
            P.bsk_ntt + ((((size_t)i * 2 + (1 - j)) * 2 + j) * 2 + q) * POLY_N) + lane;
        constexpr int CH = 4;   // pairs per chunk -> 8 coefficients
        double2_t bo[CH], bp[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) { bo[k] = b_own[k * 64]; bp[k] = b_par[k * 64]; }   // in flight over the barrier
#pragma unroll
        for (int c = 0; c < 32; c++) my[c * 64 + lane] = x[c];
        __syncthreads();
#pragma unroll
        for (int ch = 0; ch < 16 / CH; ch++) {
            double2_t no[CH], np[CH];
            if (ch + 1 < 16 / CH) {
#pragma unroll
                for (int k = 0; k < CH; k++) {
                    no[k] = b_own[((ch + 1) * CH + k) * 64];
                    np[k] = b_par[((ch + 1) * CH + k) * 64];
                }
            }
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int c = 2 * (ch * CH + k);
                const double o0 = partner[c * 64 + lane], o1 = partner[(c + 1) * 64 + lane];
                x[c] = mulmod(x[c], bo[k].x, p, pinv) + mulmod(o0, bp[k].x, p, pinv);
                x[c + 1] = mulmod(x[c + 1], bo[k].y, p, pinv) + mulmod(o1, bp[k].y, p, pinv);
            }
            if (ch + 1 < 16 / CH) {
#pragma unroll
                for (int k = 0; k < CH; k++) { bo[k] = no[k]; bp[k] = np[k]; }
            }
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(1);

        ntt_inverse(x, my, lane, twA, twB, inv_lane, p, pinv);
        __builtin_amdgcn_s_setprio(2);

        // ---- exchange residues, CRT for the owned half, restage the accumulator ----
        if (q == 0) phase_publish_residues<0>(x, my, lane);
        else phase_publish_residues<1>(x, my, lane);
        __syncthreads();
        if (q == 0) phase_crt<0>(x, acc, sibling, my_u, lane, crt_c, p1, p1inv);
        else phase_crt<1>(x, acc, sibling, my_u, lane, crt_c, p1, p1inv);
        __syncthreads();
    }

    // ---- sample extract (coefficient 0): a'[0] = A[0], a'[n] = -A[N-n], b' = B[0] ----
    uint64_t *out = P.out_ptrs ? P.out_ptrs[ct] : P.out + (size_t)ct * BIG_CT;
    if (j == 0) {
#pragma unroll
        for (int o = 0; o < 16; o++) {
            const int n = lane + 64 * (2 * o + q);
            if (n == 0) out[0] = acc[o];
            else out[POLY_N - n] = (uint64_t)0 - acc[o];
        }
    } else if (q == 0 && lane == 0) {
        out[BIG_N] = acc[0];
    }
}

hipError_t read_device_ntt_consts(double *fwd_uni /*[64]*/, double *inv_uni /*[128]*/, double *crt) {
    hipError_t e = hipMemcpyFromSymbol(fwd_uni, HIP_SYMBOL(C_FWD_UNI), sizeof(double) * 64);
    if (e != hipSuccess) return e;
    e = hipMemcpyFromSymbol(inv_uni, HIP_SYMBOL(C_INV_UNI), sizeof(double) * 128);
    *crt = C_CRT;
    return e;
}

size_t blind_rotate_lds_bytes() { return (size_t)4 * LDS_WAVE_SLOTS * sizeof(double); }

// The exact-NTT kernel asks for more than 64 KB of dynamic LDS: the opt-in is a per-DEVICE function attribute, so it
// is set for the current device by Context::init (every context, any thread) and not cached per process.
hipError_t prepare_device_for_kernels() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(blind_rotate_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)blind_rotate_lds_bytes());
}

hipError_t launch_blind_rotate(const BlindRotateParams &p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(blind_rotate_kernel, dim3(p.B), dim3(256), blind_rotate_lds_bytes(), s, p);
    return hipGetLastError();
}

// keyswitch: ks_kernels.hip (matrix cores).  The modulus switch to Z_4096 happens where the value is consumed
// (blind-rotation kernels); modswitch_kernel only serves fhs_keyswitch_modswitch_batch.
__global__ __launch_bounds__(256) void modswitch_kernel(const uint64_t *__restrict__ ks, uint32_t *__restrict__ ms,
                                                        size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) ms[i] = mod_switch(ks[i]);
}
hipError_t launch_modswitch(const uint64_t *d_ks, uint32_t *d_ms, int B, hipStream_t s) {
    const size_t n = (size_t)B * SMALL_CT;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(modswitch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_ks, d_ms, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// linear layer: out[k] = sum_t coef*src + const (u64 wrapping); scatter of a dense level into blocks
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lincomb_kernel(const LinDesc *__restrict__ desc,
                                                      const LinTerm *__restrict__ terms,
                                                      uint64_t *__restrict__ out) {
    const LinDesc d = desc[blockIdx.x];
    uint64_t *o = out + (size_t)blockIdx.x * BIG_CT;
    for (int e = threadIdx.x; e < BIG_CT; e += 256) {
        uint64_t v = (e == BIG_N) ? d.konst_body : 0;
        for (uint32_t t = 0; t < d.n_terms; t++) {
            const LinTerm tm = terms[d.first_term + t];
            v += (uint64_t)tm.coef * tm.src[e];
        }
        o[e] = v;
    }
}
hipError_t launch_lincomb(const LinDesc *d_desc, const LinTerm *d_terms, uint64_t *d_out, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(lincomb_kernel, dim3(n), dim3(256), 0, s, d_desc, d_terms, d_out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void scatter_blocks_kernel(const uint64_t *__restrict__ in,
                                                             uint64_t *const *__restrict__ dst) {
    uint64_t *o = dst[blockIdx.x];
    const uint64_t *s = in + (size_t)blockIdx.x * BIG_CT;
    for (int e = threadIdx.x; e < BIG_CT; e += 256) o[e] = s[e];
}
hipError_t launch_scatter_blocks(const uint64_t *d_in, uint64_t *const *d_dst, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_blocks_kernel, dim3(n), dim3(256), 0, s, d_in, d_dst);
    return hipGetLastError();
}

}  // namespace fhs
