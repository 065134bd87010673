// Blind rotation with TWO key bits per external product in EXACT integer arithmetic (gfx950 only):
// FHS_ARITH_EXACT_NTT_MB2.  Same algebra as fftmb_kernels.hip,
//     ACC <- ACC + [ K1 (X^e1 - 1) + K2 (X^e2 - 1) + K3 (X^(e1+e2) - 1) ] (.) ACC          (371 products per bootstrap),
// on the machinery of pbs_kernels.hip: the negacyclic products are computed exactly with the NTT over two 47-bit
// primes carried in FP64 registers, one workgroup of 4 wavefronts per ciphertext, wavefront (j, q) = GLWE polynomial j
// modulo prime q.  The bracket is formed pointwise in the NTT domain: the slot at array index idx holds the evaluation
// at psi^(2 bitrev11(idx) + 1), so X^e becomes psi^((2 k' + 1) e) there, an exact field element:
//     (per-lane base: one table gather per monomial) x (a wave-uniform 32nd root of unity per register pair: scalar
//     loads), the odd register of a pair differs by (-1)^e.
// Everything is exact, so the result equals the integer computation mod 2^64 (mode 5 of the CPU oracle does it with a different
// algorithm: coefficient-domain combination of the keys + Goldilocks NTT) as long as the CRT range p0 p1 / 2 > 2^93
// covers the integer result: |digit| <= 2^22, 4096 terms, |combined key| <= 6 x 2^56 -- which is why the pair key is
// rounded to the 57-bit torus grid here (2^7 instead of the classic key's 2^6; the rounding adds 2^-59 to a key noise of
// 2^-51.6).  No f64 rounding noise at all in this arithmetic: the bootstrap output's sigma is 2^48.8.
#include "ntt_transform.h"

namespace fhs {

#pragma clang fp contract(off)

namespace {

constexpr int MB_QUANT_BITS = 7;

__device__ __forceinline__ double flip_sign(double v, uint32_t signmask) {     // exact: toggles the sign bit
    typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
    u32x2 w = __builtin_bit_cast(u32x2, v);
    w.y ^= signmask;
    return __builtin_bit_cast(double, w);
}
__device__ __forceinline__ int rev6(int x) { return (int)(__builtin_bitreverse32((uint32_t)x) >> 26); }

template <int Q>
__device__ __forceinline__ void mb_phase_digits(double (&x)[32], const uint64_t (&acc)[16], double *sib_w, int lane) {
#pragma unroll
    for (int o = 0; o < 16; o++) {
        const int r = 2 * o + Q;
        const int32_t dig = (int32_t)((uint32_t)(acc[o] >> 32) + 0x100u) >> 9;    // closest multiple of 2^41
        const double dg = (double)dig;
        x[r] = dg;
        sib_w[lane + 64 * r] = dg;
    }
}
template <int Q>
__device__ __forceinline__ void mb_phase_crt(const double (&x)[32], uint64_t (&acc)[16], const double *sibling, int lane,
                                             double crt_c, double p1, double p1inv) {
#pragma unroll
    for (int o = 0; o < 16; o++) {
        const int r = 2 * o + Q;
        const double other = sibling[lane + 64 * r];
        const double r0 = Q ? other : x[r];
        const double r1 = Q ? x[r] : other;
        const double t = mulmod(r1 - r0, crt_c, p1, p1inv);
        const uint64_t v = (uint64_t)f64_to_i64_exact(r0) + NTT_P0 * (uint64_t)f64_to_i64_exact(t);
        acc[o] += v << MB_QUANT_BITS;
    }
}

}  // namespace

template <int PF>
__global__ __launch_bounds__(256, 2) void blind_rotate_ntt_mb2_kernel(BlindRotateNttMb2Params P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ct = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = wave >> 1;   // GLWE polynomial (0 mask, 1 body)
    const int q = wave & 1;    // prime
    double *my = reinterpret_cast<double *>(smem) + wave * LDS_WAVE_SLOTS;
    const double *partner = reinterpret_cast<double *>(smem) + (wave ^ 2) * LDS_WAVE_SLOTS;  // other poly
    const double *sibling = reinterpret_cast<double *>(smem) + (wave ^ 1) * LDS_WAVE_SLOTS;  // other prime
    double *sib_w = reinterpret_cast<double *>(smem) + (wave ^ 1) * LDS_WAVE_SLOTS;

    const double p = q ? (double)NTT_P1 : (double)NTT_P0;
    const double pinv = 1.0 / p;
    const double p1 = (double)NTT_P1, p1inv = 1.0 / p1;
    const double crt_c = C_CRT;

    const uint64_t *ks = P.ks + (size_t)ct * SMALL_CT;
    const double *fwd_lane = P.tw.fwd_lane + q * 32 * 64;
    const double *inv_lane = P.tw.inv_lane + q * 32 * 64;
    const double twA = lane < 32 ? C_FWD_UNI[q][lane] : C_INV_UNI[q][lane];
    const double twB = C_INV_UNI[q][lane & 31];
    typedef const __attribute__((address_space(1))) double *gd_t;
    typedef const __attribute__((address_space(4))) double *cd_t;
    const gd_t mono = (gd_t)(P.mono + q * 4096);          // psi_q^k, k < 4096
    const cd_t mono_u = (cd_t)(P.mono + q * 4096);        // the same table for wave-uniform reads
    const uint32_t lane_root = 2u * (uint32_t)rev6(lane) + 1u;      // slot root = psi^(lane_root + 128 rev5(c))

    // wave (j, q) owns the accumulator coefficients n = lane + 64 r with r = 2 o + q (see pbs_kernels.hip)
    uint64_t acc[16];
    {
        const uint32_t b = mod_switch(ks[LWE_N]);
        const uint32_t a = (2 * POLY_N - b) & (2 * POLY_N - 1);
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
        }
    }

    for (int pr = 0; pr < LWE_N / 2; pr++) {
        const uint32_t e1 = mod_switch(ks[2 * pr]), e2 = mod_switch(ks[2 * pr + 1]);
        if ((e1 | e2) == 0) continue;                     // both monomials are 1: the product is exactly zero

        // ---- decompose the accumulator itself; the sibling (other prime) gets the digits of the owned half ----
        double x[32];
        __builtin_amdgcn_s_setprio(2);
        if (q == 0) mb_phase_digits<0>(x, acc, sib_w, lane);
        else mb_phase_digits<1>(x, acc, sib_w, lane);
        __syncthreads();
        if (q == 0) phase_other_digits<0>(x, my, lane);
        else phase_other_digits<1>(x, my, lane);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_setprio(0);

        // per-lane monomial bases (one gather each), requested before the transform
        const double tla = mono[(lane_root * e1) & 4095u], tlb = mono[(lane_root * e2) & 4095u];

        ntt_forward(x, my, lane, twA, fwd_lane, p, pinv);
        __builtin_amdgcn_s_setprio(2);

        // ---- publish, pointwise multiply-accumulate with the combined key ----
        // pair key [pair][K1,K2,K3][row][col][prime][16 register pairs][64 lanes][2]: one 16-byte load per lane covers
        // the registers (c, c + 1); this wave's column j, prime q, of both rows
        typedef double __attribute__((ext_vector_type(2))) double2_t;
        typedef const __attribute__((address_space(1))) double2_t *gk_t;
        const gk_t kbase = (gk_t)(P.bsk_ntt_mb + ((size_t)pr * 12 + j) * 2 * POLY_N + (size_t)q * POLY_N) + lane;
        // key t of row `row`: kbase[((t * 2 + row) * 2) * POLY_N (doubles -> double2: / 2 ... see idx()) + cpair * 64]
        auto idx = [](int t, int row, int cpair) { return ((t * 2 + row) * 2) * POLY_N + cpair * 64; };
        // ring over register pairs: the rows of pair cp + PF are requested while pair cp is computed
        double2_t ko[PF + 1][3], kp[PF + 1][3];
#pragma unroll
        for (int h = 0; h < PF; h++)
#pragma unroll
            for (int t = 0; t < 3; t++) { ko[h][t] = kbase[idx(t, j, h)]; kp[h][t] = kbase[idx(t, 1 - j, h)]; }   // over the barrier
#pragma unroll
        for (int c = 0; c < 32; c++) my[c * 64 + lane] = x[c];
        __syncthreads();
        {
            const uint32_t s1 = (e1 & 1u) << 31, s2 = (e2 & 1u) << 31;      // odd register: psi^(2048 e) = (-1)^e
#pragma unroll
            for (int cp = 0; cp < 16; cp++) {
                if (cp + PF < 16) {
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        ko[(cp + PF) % (PF + 1)][t] = kbase[idx(t, j, cp + PF)];
                        kp[(cp + PF) % (PF + 1)][t] = kbase[idx(t, 1 - j, cp + PF)];
                    }
                }
                const double2_t (&KO)[3] = ko[cp % (PF + 1)];
                const double2_t (&KP)[3] = kp[cp % (PF + 1)];
                const int c = 2 * cp;
                // wave-uniform 32nd roots psi^(128 m), m = rev5(c) e = rev4(cp) e  (mod 32)
                const int r4 = ((cp & 1) << 3) | ((cp & 2) << 1) | ((cp & 4) >> 1) | ((cp & 8) >> 3);
                const double ua = mono_u[128u * ((r4 * e1) & 31u)], ub = mono_u[128u * ((r4 * e2) & 31u)];
                const double a0 = mulmod(tla, ua, p, pinv), b0 = mulmod(tlb, ub, p, pinv);
                const double ab0 = mulmod(a0, b0, p, pinv);
                const double a1 = flip_sign(a0, s1), b1 = flip_sign(b0, s2), ab1 = flip_sign(ab0, s1 ^ s2);
                const double am0 = a0 - 1.0, bm0 = b0 - 1.0, abm0 = ab0 - 1.0;
                const double am1 = a1 - 1.0, bm1 = b1 - 1.0, abm1 = ab1 - 1.0;
                const double o0 = partner[c * 64 + lane], o1 = partner[(c + 1) * 64 + lane];
                // combined key of each row at the two points, |A| <= 1.6 p
                const double Ao0 = mulmod(KO[0].x, am0, p, pinv) + mulmod(KO[1].x, bm0, p, pinv) + mulmod(KO[2].x, abm0, p, pinv);
                const double Ap0 = mulmod(KP[0].x, am0, p, pinv) + mulmod(KP[1].x, bm0, p, pinv) + mulmod(KP[2].x, abm0, p, pinv);
                const double Ao1 = mulmod(KO[0].y, am1, p, pinv) + mulmod(KO[1].y, bm1, p, pinv) + mulmod(KO[2].y, abm1, p, pinv);
                const double Ap1 = mulmod(KP[0].y, am1, p, pinv) + mulmod(KP[1].y, bm1, p, pinv) + mulmod(KP[2].y, abm1, p, pinv);
                x[c] = mulmod(x[c], Ao0, p, pinv) + mulmod(o0, Ap0, p, pinv);
                x[c + 1] = mulmod(x[c + 1], Ao1, p, pinv) + mulmod(o1, Ap1, p, pinv);
            }
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(1);

        ntt_inverse(x, my, lane, twA, twB, inv_lane, p, pinv);
        __builtin_amdgcn_s_setprio(2);

        // ---- exchange residues, CRT for the owned half ----
        if (q == 0) phase_publish_residues<0>(x, my, lane);
        else phase_publish_residues<1>(x, my, lane);
        __syncthreads();
        if (q == 0) mb_phase_crt<0>(x, acc, sibling, lane, crt_c, p1, p1inv);
        else mb_phase_crt<1>(x, acc, sibling, lane, crt_c, p1, p1inv);
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
    } else {
        if (q == 0 && lane == 0) out[BIG_N] = acc[0];
        uint64_t *body = P.body_ptrs ? P.body_ptrs[ct] : nullptr;      // rotation sharing: the whole body polynomial
        if (body) {
#pragma unroll
            for (int o = 0; o < 16; o++) body[lane + 64 * (2 * o + q)] = acc[o];
        }
    }
}

hipError_t prepare_device_for_ntt_mb2() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(blind_rotate_ntt_mb2_kernel<1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)blind_rotate_lds_bytes());
}

hipError_t launch_blind_rotate_ntt_mb2(const BlindRotateNttMb2Params &p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    // key rows are requested one register pair ahead; two or three ahead measured 84.4 / 91.6 ms per 3968 bootstraps
    // against 82.6 (spills: the kernel is VALU-bound at 94 % busy, not waiting for the key)
    hipLaunchKernelGGL((blind_rotate_ntt_mb2_kernel<1>), dim3(p.B), dim3(256), blind_rotate_lds_bytes(), s, p);
    return hipGetLastError();
}

}  // namespace fhs
