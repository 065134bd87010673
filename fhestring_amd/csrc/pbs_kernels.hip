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
#include "ntt_transform.h"

namespace fhs {

#pragma clang fp contract(off)

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

    uint64_t ks_next = ks[0];   // requested one iteration ahead (ks[LWE_N], the body, is a valid address)
    for (int i = 0; i < LWE_N; i++) {
        asm volatile("" : "+v"(ks_next));
        const uint32_t a = __builtin_amdgcn_readfirstlane(mod_switch(ks_next));
        __builtin_amdgcn_sched_barrier(0);
        ks_next = ks[i + 1];
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
    } else {
        if (q == 0 && lane == 0) out[BIG_N] = acc[0];
        uint64_t *body = P.body_ptrs ? P.body_ptrs[ct] : nullptr;      // rotation sharing: the whole body polynomial
        if (body) {
#pragma unroll
            for (int o = 0; o < 16; o++) body[lane + 64 * (2 * o + q)] = acc[o];
        }
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

// Rotation sharing: sample extraction of negacyclic coefficient K (index in [0, 2N)) of a finished blind rotation whose
// coefficient-0 extraction E = desc.lead is in memory (E[0] = A[0], E[j] = -A[N-j]: the mask polynomial A, rearranged) and
// whose body polynomial B = desc.body was stored by the blind-rotation kernel (body_ptrs).  With h = K mod N:
//     a_i = A[h-i] (i <= h), -A[N+h-i] (i > h)   =   E[i-h] (i >= h), -E[N-h+i] (i < h);      b = B[h];
// K >= N negates the whole LWE (X^N = -1).  Exact integer moves: the result equals the CPU oracle's orc_pbs_shifted
// bit for bit.  One workgroup per extraction, 16 KB in, 16 KB out.
__global__ __launch_bounds__(256) void extract_shift_kernel(const ExtractDesc *__restrict__ desc) {
    const ExtractDesc d = desc[blockIdx.x];
    const uint32_t h = d.K & (POLY_N - 1);
    const uint64_t neg = d.K >= (uint32_t)POLY_N ? ~(uint64_t)0 : 0;
    for (uint32_t i = threadIdx.x; i < (uint32_t)POLY_N; i += 256) {
        const uint64_t v = i >= h ? d.lead[i - h] : (uint64_t)0 - d.lead[POLY_N - h + i];
        d.out[i] = (v ^ neg) - neg;
    }
    if (threadIdx.x == 0) d.out[BIG_N] = (d.body[h] ^ neg) - neg;
}
hipError_t launch_extract_shift(const ExtractDesc *d_desc, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(extract_shift_kernel, dim3(n), dim3(256), 0, s, d_desc);
    return hipGetLastError();
}

}  // namespace fhs
