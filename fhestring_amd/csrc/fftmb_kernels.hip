// Blind rotation with TWO key bits per external product (f64 FFT arithmetic, gfx950 only): FHS_ARITH_F64_FFT_MB2.
//
// The classic blind rotation (fft_kernels.hip) spends one GGSW x GLWE external product per LWE key bit:
//     ACC <- ACC + BSK_i (.) (X^{a_i} ACC - ACC),                                    742 products per bootstrap.
// For a pair of bits (s, s') with mod-switched mask elements (e1, e2)
//     X^{e1 s + e2 s'} = 1 + s(1-s') (X^{e1} - 1) + (1-s)s' (X^{e2} - 1) + s s' (X^{e1+e2} - 1),
// so with GGSW encryptions K1, K2, K3 of the three products (fhs_client_bsk_mb2; same GLWE parameters, same
// decomposition base 2^23, one level) ONE external product per pair does the work of two:
//     ACC <- ACC + [ K1 (X^{e1} - 1) + K2 (X^{e2} - 1) + K3 (X^{e1+e2} - 1) ] (.) ACC,      371 products per bootstrap.
// The bracket is formed in the Fourier domain, where multiplying by a monomial is a pointwise multiplication by the
// evaluation point raised to the exponent.  This is the "multi-bit" bootstrap of tfhe-rs' GPU backend (Joye-Paillier
// 2022; Bourse et al. 2018) at group size 2, on the reference's own parameter set: LWE dimension, polynomial size,
// bases, noise distributions and the keyswitch are unchanged.  The price is noise: the bootstrap output's sigma is
// 2^49.62 instead of 2^48.87 (the decomposition rounding enters through (X^e - 1), and the f64 rounding of three GGSW
// products per pair instead of one per bit dominates; measured in tests/test_gpu_noise.py, which holds the string layer's
// DAGs to the same margins in this arithmetic: total error at the design limit 9.05 instead of 8.98 units of 2^52).
//
// Per pair and point (root rho = w^(4j+1), w = exp(i pi/2048), j = bitrev of the point's position):
//     a = rho^e1, b = rho^e2;   A_row = K1 (a - 1) + K2 (b - 1) + K3 (a b - 1);     out = F_own A_own + F_partner A_partner
// (a Horner form with a precomputed fourth polynomial -(K1 + K2 + K3) needs 3 FP64 instructions less per point but a
// third more key bytes: 29.1 ms instead of 21.6 ms per 3968 bootstraps -- the key stream is what this kernel waits for)
// a and b are a per-lane base (one table gather per monomial) times a wave-uniform 16th root of unity per register
// (scalar loads); the two points of a register pair differ by the sign (-1)^e.  Against two classic iterations this
// needs 0.70x the FP64 instructions, 0.4x the LDS traffic (no rotated re-read of the accumulator, half the
// transposes), half the workgroup barriers and 1.5x the key bytes.
//
// Same mapping as fft_kernels.hip: a workgroup of 2 wavefronts per ciphertext (persistent), wavefront j owns GLWE
// polynomial j, 16 complex points per lane; same transform (fft_transform.h), so mode 4 of the CPU oracle mirrors
// the kernel operation for operation and the output is checked bit for bit.
#include "fft_transform.h"

namespace fhs {

#pragma clang fp contract(off)

namespace {
using namespace fftdev;

typedef double __attribute__((ext_vector_type(2))) double2_t;

struct publish_hook {                                     // last forward stage: publish finished points for the partner
    cplx *pub; const cplx *z;
    __device__ __forceinline__ void operator()(int a, int b) const {
        pub[64 * a] = z[a];
        pub[64 * b] = z[b];
        __builtin_amdgcn_sched_barrier(0x7);
    }
};

__device__ __forceinline__ cplx cmulc(cplx a, cplx w) { return cmul(a, w.r, w.i); }
// acc + w * k   (same fused order as the forward butterfly's sum)
__device__ __forceinline__ cplx cmac(cplx w, double2_t k, cplx acc) {
    cplx t;
    t.r = __builtin_fma(-w.i, k.y, __builtin_fma(w.r, k.x, acc.r));
    t.i = __builtin_fma(w.i, k.x, __builtin_fma(w.r, k.y, acc.i));
    return t;
}
__device__ __forceinline__ cplx flip_if(cplx a, uint32_t signmask) {       // exact: toggles the sign bits
    typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
    u32x2 r = __builtin_bit_cast(u32x2, a.r), i = __builtin_bit_cast(u32x2, a.i);
    r.y ^= signmask; i.y ^= signmask;
    cplx o; o.r = __builtin_bit_cast(double, r); o.i = __builtin_bit_cast(double, i);
    return o;
}
__device__ __forceinline__ int rev6(int x) { return (int)(__builtin_bitreverse32((uint32_t)x) >> 26); }

}  // namespace

template <int PF>
__global__ __launch_bounds__(128, 2) void blind_rotate_mb2_kernel(BlindRotateMb2Params P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int j = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // GLWE polynomial of this wave
    double *my = reinterpret_cast<double *>(smem) + j * FFT_LDS_DOUBLES;
    const double *partner = reinterpret_cast<double *>(smem) + (1 - j) * FFT_LDS_DOUBLES;
    int *next_ct = reinterpret_cast<int *>(smem + 2 * FFT_LDS_DOUBLES * sizeof(double));
    const uint32_t lane_root = 4u * (uint32_t)rev6(lane) + 1u;        // rho = w^(lane_root + 256 rev4(c))

    for (;;) {                                            // persistent workgroups, see fft_kernels.hip
    if (threadIdx.x == 0) *next_ct = (int)atomicAdd(P.work_counter, 1u);
    __syncthreads();
    const int ct = __builtin_amdgcn_readfirstlane(*next_ct);
    __syncthreads();
    if (ct >= P.B) break;

    const uint64_t *ks = P.ks + (size_t)ct * SMALL_CT;

    // acc[r] = coefficient (lane + 64 r) of polynomial j (u64 torus); registers r and r+16 form one complex point
    uint64_t acc[32];
    {
        const uint32_t b = fft_mod_switch(ks[LWE_N]);
        const uint32_t a = (2 * POLY_N - b) & (2 * POLY_N - 1);
        const uint32_t s = a & (POLY_N - 1);
        const bool neg = a >= POLY_N;
        const uint64_t *lut = P.luts + (size_t)P.lut_idx[ct] * POLY_N;
#pragma unroll
        for (int r = 0; r < 32; r++) {
            uint64_t v = 0;
            if (j == 1) {
                const uint32_t n = lane + 64 * r;
                v = lut[(n - s) & (POLY_N - 1)];
                if ((n < s) != neg) v = (uint64_t)0 - v;
            }
            acc[r] = v;
        }
    }

    // the pair of mask elements of the NEXT iteration is requested one iteration ahead (see fft_kernels.hip)
    uint64_t kn0 = ks[0], kn1 = ks[1];
    for (int p = 0; p < LWE_N / 2; p++) {
        asm volatile("" : "+v"(kn0), "+v"(kn1));          // opaque until here: nothing of the next iteration is computed early
        const uint32_t e1 = __builtin_amdgcn_readfirstlane(fft_mod_switch(kn0));
        const uint32_t e2 = __builtin_amdgcn_readfirstlane(fft_mod_switch(kn1));
        __builtin_amdgcn_sched_barrier(0);
        kn0 = ks[2 * p + 2];                              // p = 370: the body, a valid address
        kn1 = ks[2 * p + 3 < SMALL_CT ? 2 * p + 3 : SMALL_CT - 1];
        if ((e1 | e2) == 0) continue;                     // both monomials are 1: the product is exactly zero
        __builtin_amdgcn_s_setprio(1);

        const double *lanetab = P.lanetab;
        asm volatile("" : "+s"(lanetab));

        // decompose the accumulator itself (closest multiple of 2^41 as a signed 23-bit digit); fold
        cplx z[16];
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const uint32_t dhi = (uint32_t)(acc[r] >> 32);
            const int32_t dig = (int32_t)(dhi + 0x100u) >> 9;
            if (r < 16) z[r].r = (double)dig; else z[r - 16].i = (double)dig;
        }

        // per-lane monomial bases rho_lane^e (one gather each) and the key rows of the first register pair
        typedef const __attribute__((address_space(1))) double2_t *gk_t;
        const gk_t mono = (gk_t)P.mono;
        const double2_t la2 = mono[(lane_root * e1) & 4095u], lb2 = mono[(lane_root * e2) & 4095u];
        // key: [pair][K1,K2,K3][row][col][16][64] complex; this wave's column j of both rows
        const gk_t kbase = (gk_t)P.bsk_mb + ((size_t)p * 12 + j) * FM + lane;
        // slot (q, row) at kbase[((q*2 + row)*2) * FM + c*64]
        // half step h = 2 c + (0: own row, 1: partner row); the rows of half step h + PF are requested while h is computed
        constexpr int NQ = 3;
        double2_t kb[PF + 1][NQ];                         // ring over half steps; [K1, K2, K3]
#pragma unroll
        for (int h = 0; h < PF; h++)
#pragma unroll
            for (int q = 0; q < NQ; q++) kb[h][q] = kbase[((q * 2 + ((h & 1) ? 1 - j : j)) * 2) * FM + (h >> 1) * 64];

        {
            LaneTw tw;
            load_lane_tw(tw, lanetab, lane);
            fft_forward(z, my, lane, tw);
            fft_forward_last(z, tw, publish_hook{reinterpret_cast<cplx *>(my) + lane, z});
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(2);
        {
            cplx la; la.r = la2.x; la.i = la2.y;
            cplx lb; lb.r = lb2.x; lb.i = lb2.y;
            const uint32_t s1 = (e1 & 1u) << 31, s2 = (e2 & 1u) << 31;   // odd registers: rho^e picks up (-1)^e
            typedef const __attribute__((address_space(4))) double2_t *ck_t;
            const ck_t r16 = (ck_t)P.r16;
            const cplx *par = reinterpret_cast<const cplx *>(partner) + lane;
            cplx gq[2] = {par[0], par[64]};               // two partner points in flight ahead of their use
            cplx a, b, a1, b1, ab1;
            double rr = 0, ii = 0;
#pragma unroll
            for (int h = 0; h < 32; h++) {
                const int c = h >> 1, row = h & 1;
                if (h + PF < 32) {
                    const int hn = h + PF, cn = hn >> 1, rown = (hn & 1) ? 1 - j : j;
#pragma unroll
                    for (int q = 0; q < NQ; q++) kb[hn % (PF + 1)][q] = kbase[((q * 2 + rown) * 2) * FM + cn * 64];
                }
                const double2_t (&k)[NQ] = kb[h % (PF + 1)];
                if (row == 0) {
                    if ((c & 1) == 0) {
                        // r = rev4(c) mod 8 = rev3(c >> 1); wave-uniform 16th roots exp(i pi r e / 8)
                        const int r3 = ((c >> 1) & 1) << 2 | ((c >> 1) & 2) | ((c >> 1) & 4) >> 2;
                        const double2_t ua = r16[(r3 * e1) & 15u], ub = r16[(r3 * e2) & 15u];
                        a = cmul(la, ua.x, ua.y);
                        b = cmul(lb, ub.x, ub.y);
                    } else {
                        a = flip_if(a, s1); b = flip_if(b, s2);
                    }
                    a1 = a; a1.r = a.r - 1.0;             // a - 1, b - 1, a b - 1
                    b1 = b; b1.r = b.r - 1.0;
                    ab1 = cmulc(a, b); ab1.r = ab1.r - 1.0;
                }
                cplx A;
                {
                    cplx y; y.r = k[0].x; y.i = k[0].y;
                    A = cmulc(y, a1);                     // K1 (a - 1)
                    A = cmac(b1, k[1], A);                // + K2 (b - 1)
                    A = cmac(ab1, k[2], A);               // + K3 (a b - 1)
                }
                if (row == 0) {
                    const double fr = z[c].r, fi = z[c].i;
                    rr = fr * A.r; rr = __builtin_fma(-fi, A.i, rr);
                    ii = fr * A.i; ii = __builtin_fma(fi, A.r, ii);
                } else {
                    const cplx g = gq[c & 1];
                    if (c + 2 < 16) gq[c & 1] = par[(c + 2) * 64];
                    rr = __builtin_fma(g.r, A.r, rr); rr = __builtin_fma(-g.i, A.i, rr);
                    ii = __builtin_fma(g.r, A.i, ii); ii = __builtin_fma(g.i, A.r, ii);
                    z[c].r = rr; z[c].i = ii;
                }
            }
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(0);

        {
            const double *lanetab2 = P.lanetab;
            asm volatile("" : "+s"(lanetab2));
            LaneTw tw2;
            load_lane_tw(tw2, lanetab2, lane);
            fft_inverse(z, my, lane, tw2);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            acc[r] += to_torus(z[r].r);
            acc[r + 16] += to_torus(z[r].i);
        }
    }

    uint64_t *out = P.out_ptrs ? P.out_ptrs[ct] : P.out + (size_t)ct * BIG_CT;
    if (j == 0) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const int n = lane + 64 * r;
            if (n == 0) out[0] = acc[r];
            else out[POLY_N - n] = (uint64_t)0 - acc[r];
        }
    } else {
        if (lane == 0) out[BIG_N] = acc[0];
        uint64_t *body = P.body_ptrs ? P.body_ptrs[ct] : nullptr;      // rotation sharing: the whole body polynomial
        if (body) {
#pragma unroll
            for (int r = 0; r < 32; r++) body[lane + 64 * r] = acc[r];
        }
    }
    }   // persistent loop
}

hipError_t launch_blind_rotate_mb2(const BlindRotateMb2Params &p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    const size_t lds = (size_t)2 * FFT_LDS_DOUBLES * sizeof(double) + 16;
    hipError_t e = hipMemsetAsync(p.work_counter, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const int grid = p.B < p.slots ? p.B : p.slots;
    hipLaunchKernelGGL((blind_rotate_mb2_kernel<1>), dim3(grid), dim3(128), lds, s, p);   // key rows one half step ahead
    return hipGetLastError();
}

}  // namespace fhs
