// Optional arithmetic mode: blind rotation with an f64 complex FFT (gfx950 only).
//
// This is the algorithm CLASS the reference's CPU engine uses (tfhe 0.5.2 + concrete-fft 0.4.0,
// Cargo.lock:168-179): the negacyclic product through a folded 1024-point complex transform
// (z[n] = x[n] + i x[n+1024]).  It needs ~3.7x fewer FP64 operations than the exact two-prime NTT of
// pbs_kernels.hip but is approximate w.r.t. exact integer arithmetic (53-bit mantissa; the error is
// far below the scheme's noise, like in the reference).  The exact NTT stays the default and the
// parity anchor; this mode is selected with fhs_set_arithmetic(ctx, FHS_ARITH_F64_FFT).
// It is deterministic, and mode 3 of the CPU oracle mirrors it lane for lane with the same
// IEEE-754 operation order, so the GPU output is still checked bit for bit.
//
// Mapping: one workgroup of 2 wavefronts per ciphertext, wavefront j owns GLWE polynomial j:
// 16 complex points per lane in registers, 4 radix-2 stages in the strided layout (lane = n mod 64,
// lane-uniform twiddles), one transpose through LDS with the two cross-lane stages fused into the
// transposed read as a radix-4 step, 4 stages in the contiguous layout (lane = n div 16).
// The twist of the negacyclic fold is merged into the twiddle table
//   W[m+i] = exp(i*pi/2048 * (1024/2m) * (4*bitrev(i) + 1))     (host: fft_tables.cpp).
#include "pbs_kernels.h"

namespace fhs {

#pragma clang fp contract(off)

namespace {

constexpr int FM = 1024;                       // complex points
constexpr int FFT_LDS_DOUBLES = 2176;          // per wave: 1040 complex slots used (same 17 408 B as the NTT path)
__device__ __forceinline__ int fslot(int n) { return n + (n >> 6); }   // complex slot, 1 pad per 64

struct cplx { double r, i; };
__device__ __forceinline__ cplx cmul(cplx a, double wr, double wi) {
    cplx t;
    t.r = __builtin_fma(-a.i, wi, a.r * wr);
    t.i = __builtin_fma(a.i, wr, a.r * wi);
    return t;
}
__device__ __forceinline__ double bcast(double v, int k) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)b, k);
    const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), k);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint32_t fft_mod_switch(uint64_t x) { return (uint32_t)(((x + (1ull << 51)) >> 52) & 4095u); }

// lane-distributed uniform constants: twWr/twWi lane k = Re/Im W[k] (k < 64); twU lanes 0..15 = Re U[k],
// lanes 16..31 = Im U[k-16]
#define W_RE(k) bcast(twWr, (k))
#define W_IM(k) bcast(twWi, (k))
#define U_RE(k) bcast(twU, (k))
#define U_IM(k) bcast(twU, 16 + (k))

// forward: z[r] = point (lane + 64 r)  ->  z[c] = value at array index 16*lane + c
__device__ __forceinline__ void fft_forward(cplx (&z)[16], double *lds, int lane, double twWr, double twWi, double twU,
                                            const double *__restrict__ lanetab /* [12][64] */) {
    // per-lane constants (L1-resident table): fused-stage twiddles and the 4 in-lane bases
    const double war = lanetab[0 * 64 + lane], wai = lanetab[1 * 64 + lane];
    const double wbr = lanetab[2 * 64 + lane], wbi = lanetab[3 * 64 + lane];
    double bre[4], bim[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { bre[k] = lanetab[(4 + 2 * k) * 64 + lane]; bim[k] = lanetab[(5 + 2 * k) * 64 + lane]; }
#pragma unroll
    for (int T = 8; T >= 1; T >>= 1) {
        const int m = 8 / T;
#pragma unroll
        for (int i = 0; i < m; i++) {
            const double wr = W_RE(m + i), wi = W_IM(m + i);
#pragma unroll
            for (int r = 2 * i * T; r < 2 * i * T + T; r++) {
                const cplx v = cmul(z[r + T], wr, wi);
                const cplx u = z[r];
                z[r].r = u.r + v.r; z[r].i = u.i + v.i;
                z[r + T].r = u.r - v.r; z[r + T].i = u.i - v.i;
            }
        }
    }
    {   // fslot(lane + 64 r) == lane + 65 r
        cplx *wr_ = reinterpret_cast<cplx *>(lds) + lane;
#pragma unroll
        for (int r = 0; r < 16; r++) wr_[65 * r] = z[r];
    }
    __builtin_amdgcn_wave_barrier();
    {
        const int q = lane & 3, gL = lane >> 2;
        const double s1 = q < 2 ? 1.0 : -1.0, s2 = (q & 1) ? -1.0 : 1.0;
        const cplx *rd = reinterpret_cast<const cplx *>(lds) + 65 * gL;   // fslot(64 gL + c) == 65 gL + c
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const cplx e0 = rd[c], e1 = rd[c + 16], e2 = rd[c + 32], e3 = rd[c + 48];
            const cplx t2 = cmul(e2, war, wai), t3 = cmul(e3, war, wai);
            cplx A, Bv;
            A.r = __builtin_fma(s1, t2.r, e0.r); A.i = __builtin_fma(s1, t2.i, e0.i);
            Bv.r = __builtin_fma(s1, t3.r, e1.r); Bv.i = __builtin_fma(s1, t3.i, e1.i);
            const cplx tb = cmul(Bv, wbr, wbi);
            z[c].r = __builtin_fma(s2, tb.r, A.r);
            z[c].i = __builtin_fma(s2, tb.i, A.i);
        }
    }
    __builtin_amdgcn_wave_barrier();
    int lg = 0;
#pragma unroll
    for (int t = 8; t >= 1; t >>= 1, lg++) {
        const int G = 8 / t;
#pragma unroll
        for (int g = 0; g < G; g++) {
            double wr = bre[lg], wi = bim[lg];
            if (g) {
                cplx b; b.r = bre[lg]; b.i = bim[lg];
                const cplx w = cmul(b, U_RE(G + g), U_IM(G + g));
                wr = w.r; wi = w.i;
            }
#pragma unroll
            for (int c = 2 * g * t; c < 2 * g * t + t; c++) {
                const cplx v = cmul(z[c + t], wr, wi);
                const cplx u = z[c];
                z[c].r = u.r + v.r; z[c].i = u.i + v.i;
                z[c + t].r = u.r - v.r; z[c + t].i = u.i - v.i;
            }
        }
    }
}

// inverse (unscaled): z[c] at array index 16*lane + c  ->  z[r] = point (lane + 64 r)
__device__ __forceinline__ void fft_inverse(cplx (&z)[16], double *lds, int lane, double twWr, double twWi, double twU,
                                            const double *__restrict__ lanetab) {
    double bre[4], bim[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { bre[k] = lanetab[(4 + 2 * k) * 64 + lane]; bim[k] = lanetab[(5 + 2 * k) * 64 + lane]; }
    int lg = 3;
#pragma unroll
    for (int t = 1; t <= 8; t <<= 1, lg--) {
        const int G = 8 / t;
#pragma unroll
        for (int g = 0; g < G; g++) {
            double wr = bre[lg], wi = bim[lg];
            if (g) {
                cplx b; b.r = bre[lg]; b.i = bim[lg];
                const cplx w = cmul(b, U_RE(G + g), U_IM(G + g));
                wr = w.r; wi = w.i;
            }
#pragma unroll
            for (int c = 2 * g * t; c < 2 * g * t + t; c++) {
                const cplx u = z[c], v = z[c + t];
                z[c].r = u.r + v.r; z[c].i = u.i + v.i;
                cplx d; d.r = u.r - v.r; d.i = u.i - v.i;
                z[c + t] = cmul(d, wr, -wi);                       // conjugate twiddle
            }
        }
    }
    {   // fslot(16 lane + c) == 16 lane + c + (lane >> 2)
        cplx *wr_ = reinterpret_cast<cplx *>(lds) + 16 * lane + (lane >> 2);
#pragma unroll
        for (int c = 0; c < 16; c++) wr_[c] = z[c];
    }
    __builtin_amdgcn_wave_barrier();
    {
        const int q = (lane >> 4) & 3, l4 = lane & 15;
        const bool odd = q & 1, upper = q >= 2;
        const double s2 = odd ? -1.0 : 1.0, s1 = upper ? -1.0 : 1.0;
        const cplx *rd = reinterpret_cast<const cplx *>(lds) + l4;   // fslot(64 r) + l4 == 65 r + l4
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const cplx e0 = rd[65 * r], e1 = rd[65 * r + 16], e2 = rd[65 * r + 32], e3 = rd[65 * r + 48];
            cplx d01, d23;
            d01.r = __builtin_fma(s2, e1.r, e0.r); d01.i = __builtin_fma(s2, e1.i, e0.i);
            d23.r = __builtin_fma(s2, e3.r, e2.r); d23.i = __builtin_fma(s2, e3.i, e2.i);
            // twiddles stay scalar (readlane); lanes that take the sum branch keep the untwiddled value
            const cplx p1 = cmul(d01, W_RE(32 + 2 * r), -W_IM(32 + 2 * r));
            const cplx q1 = cmul(d23, W_RE(32 + 2 * r + 1), -W_IM(32 + 2 * r + 1));
            const cplx p = odd ? p1 : d01, qv = odd ? q1 : d23;
            cplx h;
            h.r = __builtin_fma(s1, qv.r, p.r); h.i = __builtin_fma(s1, qv.i, p.i);
            const cplx h1 = cmul(h, W_RE(16 + r), -W_IM(16 + r));
            z[r] = upper ? h1 : h;
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int T = 1; T <= 8; T <<= 1) {
        const int h = 8 / T;
#pragma unroll
        for (int i = 0; i < h; i++) {
            const double wr = W_RE(h + i), wi = -W_IM(h + i);
#pragma unroll
            for (int r = 2 * i * T; r < 2 * i * T + T; r++) {
                const cplx u = z[r], v = z[r + T];
                z[r].r = u.r + v.r; z[r].i = u.i + v.i;
                cplx d; d.r = u.r - v.r; d.i = u.i - v.i;
                z[r + T] = cmul(d, wr, wi);
            }
        }
    }
}

// torus value (mod 2^64) of an approximately integral double of any magnitude
__device__ __forceinline__ uint64_t to_torus(double v) {
    const double k = __builtin_floor(v * 5.421010862427522e-20 + 0.5);           // 2^-64
    const double rr = __builtin_fma(-k, 18446744073709551616.0, v);              // in [-2^63, 2^63), exact
    const double hi = __builtin_floor(rr * 2.3283064365386963e-10);              // 2^-32
    const double lo = __builtin_fma(-hi, 4294967296.0, rr);                      // in [0, 2^32), exact
    return ((uint64_t)(int64_t)(int32_t)hi << 32) + (uint64_t)(uint32_t)lo;
}

}  // namespace

__global__ __launch_bounds__(128, 2) void blind_rotate_fft_kernel(BlindRotateFftParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ct = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int j = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // GLWE polynomial of this wave
    double *my = reinterpret_cast<double *>(smem) + j * FFT_LDS_DOUBLES;
    const double *partner = reinterpret_cast<double *>(smem) + (1 - j) * FFT_LDS_DOUBLES;
    uint64_t *my_u = reinterpret_cast<uint64_t *>(my);

    const uint64_t *ks = P.ks + (size_t)ct * SMALL_CT;
    const double twWr = P.w_re[lane], twWi = P.w_im[lane];
    const double twU = lane < 16 ? P.u_re[lane] : P.u_im[(lane - 16) & 15];

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
                v = (n >= s) ? lut[n - s] : (uint64_t)0 - lut[n - s + POLY_N];
                if (neg) v = (uint64_t)0 - v;
            }
            acc[r] = v;
        }
    }

    for (int i = 0; i < LWE_N; i++) {
        const uint32_t a = fft_mod_switch(ks[i]);
        if (a == 0) continue;
        const uint32_t s = a & (POLY_N - 1);
        const bool neg = a >= POLY_N;

        // rotate, subtract, decompose; fold: z[r] = digit[r] + i * digit[r + 16]
        cplx z[16];
#pragma unroll
        for (int r = 0; r < 32; r++) my_u[lane + 64 * r] = acc[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const uint32_t n = lane + 64 * r;
            const uint32_t m = (n - s) & (POLY_N - 1);
            uint64_t v = my_u[m];
            if ((n < s) != neg) v = (uint64_t)0 - v;
            const uint64_t d = v - acc[r];
            const int32_t dig = (int32_t)((uint32_t)(d >> 32) + 0x100u) >> 9;
            if (r < 16) z[r].r = (double)dig; else z[r - 16].i = (double)dig;
        }
        __builtin_amdgcn_wave_barrier();

        fft_forward(z, my, lane, twWr, twWi, twU, P.lanetab);

        // publish, pointwise: own transform first (row j), then the partner's (row 1-j); column j
        typedef double __attribute__((ext_vector_type(2))) double2_t;
        const double2_t *b_own = reinterpret_cast<const double2_t *>(P.bsk_fft) + ((((size_t)i * 2 + j) * 2 + j)) * FM + lane;
        const double2_t *b_par = reinterpret_cast<const double2_t *>(P.bsk_fft) + ((((size_t)i * 2 + (1 - j)) * 2 + j)) * FM + lane;
        constexpr int CH = 2;
        double2_t bo[CH], bp[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) { bo[k] = b_own[k * 64]; bp[k] = b_par[k * 64]; }
        {
            cplx *pub = reinterpret_cast<cplx *>(my) + lane;
#pragma unroll
            for (int c = 0; c < 16; c++) pub[c * 64] = z[c];
        }
        __syncthreads();
        {
            const cplx *par = reinterpret_cast<const cplx *>(partner) + lane;
#pragma unroll
            for (int ch = 0; ch < 16 / CH; ch++) {
                double2_t no[CH], np[CH];
                if (ch + 1 < 16 / CH) {
#pragma unroll
                    for (int k = 0; k < CH; k++) { no[k] = b_own[((ch + 1) * CH + k) * 64]; np[k] = b_par[((ch + 1) * CH + k) * 64]; }
                }
#pragma unroll
                for (int k = 0; k < CH; k++) {
                    const int c = ch * CH + k;
                    const cplx g = par[c * 64];
                    const double fr = z[c].r, fi = z[c].i;
                    double rr = fr * bo[k].x; rr = __builtin_fma(-fi, bo[k].y, rr);
                    rr = __builtin_fma(g.r, bp[k].x, rr); rr = __builtin_fma(-g.i, bp[k].y, rr);
                    double ii = fr * bo[k].y; ii = __builtin_fma(fi, bo[k].x, ii);
                    ii = __builtin_fma(g.r, bp[k].y, ii); ii = __builtin_fma(g.i, bp[k].x, ii);
                    z[c].r = rr; z[c].i = ii;
                }
                if (ch + 1 < 16 / CH) {
#pragma unroll
                    for (int k = 0; k < CH; k++) { bo[k] = no[k]; bp[k] = np[k]; }
                }
            }
        }
        __syncthreads();

        fft_inverse(z, my, lane, twWr, twWi, twU, P.lanetab);

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
    } else if (lane == 0) {
        out[BIG_N] = acc[0];
    }
}

hipError_t launch_blind_rotate_fft(const BlindRotateFftParams &p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    const size_t lds = (size_t)2 * FFT_LDS_DOUBLES * sizeof(double);
    hipLaunchKernelGGL(blind_rotate_fft_kernel, dim3(p.B), dim3(128), lds, s, p);
    return hipGetLastError();
}

}  // namespace fhs
