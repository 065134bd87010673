// EXPERIMENT, NOT BUILT (round 3): 8 wavefronts per ciphertext.  Bit-exact against oracle mode 3 on the first run (B = 1, 5, 32,
// 256), half the instruction slots per wavefront of the 4-wavefront kernel (650 vs 1 350) -- and NO faster: 4.0 ms per narrow
// level against 3.8-3.9 ms (gpurun_out/r3m).  What a lone ciphertext waits for is the chain of dependent LDS round trips and
// workgroup barriers (here 10 in-wave transposes + 5 exchanges + 6 barriers per iteration), not instruction issue; the
// 4-wavefront kernel is the sweet spot (2 wavefronts: 7.0 us per iteration, 4: 5.2, 8: 5.3).  Kept for the record.

// f64-FFT blind rotation with 8 wavefronts per ciphertext (gfx950 only): the latency kernel.
//
// A dependency level narrower than one workgroup per CU pays the latency of ONE bootstrap, and that latency is the
// number of instructions one wavefront has to issue: a lone wave issues one instruction every ~6 cycles whatever the
// opcode (tools/ubench_clock.hip), so the 1 350 instruction slots per iteration of the 4-wavefront kernel are ~8 k of
// its 10.9 k cycles.  Here each GLWE polynomial is spread over FOUR wavefronts (4 complex points and 8 accumulator words
// per lane): ~800 slots per wavefront and iteration, two wavefronts per SIMD of the one CU a ciphertext occupies.
// Same arithmetic as fft_kernels.hip / fft4_kernels.hip -- the same radix-2 butterflies on the same values with the same
// effective twiddles (fft_tables.cpp: weff), so the one CPU mirror (oracle mode 3) checks this kernel bit for bit too.
//
//   wave w = 4 j + q: polynomial j, quarter q = bits 9..8 of the point index n.  The two coarsest radix-2 stages
//   (t = 512: q <-> q ^ 2, t = 256: q <-> q ^ 1) go through exchanges in LDS; the other 8 stages are a 256-point
//   transform inside one wavefront in four register layouts (2 index bits in the register number each):
//     A8  n' = lane + 64 r                                 stages t = 128, 64  (wave-uniform twiddles)
//     B8  n' = 64 (lane >> 4) + 16 r + (lane & 15)         stages t = 32, 16
//     C8  n' = 16 (lane >> 2) + 4 r + (lane & 3)           stages t = 8, 4
//     D8  n' = 4 lane + r                                  stages t = 2, 1
//   with three in-wave transposes through the wavefront's private 4.25 KB of LDS.
// LDS per workgroup (one workgroup per CU): accumulator staging 2 x 16.5 KB | exchange X1 (t = 512, also the published
// transform) | exchange X2 (t = 256) | private transposes, 8 x 4.25 KB each = 135 KB.  Every exchange has an area of
// its own, rewritten only after a later barrier that all its readers have passed: 6 workgroup barriers per iteration
// (staged accumulator, forward t = 512, forward t = 256, published transform, inverse t = 256, inverse t = 512).
#include "fft_device.h"

namespace fhs {

#pragma clang fp contract(off)

namespace {
using namespace fftdev;

constexpr int F8_WAVE_BYTES = 4352;                 // 272 slots of 16 B per wavefront
constexpr int F8_STAGE_BYTES = 2 * 2112 * 8;        // per polynomial: 64 words (row 31 again) + 2048 words
constexpr int F8_LDS_BYTES = F8_STAGE_BYTES + 3 * 8 * F8_WAVE_BYTES;
__device__ __forceinline__ int qslot(int n) { return n + (n >> 4); }      // n < 256 -> < 272

typedef const __attribute__((address_space(1))) double *gd8_t;
struct tw8 { double r, i; };
__device__ __forceinline__ tw8 ld_tw8(gd8_t weff, int idx) {
    typedef double __attribute__((ext_vector_type(2))) d2;
    const d2 v = *reinterpret_cast<const __attribute__((address_space(1))) d2 *>(weff + 2 * idx);
    tw8 t; t.r = v.x; t.i = v.y;
    return t;
}

// two in-lane stages on 4 registers: distance 2 with one twiddle, distance 1 with a twiddle per pair
__device__ __forceinline__ void fwd2(cplx (&z)[4], tw8 a, tw8 b0, tw8 b1) {
    bf_fwd<false>(z[0], z[2], a.r, a.i);
    bf_fwd<false>(z[1], z[3], a.r, a.i);
    bf_fwd<false>(z[0], z[1], b0.r, b0.i);
    bf_fwd<false>(z[2], z[3], b1.r, b1.i);
}
__device__ __forceinline__ void inv2(cplx (&z)[4], tw8 a, tw8 b0, tw8 b1) {
    bf_inv<false>(z[0], z[1], b0.r, b0.i);
    bf_inv<false>(z[2], z[3], b1.r, b1.i);
    bf_inv<false>(z[0], z[2], a.r, a.i);
    bf_inv<false>(z[1], z[3], a.r, a.i);
}

}  // namespace

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void blind_rotate_fft8_kernel(BlindRotateFftParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ct = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = w >> 2, q = w & 3;
    uint64_t *stage = reinterpret_cast<uint64_t *>(smem + j * (F8_STAGE_BYTES / 2));       // 64 + 2048 words of polynomial j
    char *const x1b = smem + F8_STAGE_BYTES, *const x2b = x1b + 8 * F8_WAVE_BYTES, *const mb = x2b + 8 * F8_WAVE_BYTES;
    cplx *x1 = reinterpret_cast<cplx *>(x1b + w * F8_WAVE_BYTES);                          // t = 512 exchange / published transform
    const cplx *x1p = reinterpret_cast<const cplx *>(x1b + (w ^ 2) * F8_WAVE_BYTES);       // quarter q ^ 2, same polynomial
    const cplx *pub = reinterpret_cast<const cplx *>(x1b + (w ^ 4) * F8_WAVE_BYTES);       // same quarter, other polynomial
    cplx *x2 = reinterpret_cast<cplx *>(x2b + w * F8_WAVE_BYTES);                          // t = 256 exchange
    const cplx *x2p = reinterpret_cast<const cplx *>(x2b + (w ^ 1) * F8_WAVE_BYTES);       // quarter q ^ 1
    cplx *mine = reinterpret_cast<cplx *>(mb + w * F8_WAVE_BYTES);                         // private transposes

    const uint64_t *ks = P.ks + (size_t)ct * SMALL_CT;
    const gd8_t weff = (gd8_t)P.weff;

    // twiddles, resident for the whole bootstrap: wave-uniform ones in scalar registers, 9 per-lane ones (36 VGPRs)
    const tw8 c512 = ld_tw8(weff, 1), c256 = ld_tw8(weff, 2 + (q >> 1));
    const tw8 a128 = ld_tw8(weff, 4 + q), a64a = ld_tw8(weff, 8 + 2 * q), a64b = ld_tw8(weff, 9 + 2 * q);
    const int la = lane >> 4, lc = lane >> 2;
    const tw8 b32 = ld_tw8(weff, 16 + 4 * q + la), b16a = ld_tw8(weff, 32 + 8 * q + 2 * la), b16b = ld_tw8(weff, 33 + 8 * q + 2 * la);
    const tw8 c8 = ld_tw8(weff, 64 + 16 * q + lc), c4a = ld_tw8(weff, 128 + 32 * q + 2 * lc), c4b = ld_tw8(weff, 129 + 32 * q + 2 * lc);
    const tw8 d2 = ld_tw8(weff, 256 + 64 * q + lane), d1a = ld_tw8(weff, 512 + 128 * q + 2 * lane), d1b = ld_tw8(weff, 513 + 128 * q + 2 * lane);

    // acc[r]: coefficient k(r) = 256 q + lane + 64 (r & 3) + 1024 (r >> 2) of polynomial j
    const uint32_t k0 = 256 * q + lane;
    uint64_t acc[8];
    {
        const uint32_t b = fft_mod_switch(ks[LWE_N]);
        const uint32_t a = (2 * POLY_N - b) & (2 * POLY_N - 1);
        const uint32_t s = a & (POLY_N - 1);
        const bool neg = a >= POLY_N;
        const uint64_t *lut = P.luts + (size_t)P.lut_idx[ct] * POLY_N;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint64_t v = 0;
            if (j == 1) {
                const uint32_t n = k0 + 64 * (r & 3) + 1024 * (r >> 2);
                v = lut[(n - s) & (POLY_N - 1)];
                if ((n < s) != neg) v = (uint64_t)0 - v;
            }
            acc[r] = v;
        }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) stage[64 + k0 + 64 * (r & 3) + 1024 * (r >> 2)] = acc[r];
    if (q == 3) stage[lane] = acc[7];                 // row 31 again in front of row 0 (see the rotated read)

    // LDS addresses of the three in-wave transposes (slot of register r = base + stride * r, padded with qslot)
    const int sA = qslot(lane), sB = qslot(64 * la + (lane & 15)), sC = qslot(16 * lc + (lane & 3)), sD = qslot(4 * lane);

    // key: [i][row][col][16][64 lanes] complex; point n = 256 q + 4 lane + r of layout D8 sits at
    // [c16 = 4 (lane & 3) + r][L = 16 q + (lane >> 2)]
    typedef double __attribute__((ext_vector_type(2))) double2_t;
    const size_t koff = (size_t)(4 * (lane & 3)) * 64 + 16 * q + lc;

    uint64_t ks_next = ks[0];                         // requested one iteration ahead (ks[LWE_N] is a valid address)
    for (int i = 0; i < LWE_N; i++) {
        asm volatile("" : "+v"(ks_next));
        const uint32_t a = __builtin_amdgcn_readfirstlane(fft_mod_switch(ks_next));
        __builtin_amdgcn_sched_barrier(0);
        ks_next = ks[i + 1];
        if (a == 0) continue;
        const uint32_t s = a & (POLY_N - 1);
        const bool neg = a >= POLY_N;

        // key rows of this iteration, all 4 points of both rows (32 VGPRs): in flight across the whole forward transform
        const double2_t *b_own = reinterpret_cast<const double2_t *>(P.bsk_fft) + ((((size_t)i * 2 + j) * 2 + j)) * FM + koff;
        const double2_t *b_par = reinterpret_cast<const double2_t *>(P.bsk_fft) + ((((size_t)i * 2 + (1 - j)) * 2 + j)) * FM + koff;
        double2_t bo[4], bp[4];
#pragma unroll
        for (int c = 0; c < 4; c++) { bo[c] = b_own[c * 64]; bp[c] = b_par[c * 64]; }

        // ---- rotate, subtract, decompose: z[r] = digit(k(r)) + i digit(k(r) + 1024) -------------------------------
        __syncthreads();                              // B1: staged accumulator of all four quarters visible
        cplx z[4];
        {
            const uint32_t sl = s & 63, sh = s >> 6;
            const bool borrow = (uint32_t)lane < sl;
            const uint64_t negmask = neg ? ~0ull : 0ull;
            const uint64_t *vbase = stage + (((uint32_t)lane - sl) & 63) + (borrow ? 0 : 64);
            const int32_t thr = (int32_t)s - lane;    // the index wrapped where 64 row + lane < s
            uint64_t v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = vbase[64 * ((4 * q + (r & 3) + 16 * (r >> 2) - sh) & 31)];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int row = 4 * q + (r & 3) + 16 * (r >> 2);
                const uint64_t wrapmask = __builtin_amdgcn_ballot_w64(64 * row < thr);
                const uint32_t dhi = rot_sub_hi(v[r], acc[r], wrapmask ^ negmask);
                const int32_t dig = (int32_t)(dhi + 0x100u) >> 9;
                if (r < 4) z[r].r = (double)dig; else z[r - 4].i = (double)dig;
            }
        }

        // ---- forward transform -------------------------------------------------------------------------------
        {   // t = 512 across quarters q and q ^ 2: (a, b) = (lower, upper) point, every wave keeps its own output
#pragma unroll
            for (int r = 0; r < 4; r++) x1[qslot(lane + 64 * r)] = z[r];
            __syncthreads();                          // B2
            cplx o[4];
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = x1p[qslot(lane + 64 * r)];
            if ((q & 2) == 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) bf_fwd<false>(z[r], o[r], c512.r, c512.i);
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) bf_fwd<false>(o[r], z[r], c512.r, c512.i);
            }
        }
        {   // t = 256 across quarters q and q ^ 1
#pragma unroll
            for (int r = 0; r < 4; r++) x2[qslot(lane + 64 * r)] = z[r];
            __syncthreads();                          // B3
            cplx o[4];
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = x2p[qslot(lane + 64 * r)];
            if ((q & 1) == 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) bf_fwd<false>(z[r], o[r], c256.r, c256.i);
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) bf_fwd<false>(o[r], z[r], c256.r, c256.i);
            }
        }
        fwd2(z, a128, a64a, a64b);                    // layout A8: t = 128, 64
#pragma unroll
        for (int r = 0; r < 4; r++) mine[sA + 68 * r] = z[r];                 // qslot(lane + 64 r) = qslot(lane) + 68 r
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; r++) z[r] = mine[sB + 17 * r];                 // qslot(64 a + 16 r + b) = qslot(64 a + b) + 17 r
        __builtin_amdgcn_wave_barrier();
        fwd2(z, b32, b16a, b16b);                     // layout B8: t = 32, 16
#pragma unroll
        for (int r = 0; r < 4; r++) mine[sB + 17 * r] = z[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; r++) z[r] = mine[sC + 4 * r];                  // qslot(16 a' + 4 r + b') = qslot(16 a' + b') + 4 r
        __builtin_amdgcn_wave_barrier();
        fwd2(z, c8, c4a, c4b);                        // layout C8: t = 8, 4
#pragma unroll
        for (int r = 0; r < 4; r++) mine[sC + 4 * r] = z[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; r++) z[r] = mine[sD + r];                      // qslot(4 lane + r) = qslot(4 lane) + r
        __builtin_amdgcn_wave_barrier();
        fwd2(z, d2, d1a, d1b);                        // layout D8: t = 2, 1

        // ---- publish, pointwise multiply-accumulate with GGSW_i -------------------------------------------------
#pragma unroll
        for (int c = 0; c < 4; c++) x1[c * 64 + lane] = z[c];
        __syncthreads();                              // B4
        {
            cplx g[4];
#pragma unroll
            for (int c = 0; c < 4; c++) g[c] = pub[c * 64 + lane];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const double fr = z[c].r, fi = z[c].i;
                double rr = fr * bo[c].x; rr = __builtin_fma(-fi, bo[c].y, rr);
                rr = __builtin_fma(g[c].r, bp[c].x, rr); rr = __builtin_fma(-g[c].i, bp[c].y, rr);
                double ii = fr * bo[c].y; ii = __builtin_fma(fi, bo[c].x, ii);
                ii = __builtin_fma(g[c].r, bp[c].y, ii); ii = __builtin_fma(g[c].i, bp[c].x, ii);
                z[c].r = rr; z[c].i = ii;
            }
        }

        // ---- inverse transform -------------------------------------------------------------------------------
        inv2(z, d2, d1a, d1b);
#pragma unroll
        for (int r = 0; r < 4; r++) mine[sD + r] = z[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; r++) z[r] = mine[sC + 4 * r];
        __builtin_amdgcn_wave_barrier();
        inv2(z, c8, c4a, c4b);
#pragma unroll
        for (int r = 0; r < 4; r++) mine[sC + 4 * r] = z[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; r++) z[r] = mine[sB + 17 * r];
        __builtin_amdgcn_wave_barrier();
        inv2(z, b32, b16a, b16b);
#pragma unroll
        for (int r = 0; r < 4; r++) mine[sB + 17 * r] = z[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; r++) z[r] = mine[sA + 68 * r];
        __builtin_amdgcn_wave_barrier();
        inv2(z, a128, a64a, a64b);
        {   // t = 256 across quarters q and q ^ 1
#pragma unroll
            for (int r = 0; r < 4; r++) x2[qslot(lane + 64 * r)] = z[r];
            __syncthreads();                          // B5
            cplx o[4];
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = x2p[qslot(lane + 64 * r)];
            if ((q & 1) == 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) bf_inv<false>(z[r], o[r], c256.r, c256.i);
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) bf_inv<false>(o[r], z[r], c256.r, c256.i);
            }
        }
        {   // t = 512 across quarters q and q ^ 2
#pragma unroll
            for (int r = 0; r < 4; r++) x1[qslot(lane + 64 * r)] = z[r];
            __syncthreads();                          // B6
            cplx o[4];
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = x1p[qslot(lane + 64 * r)];
            if ((q & 2) == 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) bf_inv<false>(z[r], o[r], c512.r, c512.i);
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) bf_inv<false>(o[r], z[r], c512.r, c512.i);
            }
        }

        // ---- back to the torus, update and restage the accumulator -------------------------------------------
#pragma unroll
        for (int r = 0; r < 4; r++) {
            acc[r] += to_torus(z[r].r);
            stage[64 + k0 + 64 * r] = acc[r];
            acc[r + 4] += to_torus(z[r].i);
            stage[64 + k0 + 64 * r + 1024] = acc[r + 4];
            if (r == 3 && q == 3) stage[lane] = acc[7];
        }
    }

    uint64_t *out = P.out_ptrs ? P.out_ptrs[ct] : P.out + (size_t)ct * BIG_CT;
    if (j == 0) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int n = k0 + 64 * (r & 3) + 1024 * (r >> 2);
            if (n == 0) out[0] = acc[r];
            else out[POLY_N - n] = (uint64_t)0 - acc[r];
        }
    } else if (k0 == 0) {
        out[BIG_N] = acc[0];
    }
}

// 135 KB of dynamic LDS: a per-DEVICE opt-in, set by Context::init with the device current
hipError_t prepare_device_for_fft8() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(blind_rotate_fft8_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS_BYTES);
}

hipError_t launch_blind_rotate_fft8(const BlindRotateFftParams &p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(blind_rotate_fft8_kernel, dim3(p.B), dim3(512), F8_LDS_BYTES, s, p);
    return hipGetLastError();
}

}  // namespace fhs
