// The 1024-point folded negacyclic transform of the 2-wavefront f64-FFT kernels, one wavefront per polynomial (16 complex
// points per lane): register layouts A / B / C, the two LDS transposes per transform, the per-lane twiddle bases.  Shared
// by fft_kernels.hip (classic blind rotation) and fftmb_kernels.hip (two key bits per external product); see the
// header comment of fft_kernels.hip for the mapping.
#pragma once
#include "fft_device.h"

namespace fhs {
namespace fftdev {

#pragma clang fp contract(off)

// Hook called by a stage right after the butterfly on registers (a, b): the transposes use it to send finished
// points to LDS while the next butterflies run (the store path needs ~13 cycles per 16-byte wave store, 16 of them in
// one burst would stall the wave); the scheduling barrier lets arithmetic move across it but pins the stores.
struct no_hook { __device__ __forceinline__ void operator()(int, int) const {} };
struct store_hook {
    cplx *base; const cplx *z; int m0, m1;               // slot of register r = m0 * r + (r >> 2) * m1
    __device__ __forceinline__ void operator()(int a, int b) const {
        base[m0 * a + (a >> 2) * m1] = z[a];
        base[m0 * b + (b >> 2) * m1] = z[b];
        __builtin_amdgcn_sched_barrier(0x7);              // ALU may cross, memory operations may not
    }
};

// Hook of the LAST forward stage: as a butterfly finishes two points of the transform they are (1) published for the
// partner wavefront ([c][lane] order), (2) multiplied by this wavefront's own key row (the first half of the pointwise
// product: rr = fr*kx - fi*ky, ii = fr*ky + fi*kx, same operations in the same order as when the whole product ran
// after the barrier), and (3) the key slot they used is refilled: own-row points 8..15 first, then the partner row's
// points 0..7, which are only needed after the workgroup barrier -- their latency hides behind the rest of this stage
// and the barrier wait.  One 8-point window (32 VGPRs) serves both rows because the two halves no longer overlap.
typedef double __attribute__((ext_vector_type(2))) double2_t;
struct publish_mul_hook {
    cplx *pub; cplx *z; double2_t *kb; const double2_t *b_own, *b_par;
    __device__ __forceinline__ void one(int p) const {
        pub[64 * p] = z[p];
        const double fr = z[p].r, fi = z[p].i;
        const double2_t k = kb[p & 7];
        double rr = fr * k.x; rr = __builtin_fma(-fi, k.y, rr);
        double ii = fr * k.y; ii = __builtin_fma(fi, k.x, ii);
        z[p].r = rr; z[p].i = ii;
        kb[p & 7] = p < 8 ? b_own[(p + 8) * 64] : b_par[(p - 8) * 64];
    }
    __device__ __forceinline__ void operator()(int a, int b) const {
        one(a);
        one(b);
        __builtin_amdgcn_sched_barrier(0x7);              // ALU may cross, memory operations may not
    }
};

// the 4 lane-uniform stages of layout A (twiddles are scalar immediates)
template <bool INV, class Hook> __device__ __forceinline__ void stages_uniform(cplx (&z)[16], const Hook &hook) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int T = INV ? (1 << s) : (8 >> s);
        const int m = 8 / T;
#pragma unroll
        for (int i = 0; i < m; i++) {
            const int k = m == 1 ? 1 : ((m + i) & ~1);     // W[1] for m = 1, else the even index
            const double wr = FW_RE[k], wi = FW_IM[k];
#pragma unroll
            for (int r = 2 * i * T; r < 2 * i * T + T; r++) {
                if (i & 1) { if (INV) bf_inv<true>(z[r], z[r + T], wr, wi); else bf_fwd<true>(z[r], z[r + T], wr, wi); }
                else       { if (INV) bf_inv<false>(z[r], z[r + T], wr, wi); else bf_fwd<false>(z[r], z[r + T], wr, wi); }
                if (s == 3) hook(r, r + T);
            }
        }
    }
}

// one in-lane stage of layout B or C: register distance TAU, G = 8/TAU twiddle groups,
// twiddle of group g = per-lane base * U_G[g] (g even), rotated by i for odd g
template <bool INV, int TAU, class Hook = no_hook>
__device__ __forceinline__ void stage_lane(cplx (&z)[16], double br, double bi, const Hook &hook = Hook()) {
    constexpr int G = 8 / TAU;
#pragma unroll
    for (int g = 0; g < G; g += 2) {
        double wr = br, wi = bi;
        if (g) {
            // U_G[g] = exp(i*pi*bitrev(g)/G): G=4: g=2 -> pi/4;  G=8: g=2 -> pi/4, g=4 -> pi/8, g=6 -> 3pi/8
            const int u = (G == 8 && g == 4) ? 1 : (G == 8 && g == 6) ? 2 : 0;
            cplx b; b.r = br; b.i = bi;
            const cplx w = cmul(b, FU_RE[u], FU_IM[u]);
            wr = w.r; wi = w.i;
        }
#pragma unroll
        for (int c = 2 * g * TAU; c < 2 * g * TAU + TAU; c++) {
            if (INV) bf_inv<false>(z[c], z[c + TAU], wr, wi); else bf_fwd<false>(z[c], z[c + TAU], wr, wi);
            hook(c, c + TAU);
        }
        if (G > 1) {
#pragma unroll
            for (int c = 2 * (g + 1) * TAU; c < 2 * (g + 1) * TAU + TAU; c++) {
                if (INV) bf_inv<true>(z[c], z[c + TAU], wr, wi); else bf_fwd<true>(z[c], z[c + TAU], wr, wi);
                hook(c, c + TAU);
            }
        }
    }
}

// LDS slots (16-byte complex): point n lives at n + (n >> 4)
//   layout A: lane + (lane >> 4) + 68 r          layout B: 68 hi + lo + 4 rho + (rho >> 2)        layout C: 17 lane + c
__device__ __forceinline__ cplx *slotA(double *lds, int lane) { return reinterpret_cast<cplx *>(lds) + lane + (lane >> 4); }
__device__ __forceinline__ cplx *slotB(double *lds, int lane) { return reinterpret_cast<cplx *>(lds) + 68 * (lane >> 2) + (lane & 3); }
__device__ __forceinline__ cplx *slotC(double *lds, int lane) { return reinterpret_cast<cplx *>(lds) + 17 * lane; }

// per-lane twiddle bases: rows (re, im) x {B: G=1,2,4,8; C: G=4,8}
struct LaneTw { double re[6], im[6]; };
__device__ __forceinline__ void load_lane_tw(LaneTw &t, const double *__restrict__ lanetab, int lane) {
    // global address space spelled out: after the opaque asm in the caller the pointer would otherwise be generic
    // and the loads flat (which also count on the LDS counter)
    typedef const __attribute__((address_space(1))) double *gptr_t;
    gptr_t g = (gptr_t)lanetab;
#pragma unroll
    for (int k = 0; k < 6; k++) { t.re[k] = g[(2 * k) * 64 + lane]; t.im[k] = g[(2 * k + 1) * 64 + lane]; }
}

// forward: z[r] = point (lane + 64 r)  ->  z[c] = value at array index 16*lane + c
__device__ __forceinline__ void fft_forward(cplx (&z)[16], double *lds, int lane, const LaneTw &tw) {
    stages_uniform<false>(z, store_hook{slotA(lds, lane), z, 68, 0});             // slot A of register r: 68 r
    __builtin_amdgcn_wave_barrier();
    {
        const cplx *rd = slotB(lds, lane);
#pragma unroll
        for (int q = 0; q < 16; q++) {                    // in the order stage t = 32 pairs them: (p, p + 8)
            const int p = (q >> 1) + 8 * (q & 1);
            z[p] = rd[4 * p + (p >> 2)];
        }
    }
    __builtin_amdgcn_wave_barrier();
    stage_lane<false, 8>(z, tw.re[0], tw.im[0]);
    stage_lane<false, 4>(z, tw.re[1], tw.im[1]);
    stage_lane<false, 2>(z, tw.re[2], tw.im[2]);
    stage_lane<false, 1>(z, tw.re[3], tw.im[3], store_hook{slotB(lds, lane), z, 4, 1});   // slot B: 4 p + (p >> 2)
    __builtin_amdgcn_wave_barrier();
    {
        const cplx *rd = slotC(lds, lane);
#pragma unroll
        for (int q = 0; q < 16; q++) {                    // in the order stage t = 2 pairs them: (c, c + 2)
            const int c = (q & ~3) + ((q & 2) >> 1) + 2 * (q & 1);
            z[c] = rd[c];
        }
    }
    __builtin_amdgcn_wave_barrier();
    stage_lane<false, 2>(z, tw.re[4], tw.im[4]);
}
// last forward stage; every finished point goes through the hook (publication + own-row half of the pointwise product)
template <class Hook>
__device__ __forceinline__ void fft_forward_last(cplx (&z)[16], const LaneTw &tw, const Hook &hook) {
    stage_lane<false, 1>(z, tw.re[5], tw.im[5], hook);
}

// inverse (unscaled): z[c] at array index 16*lane + c  ->  z[r] = point (lane + 64 r)
__device__ __forceinline__ void fft_inverse(cplx (&z)[16], double *lds, int lane, const LaneTw &tw) {
    _Pragma("unroll") for (int c = 0; c < 16; c += 2) bf_triv(z[c], z[c + 1]);
    { const store_hook hk{slotC(lds, lane), z, 1, 0};
      _Pragma("unroll") for (int c = 0; c < 16; c += 4) { bf_triv(z[c], z[c + 2]); hk(c, c + 2); bf_triv(z[c + 1], z[c + 3]); hk(c + 1, c + 3); } }
    __builtin_amdgcn_wave_barrier();
    {
        const cplx *rd = slotB(lds, lane);
#pragma unroll
        for (int p = 0; p < 16; p++) z[p] = rd[4 * p + (p >> 2)];
    }
    __builtin_amdgcn_wave_barrier();
    stage_lane<true, 1>(z, tw.re[3], tw.im[3]);
    stage_lane<true, 2>(z, tw.re[2], tw.im[2]);
    stage_lane<true, 4>(z, tw.re[1], tw.im[1]);
    stage_lane<true, 8>(z, tw.re[0], tw.im[0], store_hook{slotB(lds, lane), z, 4, 1});
    __builtin_amdgcn_wave_barrier();
    {
        const cplx *rd = slotA(lds, lane);
#pragma unroll
        for (int r = 0; r < 16; r++) z[r] = rd[68 * r];
    }
    __builtin_amdgcn_wave_barrier();
    stages_uniform<true>(z, no_hook());
}

}  // namespace fftdev
}  // namespace fhs
