// Device helpers shared by the f64-FFT blind-rotation kernels (fft_kernels.hip: 2 wavefronts per ciphertext,
// fft4_kernels.hip: 4 wavefronts per ciphertext).  Both kernels perform the same butterflies on the same values in
// the same order, so one CPU mirror (oracle mode 3) checks either bit for bit.
#pragma once
#include "pbs_kernels.h"

namespace fhs {
namespace fftdev {

#pragma clang fp contract(off)

#include "fft_consts.inc"

constexpr int FM = 1024;                       // complex points
constexpr int FFT_LDS_DOUBLES = 2176;          // per wave: 1088 complex slots (same 17 408 B as the NTT path)

struct cplx { double r, i; };
__device__ __forceinline__ cplx cmul(cplx a, double wr, double wi) {
    cplx t;
    t.r = __builtin_fma(-a.i, wi, a.r * wr);
    t.i = __builtin_fma(a.i, wr, a.r * wi);
    return t;
}
__device__ __forceinline__ uint32_t fft_mod_switch(uint64_t x) { return (uint32_t)(((x + (1ull << 51)) >> 52) & 4095u); }

// Cooley-Tukey butterfly (forward): (a, b) <- (a + w' b, a - w' b), w' = w or i*w (ROT), in 6 fused
// operations: the sum is accumulated straight onto a, the difference is 2a - sum
template <bool ROT> __device__ __forceinline__ void bf_fwd(cplx &a, cplx &b, double wr, double wi) {
    const cplx u = a;
    if (!ROT) {
        a.r = __builtin_fma(-b.i, wi, __builtin_fma(b.r, wr, u.r));
        a.i = __builtin_fma(b.i, wr, __builtin_fma(b.r, wi, u.i));
    } else {
        a.r = __builtin_fma(-b.i, wr, __builtin_fma(-b.r, wi, u.r));
        a.i = __builtin_fma(-b.i, wi, __builtin_fma(b.r, wr, u.i));
    }
    b.r = __builtin_fma(2.0, u.r, -a.r);
    b.i = __builtin_fma(2.0, u.i, -a.i);
}
// Gentleman-Sande butterfly (inverse): (a, b) <- (a + b, (a - b) conj(w')), w' = w or i*w (ROT)
template <bool ROT> __device__ __forceinline__ void bf_inv(cplx &a, cplx &b, double wr, double wi) {
    const cplx u = a, v = b;
    a.r = u.r + v.r; a.i = u.i + v.i;
    cplx d; d.r = u.r - v.r; d.i = u.i - v.i;
    const cplx q = cmul(d, wr, -wi);
    if (!ROT) b = q;
    else { b.r = q.i; b.i = -q.r; }
}

// floor(v) mod 2^64 of a double of any magnitude (v is integral whenever |v| >= 2^52).
// c64 = 2^-64 and c32 = 2^32 arrive in scalar registers the compiler cannot see through: a literal power of two
// becomes v_ldexp_f64, which issues slower than the v_mul_f64 it replaces.
__device__ __forceinline__ uint64_t to_torus(double v, double c64, double c32) {
    const double f = __builtin_amdgcn_fract(v * c64);          // in [0, 1), exact (clamped below 1)
    const double h = f * c32;                                  // exact
    const uint32_t hi = (uint32_t)h;                           // truncation = floor
    const uint32_t lo = (uint32_t)(__builtin_amdgcn_fract(h) * c32);
    return ((uint64_t)hi << 32) | lo;
}

}  // namespace fftdev
}  // namespace fhs
