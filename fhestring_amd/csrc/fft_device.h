// Device helpers shared by the f64-FFT blind-rotation kernels (fft_kernels.hip: 2 wavefronts per ciphertext,
// fft4_kernels.hip: 4 wavefronts per ciphertext).  Both kernels perform the same butterflies on the same values in
// the same order, so one CPU mirror (mode 3 of the CPU oracle) checks either bit for bit.
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

// high word of (flip ? -v : v) - a (64-bit wrapping), flip given as a lane mask: the carry chains run through VCC
// back to back (the compiler's own sequence carries through SGPR pairs and pays a wait state after each borrow)
__device__ __forceinline__ uint32_t rot_sub_hi(uint64_t v, uint64_t a, uint64_t flipmask) {
    const uint32_t vl = (uint32_t)v, vh = (uint32_t)(v >> 32), al = (uint32_t)a, ah = (uint32_t)(a >> 32);
    uint32_t tl, th;
    asm("v_sub_co_u32 %0, vcc, 0, %2\n\t"
        "v_subb_co_u32 %1, vcc, 0, %3, vcc\n\t"
        "v_cndmask_b32 %0, %2, %0, %6\n\t"
        "v_cndmask_b32 %1, %3, %1, %6\n\t"
        "v_sub_co_u32 %0, vcc, %0, %4\n\t"
        "v_subb_co_u32 %1, vcc, %1, %5, vcc"
        : "=&v"(tl), "=&v"(th)
        : "v"(vl), "v"(vh), "v"(al), "v"(ah), "s"(flipmask)
        : "vcc");
    (void)tl;
    return th;
}

// The same difference for operands kept with an offset of -1 (vm = v - 1, am = a - 1, both mod 2^64): with t = vm or
// ~vm (flip) and a borrow-in of 1 where the value is NOT flipped,
//     am - t - borrow = a - v - 1 = ~(v - a)        (not flipped)
//                     = a + v - 1 = ~(-v - a)       (flipped, ~vm = -v),
// i.e. the complement of the wanted difference, exactly, in five operations instead of six -- and the rounding add
// becomes a subtraction: digit = (hi(x) + 0x100) >> 9 = (0xFF - hi(~x)) >> 9.  Returns hi(~x).
__device__ __forceinline__ uint32_t rot_sub_hi_compl(uint64_t vm, uint64_t am, uint64_t keepmask /* lanes NOT flipped */) {
    const uint32_t vl = (uint32_t)vm, vh = (uint32_t)(vm >> 32), al = (uint32_t)am, ah = (uint32_t)(am >> 32);
    uint32_t m, tl, th;
    asm("v_cndmask_b32 %0, -1, 0, %7\n\t"
        "v_xor_b32 %1, %3, %0\n\t"
        "v_xor_b32 %2, %4, %0\n\t"
        "v_subb_co_u32 %1, vcc, %5, %1, %7\n\t"
        "v_subb_co_u32 %2, vcc, %6, %2, vcc"
        : "=&v"(m), "=&v"(tl), "=&v"(th)
        : "v"(vl), "v"(vh), "v"(al), "v"(ah), "s"(keepmask)
        : "vcc");
    (void)m; (void)tl;
    return th;
}

// Torus value (mod 2^64) of t * 2^64, where t is the inverse transform's output: the Fourier-domain key carries
// the factor 2^-64 (beside 1/1024), so the accumulator increment is the fractional part of t.  fract is exact;
// 1 + f puts that fraction into the 52 mantissa bits of a double in [1, 2] (one rounding at 2^-52, i.e. 2^12 torus
// units, far below the noise), which two 32-bit shifts move to the top of the 64-bit word.
__device__ __forceinline__ uint64_t to_torus(double t) {
    const double g = 1.0 + __builtin_amdgcn_fract(t);
    const uint64_t b = __builtin_bit_cast(uint64_t, g);
    typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
    const u32x2 w = __builtin_bit_cast(u32x2, b);
    u32x2 o;
    o.x = w.x << 12;
    o.y = __builtin_amdgcn_alignbit(w.y, w.x, 20);
    return __builtin_bit_cast(uint64_t, o);                    // (b << 12) mod 2^64
}

}  // namespace fftdev
}  // namespace fhs
