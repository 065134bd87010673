// f64-FFT blind rotation with 4 wavefronts per ciphertext (gfx950 only).
//
// Same arithmetic as fft_kernels.hip (same butterflies on the same values in the same order: the CPU mirror,
// mode 3 of the CPU oracle, checks both bit for bit) but each GLWE polynomial is spread over a PAIR of wavefronts, 8 complex
// points and 16 accumulator words per lane.  Half the work per wavefront shortens the latency of one PBS (4.3 ms
// instead of 5.2 ms), which is what a narrow dependency level pays; Context::blind_rotate uses this kernel for batches
// of at most fft4_max_batch (512) ciphertexts, where at most 2 workgroups share a CU.
//
//   wave w = 2*j + h: polynomial j, half h = bit 9 of the point index n.  The first (last, for the inverse) radix-2
//   stage t = 512 pairs the two halves and is done through one exchange in LDS; the other 9 stages are a 512-point
//   transform inside one wavefront in three register layouts (3 index bits in the register number each):
//     A'  n' = lane + 64 r                      stages t = 256, 128, 64  (wave-uniform twiddles)
//     B'  n' = 64 (lane >> 3) + 8 r + (lane & 7)  stages t = 32, 16, 8
//     C'  n' = 8 lane + r                       stages t = 4, 2, 1
//   with two in-wave transposes through the wavefront's private 8.7 KB of LDS.
// Per-lane twiddles come from the effective table built by fft_tables.cpp (weff): the values the 2-wavefront kernel
// derives on the fly, so nothing is re-rounded differently here.
#include "fft_device.h"

namespace fhs {

#pragma clang fp contract(off)

namespace {
using namespace fftdev;

// One pad slot after every 8 points: layout A' puts consecutive lanes on consecutive slots, B' (72 a + 9 r + b) too, and
// C' becomes 9 lane + c -- every group of 8 lanes covers all banks with its 16-byte accesses.  (With a pad every 16
// points C' was 8 lane + c + (lane >> 1): two lanes per bank group; 22 % of the LDS-active cycles were bank conflicts.)
constexpr int F4_WAVE_BYTES = 9216;                 // 576 slots of 16 B per wavefront (4 x 9216 = 36 864 B per WG)
__device__ __forceinline__ int pslot(int n) { return n + (n >> 3); }     // n < 512 -> < 576

typedef const __attribute__((address_space(1))) double *gdptr_t;
struct tw_t { double r, i; };
__device__ __forceinline__ tw_t ld_tw(gdptr_t weff, int idx) {
    typedef double __attribute__((ext_vector_type(2))) d2;
    const d2 v = *reinterpret_cast<const __attribute__((address_space(1))) d2 *>(weff + 2 * idx);
    tw_t t; t.r = v.x; t.i = v.y;
    return t;
}

// one in-lane stage on 8 registers: distance TAU, groups of 2*TAU registers; group g uses twA (g = 0), i*twA (1),
// twB (2), i*twB (3)
template <bool INV, int TAU> __device__ __forceinline__ void stage8(cplx (&z)[8], tw_t a, tw_t b) {
    constexpr int G = 4 / TAU;
#pragma unroll
    for (int g = 0; g < G; g++) {
        const tw_t w = g < 2 ? a : b;
#pragma unroll
        for (int c = 2 * g * TAU; c < 2 * g * TAU + TAU; c++) {
            if (g & 1) { if (INV) bf_inv<true>(z[c], z[c + TAU], w.r, w.i); else bf_fwd<true>(z[c], z[c + TAU], w.r, w.i); }
            else       { if (INV) bf_inv<false>(z[c], z[c + TAU], w.r, w.i); else bf_fwd<false>(z[c], z[c + TAU], w.r, w.i); }
        }
    }
}
// the three wave-uniform stages of layout A' (every group has its own explicit twiddle, no rotation), in two parts: the
// step next to the exchange or the transpose (forward: the LAST one, t = 64; inverse: the LAST one, t = 256) is run by the
// caller through pipeline4 together with its stores
template <bool INV> __device__ __forceinline__ void stagesA_head(cplx (&z)[8], tw_t w2, const tw_t (&w4)[2], const tw_t (&w8)[4]) {
    if (!INV) {
#pragma unroll
        for (int r = 0; r < 4; r++) bf_fwd<false>(z[r], z[r + 4], w2.r, w2.i);
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int r = 4 * g; r < 4 * g + 2; r++) bf_fwd<false>(z[r], z[r + 2], w4[g].r, w4[g].i);
    } else {
#pragma unroll
        for (int g = 0; g < 4; g++) bf_inv<false>(z[2 * g], z[2 * g + 1], w8[g].r, w8[g].i);
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int r = 4 * g; r < 4 * g + 2; r++) bf_inv<false>(z[r], z[r + 2], w4[g].r, w4[g].i);
    }
}

// Round 6 (profiles/r06_fft4_timeline_before.txt): a lone ciphertext overlaps nothing -- its FP64 issue (4.5 k cycles per
// iteration), its LDS traffic (4.5 k: ds_write_b128 moves 79 B/clk per CU, the four wavefronts store in the same phases),
// the tail of the key loads (1.4 k) and the write -> barrier -> read exposures (2.7 k) simply add up.  Two remedies, both
// pure instruction ORDER (same butterflies on the same values: the same CPU mirror, bit for bit):
//   FFT4_OVERLAP  the stores of every exchange / transpose are issued from inside the stage that produces their data:
//                 the two registers a butterfly finishes are stored while the next butterfly computes (pipeline4), and
//                 the forward exchange is written while the imaginary digits are still being decomposed;
//   FFT4_HB_WIDE  the one-workgroup-per-CU kernel requests ALL key rows of the iteration (8 chunks per row) before the
//                 forward transform instead of keeping 2 in flight inside the pointwise product (one wave per SIMD may
//                 use the whole 512-register file).
#ifndef FFT4_OVERLAP
#define FFT4_OVERLAP 2       // 2: one butterfly per step (measured best, 2.94 ms per 64-row level); 1: two; 0: off
#endif
#ifndef FFT4_HB_WIDE
#define FFT4_HB_WIDE 8
#endif
//   FFT4_KEY_SPREAD  (with FFT4_HB_WIDE 8) the 16 key loads of a wavefront-iteration are requested two at a time at eight
//                 points of the forward transform instead of back to back: a vector-memory instruction holds its
//                 wavefront while the texture-address unit takes its 64 addresses, and the four wavefronts of the
//                 workgroup issue in the same phase (nokey ablation: 0.44 ms per level; requesting all 16 at once, however
//                 early, gave none of it back)
#ifndef FFT4_KEY_SPREAD
#define FFT4_KEY_SPREAD 1
#endif
//   FFT4_XCHG_I32  (one workgroup per CU) the FORWARD cross-half exchange carries the digits as two 32-bit integers per point
//                 (8 B: ds_write_b64 / ds_read_b64) instead of two doubles (16 B); the partner converts them -- the same
//                 exact values, half the bytes through the LDS store path.
#ifndef FFT4_XCHG_I32
#define FFT4_XCHG_I32 1
#endif
// the two-workgroups-per-CU kernel (257..512 rows) has 256 registers per wavefront: a window of 2 chunks as in round 5.
// (4 chunks, spread the same way, measured 4.95 vs 4.76 ms at 300 rows and 5.13 vs 5.22 ms at 512: inside the noise, not
// adopted -- its second wavefront per SIMD already fills what a held wavefront leaves.)
#ifndef FFT4_HB_NARROW
#define FFT4_HB_NARROW 2
#endif
//   FFT4_REG_T1 / FFT4_REG_T2  (experiments, OFF: measured and not adopted, profiles/r06_fft4_ab_timeline.txt) the private
//                 transposes inside the register file instead of through LDS: v_permlane32_swap / v_permlane16_swap exchange
//                 lane bit 5 / 4 with a register bit, row_ror:8 / row_shr:4 / row_shl:4 moves under bank masks do it for lane
//                 bits 3 / 2, quad_perm moves + selects for bits 1 / 0.  The results are layouts B' / C' exactly (same digest),
//                 but 80-160 cross-lane VALU instructions per transpose take as long as its LDS round trip: A' <-> B' in
//                 registers 2.914 vs 2.923 ms per level, both transposes 3.11 ms.
#ifndef FFT4_REG_T1
#define FFT4_REG_T1 0
#endif
#ifndef FFT4_REG_T2
#define FFT4_REG_T2 0
#endif
// exchanges register-index bit DST (4, 2, 1) of the 8 points with lane bit log2(W): element (register bit 1, lane bit 0)
// <-> (register bit 0, lane bit 1).  An involution: the same call undoes it.
template <int W, int DST> __device__ __forceinline__ void reg_lane_swap(cplx (&z)[8]) {
#pragma unroll
    for (int p = 0; p < 8; p++) {
        if (p & DST) continue;
        double *a[2] = {&z[p].r, &z[p].i}, *b[2] = {&z[p + DST].r, &z[p + DST].i};
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const uint64_t x = __builtin_bit_cast(uint64_t, *a[hh]), y = __builtin_bit_cast(uint64_t, *b[hh]);
            uint32_t xw[2] = {(uint32_t)x, (uint32_t)(x >> 32)}, yw[2] = {(uint32_t)y, (uint32_t)(y >> 32)};
#pragma unroll
            for (int q = 0; q < 2; q++) {
                uint32_t nx, ny;
                if (W == 32) {          // lanes 32..63 of x <-> lanes 0..31 of y
                    auto l = __builtin_amdgcn_permlane32_swap(xw[q], yw[q], false, false); nx = l[0]; ny = l[1];
                } else if (W == 16) {   // odd rows of x <-> even rows of y
                    auto l = __builtin_amdgcn_permlane16_swap(xw[q], yw[q], false, false); nx = l[0]; ny = l[1];
                } else if (W == 8) {    // row_ror:8 = lane ^ 8; banks 2, 3 = lanes with bit 3 set
                    nx = (uint32_t)__builtin_amdgcn_update_dpp((int)xw[q], (int)yw[q], 0x128, 0xF, 0xC, false);
                    ny = (uint32_t)__builtin_amdgcn_update_dpp((int)yw[q], (int)xw[q], 0x128, 0xF, 0x3, false);
                } else if (W == 4) {    // row_shr:4 into banks 1, 3 (lane bit 2 set), row_shl:4 into banks 0, 2
                    nx = (uint32_t)__builtin_amdgcn_update_dpp((int)xw[q], (int)yw[q], 0x114, 0xF, 0xA, false);
                    ny = (uint32_t)__builtin_amdgcn_update_dpp((int)yw[q], (int)xw[q], 0x104, 0xF, 0x5, false);
                } else {                // W = 2 / 1: quad_perm [2,3,0,1] / [1,0,3,2] = lane ^ 2 / lane ^ 1, merged by lane mask
                    const int ctrl = W == 2 ? 0x4E : 0xB1;
                    const uint32_t py = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)yw[q], ctrl, 0xF, 0xF, true);
                    const uint32_t px = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xw[q], ctrl, 0xF, 0xF, true);
                    const bool hi = (threadIdx.x & W) != 0;                         // this lane has the bit set
                    nx = hi ? py : xw[q];
                    ny = hi ? yw[q] : px;
                }
                xw[q] = nx; yw[q] = ny;
            }
            *a[hh] = __builtin_bit_cast(double, ((uint64_t)xw[1] << 32) | xw[0]);
            *b[hh] = __builtin_bit_cast(double, ((uint64_t)yw[1] << 32) | yw[0]);
        }
    }
}
// FFT4_TIMELINE (tools/fft4_timeline.py, timing experiments only; off in the product): every wavefront stamps the shader
// clock (s_memtime) at twelve points of iterations 300..331 into a device array
#ifdef FFT4_TIMELINE
#define FFT4_TL_ITER0 300
#define FFT4_TL_ITERS 32
#define FFT4_TL_STAMPS 12
#define FFT4_TL_MAXB 256
__device__ unsigned long long g_fft4_tl[FFT4_TL_MAXB * 4 * FFT4_TL_ITERS * FFT4_TL_STAMPS];
extern "C" int fhs_exp_fft4_timeline(unsigned long long *out, size_t n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fft4_tl), n * 8);
}
#define FFT4_TL_TOP() const bool tl_on = i >= FFT4_TL_ITER0 && i < FFT4_TL_ITER0 + FFT4_TL_ITERS && blockIdx.x < FFT4_TL_MAXB; \
    unsigned long long *tlp = g_fft4_tl + (((size_t)blockIdx.x * 4 + w) * FFT4_TL_ITERS + (tl_on ? i - FFT4_TL_ITER0 : 0)) * FFT4_TL_STAMPS; \
    FFT4_TL(0)
#define FFT4_TL(k) do { __builtin_amdgcn_sched_barrier(0); if (tl_on) { const unsigned long long t_ = __builtin_readcyclecounter(); \
    if (lane == 0) tlp[k] = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FFT4_TL_TOP() do {} while (0)
#define FFT4_TL(k) do {} while (0)
#endif
// the four butterflies bf(0..3) of a stage and the stores st(k) of the two registers butterfly k finishes
// (OV = false, the kernel with two workgroups per CU: plain order -- a second wavefront per SIMD already fills the gaps, and
// pinning cost 1 % there)
template <bool OV, class BF, class ST> __device__ __forceinline__ void pipeline4(BF bf, ST st) {
    if (OV && FFT4_OVERLAP == 2) {                            // one butterfly per step, two stores at the tail
        bf(0);
        __builtin_amdgcn_sched_barrier(0);
        st(0); bf(1);
        __builtin_amdgcn_sched_barrier(0);
        st(1); bf(2);
        __builtin_amdgcn_sched_barrier(0);
        st(2); bf(3);
        __builtin_amdgcn_sched_barrier(0);
        st(3);
    } else if (OV) {                                          // (FFT4_OVERLAP 1: 3.02 ms)
        bf(0); bf(1);
        __builtin_amdgcn_sched_barrier(0);
        st(0); bf(2);
        __builtin_amdgcn_sched_barrier(0);
        st(1); bf(3);
        __builtin_amdgcn_sched_barrier(0);
        st(2); st(3);
    } else {
        bf(0); bf(1); bf(2); bf(3);
        st(0); st(1); st(2); st(3);
    }
}
// butterfly k of the last in-lane stage of a layout: TAU = 1 pairs registers (2k, 2k + 1) with twiddle a, i a, b, i b
// (stage8<INV, 1>), TAU = 4 pairs (k, k + 4) with twiddle a (stage8<INV, 4>)
template <bool INV, int TAU> __device__ __forceinline__ void last_bf(cplx (&z)[8], int k, tw_t a, tw_t b) {
    if (TAU == 1) {
        const tw_t w = k < 2 ? a : b;
        if (k & 1) { if (INV) bf_inv<true>(z[2 * k], z[2 * k + 1], w.r, w.i); else bf_fwd<true>(z[2 * k], z[2 * k + 1], w.r, w.i); }
        else       { if (INV) bf_inv<false>(z[2 * k], z[2 * k + 1], w.r, w.i); else bf_fwd<false>(z[2 * k], z[2 * k + 1], w.r, w.i); }
    } else {
        if (INV) bf_inv<false>(z[k], z[k + 4], a.r, a.i); else bf_fwd<false>(z[k], z[k + 4], a.r, a.i);
    }
}

// one ciphertext on the 4 wavefronts of the calling workgroup.
// WIDE: four separate LDS areas per workgroup (accumulator staging | cross-half exchange | private transposes | published
// transform; 4 x 36 864 B, one workgroup per CU) instead of one area reused for everything.  Half of the 8 workgroup
// barriers per iteration only kept a reader ahead of the NEXT writer of a shared area; with an area of its own every
// exchange is rewritten only after a later barrier that all its readers have passed, and 4 barriers remain (staged
// accumulator visible, forward cross stage visible, transform published, inverse cross stage visible).  A lone
// ciphertext waits at each of them for the slowest of its four wavefronts.
template <bool WIDE>
__device__ __forceinline__ void fft4_bootstrap(const BlindRotateFftParams &P, char *smem, const int ct) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = w >> 1, h = w & 1;
    constexpr int AREA = 4 * F4_WAVE_BYTES;
    constexpr bool OVERLAP = WIDE && FFT4_OVERLAP;
    constexpr bool EARLY_X = OVERLAP;
    constexpr bool XI32 = EARLY_X && FFT4_XCHG_I32;
    constexpr bool REG_T1 = WIDE && FFT4_REG_T1;
    constexpr bool REG_T2 = WIDE && FFT4_REG_T2;
    char *const xbase = smem + (WIDE ? AREA : 0), *const mbase = smem + (WIDE ? 2 * AREA : 0), *const pbase = smem + (WIDE ? 3 * AREA : 0);
    cplx *mine = reinterpret_cast<cplx *>(mbase + w * F4_WAVE_BYTES);                      // private transposes
    cplx *xmine = reinterpret_cast<cplx *>(xbase + w * F4_WAVE_BYTES);                     // cross-half exchange, own half
    const cplx *pair = reinterpret_cast<const cplx *>(xbase + (w ^ 1) * F4_WAVE_BYTES);    // other half, same polynomial
    typedef int32_t __attribute__((ext_vector_type(2))) int2_t;
    int2_t *xmine_i = reinterpret_cast<int2_t *>(xbase + w * F4_WAVE_BYTES);               // (XI32) point n at slot n, 8 B each
    const int2_t *pair_i = reinterpret_cast<const int2_t *>(xbase + (w ^ 1) * F4_WAVE_BYTES);
    cplx *pmine = reinterpret_cast<cplx *>(pbase + w * F4_WAVE_BYTES);                     // published transform
    const cplx *other = reinterpret_cast<const cplx *>(pbase + (w ^ 2) * F4_WAVE_BYTES);   // same half, other polynomial
    uint64_t *stage = reinterpret_cast<uint64_t *>(smem + j * 2 * F4_WAVE_BYTES);          // 2048 words of polynomial j

    const uint64_t *ks = P.ks + (size_t)ct * SMALL_CT;
    const gdptr_t weff0 = (gdptr_t)P.weff;

    // wave-uniform twiddles of the cross stage and of layout A' (scalar registers)
    const tw_t w1 = ld_tw(weff0, 1), w2 = ld_tw(weff0, 2 + h);
    const tw_t w4[2] = {ld_tw(weff0, 4 + 2 * h), ld_tw(weff0, 5 + 2 * h)};
    const tw_t w8[4] = {ld_tw(weff0, 8 + 4 * h), ld_tw(weff0, 9 + 4 * h), ld_tw(weff0, 10 + 4 * h), ld_tw(weff0, 11 + 4 * h)};
    const int eB = 8 * h + (lane >> 3);             // bits 9..6 of n in layout B'
    const int eC = 64 * h + lane;                   // bits 9..3 of n in layout C'
    // per-lane twiddles, RESIDENT for the whole bootstrap (32 VGPRs): this kernel only runs narrow levels (at most two
    // workgroups per CU, 256 registers per wave either way), where reloading them four times per iteration put four
    // L1 / L2 round trips into the latency of a lone ciphertext
    const tw_t t32 = ld_tw(weff0, 16 + eB), t16 = ld_tw(weff0, 32 + 2 * eB);
    const tw_t t8a = ld_tw(weff0, 64 + 4 * eB), t8b = ld_tw(weff0, 64 + 4 * eB + 2);
    const tw_t t4 = ld_tw(weff0, 128 + eC), t2 = ld_tw(weff0, 256 + 2 * eC);
    const tw_t t1a = ld_tw(weff0, 512 + 4 * eC), t1b = ld_tw(weff0, 512 + 4 * eC + 2);

    // acc[r]: coefficient k(r) = 512 h + lane + 64 (r & 7) + 1024 (r >> 3) of polynomial j, MINUS ONE (the offset makes
    // the rotate-and-subtract exact in five operations, fft_device.h: rot_sub_hi_compl)
    const uint32_t k0 = 512 * h + lane;
    uint64_t acc[16];
    {
        const uint32_t b = fft_mod_switch(ks[LWE_N]);
        const uint32_t a = (2 * POLY_N - b) & (2 * POLY_N - 1);
        const uint32_t s = a & (POLY_N - 1);
        const bool neg = a >= POLY_N;
        const uint64_t *lut = P.luts + (size_t)P.lut_idx[ct] * POLY_N;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            uint64_t v = 0;
            if (j == 1) {
                const uint32_t n = k0 + 64 * (r & 7) + 1024 * (r >> 3);
                v = lut[(n - s) & (POLY_N - 1)];
                if ((n < s) != neg) v = (uint64_t)0 - v;
            }
            acc[r] = v - 1;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) stage[64 + k0 + 64 * (r & 7) + 1024 * (r >> 3)] = acc[r];
    if (h == 1) stage[lane] = acc[15];                // row 31 again in front of row 0 (see the rotated read)
    else stage[64 + 2048 + lane] = acc[0];            // and row 0 again behind row 31: rows are read in pairs

    // the mask element of the NEXT iteration is requested one iteration ahead (ks[LWE_N], the body, is a valid address):
    // for a lone ciphertext its global-memory round trip at the top of every iteration was pure latency
    uint64_t ks_next = ks[0];
    for (int i = 0; i < LWE_N; i++) {
        asm volatile("" : "+v"(ks_next));
        const uint32_t a = __builtin_amdgcn_readfirstlane(fft_mod_switch(ks_next));
        __builtin_amdgcn_sched_barrier(0);
        ks_next = ks[i + 1];
        if (a == 0) continue;
        const uint32_t s = a & (POLY_N - 1);
        const bool neg = a >= POLY_N;

        FFT4_TL_TOP();
        // ---- rotate, subtract, decompose: z[r] = digit(k(r)) + i digit(k(r) + 1024) -------------------------------
        __syncthreads();                              // staged accumulator of both halves visible
        FFT4_TL(1);
        cplx z[8];
        int32_t dre[8];                               // (XI32) the real digits, kept as integers until their point goes out
        // as in fft_kernels.hip: lane rotation by s mod 64 (per-lane base), row rotation by s div 64 (scalar offset per
        // register; this wave's register r is row 8 h + (r & 7) + 16 (r >> 3)), borrowing lanes one row lower
        const uint32_t sl = s & 63, sh = s >> 6;
        const bool borrow = (uint32_t)lane < sl;
        const uint64_t keep_unless_wrapped = neg ? 0ull : ~0ull;
        const uint64_t *vbase = stage + (((uint32_t)lane - sl) & 63) + (borrow ? 0 : 64);
        // as in fft_kernels.hip: the index wrapped where 64 row + lane < s (one vector compare per row against a per-lane
        // threshold), and the reads run RW rows ahead of their use instead of paying one LDS round trip each
        const int32_t thr = (int32_t)s - lane;
        // rows in pairs (one address, ds_read2st64_b64): this wave's register pair (r, r + 1) is two consecutive rows
        constexpr int RW = 8;
        uint64_t vq[RW];
#pragma unroll
        for (int k = 0; k < RW; k += 2) {
            const uint64_t *pb = vbase + 64 * ((8 * h + (k & 7) + 16 * (k >> 3) - sh) & 31);
            vq[k] = pb[0];
            vq[k + 1] = pb[64];
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = 8 * h + (r & 7) + 16 * (r >> 3);
            const uint64_t v = vq[r % RW];
            if ((r & 1) && r + RW - 1 < 16) {
                const int q = r + RW - 1;
                const uint64_t *pb = vbase + 64 * ((8 * h + (q & 7) + 16 * (q >> 3) - sh) & 31);
                vq[(r - 1) % RW] = pb[0];
                vq[r % RW] = pb[64];
            }
            const uint64_t wrapmask = __builtin_amdgcn_ballot_w64(64 * row < thr);
            const uint32_t nhi = rot_sub_hi_compl(v, acc[r], wrapmask ^ keep_unless_wrapped);
            const int32_t dig = (int32_t)(0xFFu - nhi) >> 9;
            if (r < 8) z[r].r = (double)dig; else z[r - 8].i = (double)dig;
            if (XI32 && r < 8) dre[r] = dig;
            // WIDE: the exchange area is this wavefront's own, so point r - 8 goes out as soon as its imaginary digit exists
            // (the partner's last read of it was before barrier 4 of the previous iteration)
            if (XI32 && r >= 8) { int2_t d2; d2.x = dre[r - 8]; d2.y = dig; xmine_i[lane + 64 * (r - 8)] = d2; }
            else if (EARLY_X && r >= 8) xmine[pslot(lane + 64 * (r - 8))] = z[r - 8];
            __builtin_amdgcn_sched_barrier(0);
        }
        FFT4_TL(2);
        if (!WIDE) __syncthreads();                   // all rotated reads done before the area is reused

        // key rows of this iteration: own transform first (row j), then the partner polynomial's (row 1-j), column j;
        // point n = 8 (64 h + lane) + c sits at [c16 = 8 (lane & 1) + c][L = 32 h + (lane >> 1)] of the key layout
        typedef double __attribute__((ext_vector_type(2))) double2_t;
        const size_t koff = (size_t)(8 * (lane & 1)) * 64 + 32 * h + (lane >> 1);
        const double2_t *b_own = reinterpret_cast<const double2_t *>(P.bsk_fft) + ((((size_t)i * 2 + j) * 2 + j)) * FM + koff;
        const double2_t *b_par = reinterpret_cast<const double2_t *>(P.bsk_fft) + ((((size_t)i * 2 + (1 - j)) * 2 + j)) * FM + koff;
        constexpr int HB = WIDE ? FFT4_HB_WIDE : FFT4_HB_NARROW;
        double2_t bo[HB], bp[HB];
        constexpr bool SPREAD = FFT4_KEY_SPREAD && (WIDE ? HB == 8 : HB == 4);
        constexpr int LDK_STRIDE = 8 / HB;           // HB = 8: chunk k at point k; HB = 4: chunk k at point 2 k, 4..7 in the product
        // SPREAD: the first HB chunks are requested at points of the forward transform, pinned there
#define FFT4_LDK(pt) do { if (SPREAD && (pt) % LDK_STRIDE == 0) { constexpr int k = (pt) / LDK_STRIDE; __builtin_amdgcn_sched_barrier(0); bo[k] = b_own[k * 64]; bp[k] = b_par[k * 64]; \
                                        __builtin_amdgcn_sched_barrier(0); } } while (0)
        if (!SPREAD) {
#pragma unroll
            for (int k = 0; k < HB; k++) { bo[k] = b_own[k * 64]; bp[k] = b_par[k * 64]; }
        }
        FFT4_LDK(0);

        // ---- forward transform -------------------------------------------------------------------------------
        __builtin_amdgcn_s_setprio(1);
        {   // stage t = 512 across the two halves: (a, b) = (lower, upper) point, this half keeps its own output
            if (!EARLY_X) {
#pragma unroll
                for (int r = 0; r < 8; r++) xmine[pslot(lane + 64 * r)] = z[r];
            }
            __syncthreads();
            FFT4_TL(3);
            // all 8 partner points requested first, ONE wave-uniform branch around the butterflies (a branch and a
            // serialized LDS round trip per point before)
            cplx o[8];
            if (XI32) {
                int2_t oi[8];
#pragma unroll
                for (int r = 0; r < 8; r++) oi[r] = pair_i[lane + 64 * r];
#pragma unroll
                for (int r = 0; r < 8; r++) { o[r].r = (double)oi[r].x; o[r].i = (double)oi[r].y; }
            } else {
#pragma unroll
                for (int r = 0; r < 8; r++) o[r] = pair[pslot(lane + 64 * r)];
            }
            if (h == 0) {
#pragma unroll
                for (int r = 0; r < 8; r++) bf_fwd<false>(z[r], o[r], w1.r, w1.i);
            } else {
#pragma unroll
                for (int r = 0; r < 8; r++) bf_fwd<false>(o[r], z[r], w1.r, w1.i);
            }
            if (!WIDE) __syncthreads();               // the partner has read this half's points
        }
        FFT4_TL(4);
        FFT4_LDK(1);
        stagesA_head<false>(z, w2, w4, w8);
        FFT4_LDK(2);
        if (REG_T1) {                                 // A' -> B' inside the register file
#pragma unroll
            for (int k = 0; k < 4; k++) bf_fwd<false>(z[2 * k], z[2 * k + 1], w8[k].r, w8[k].i);
            reg_lane_swap<32, 4>(z);
            reg_lane_swap<16, 2>(z);
            reg_lane_swap<8, 1>(z);
        } else {
            pipeline4<OVERLAP>([&](int k) { bf_fwd<false>(z[2 * k], z[2 * k + 1], w8[k].r, w8[k].i); },
                      [&](int k) { mine[pslot(lane + 64 * (2 * k))] = z[2 * k]; mine[pslot(lane + 64 * (2 * k + 1))] = z[2 * k + 1]; });
            __builtin_amdgcn_wave_barrier();
            {
                const cplx *rd = mine + pslot(64 * (lane >> 3)) + (lane & 7);     // pslot(64 a + 8 r + b) = 72 a + 9 r + b
#pragma unroll
                for (int r = 0; r < 8; r++) z[r] = rd[9 * r];
            }
            __builtin_amdgcn_wave_barrier();
        }
        FFT4_LDK(3);
        {
            stage8<false, 4>(z, t32, t32);
            FFT4_LDK(4);
            stage8<false, 2>(z, t16, t16);
            FFT4_LDK(5);
            if (REG_T2) {                             // B' -> C' inside the register file
                stage8<false, 1>(z, t8a, t8b);
                reg_lane_swap<4, 4>(z);
                reg_lane_swap<2, 2>(z);
                reg_lane_swap<1, 1>(z);
            } else {
                cplx *wr = mine + pslot(64 * (lane >> 3)) + (lane & 7);
                pipeline4<OVERLAP>([&](int k) { last_bf<false, 1>(z, k, t8a, t8b); },
                          [&](int k) { wr[9 * (2 * k)] = z[2 * k]; wr[9 * (2 * k + 1)] = z[2 * k + 1]; });
            }
        }
        if (!REG_T2) {
            __builtin_amdgcn_wave_barrier();
            {
                const cplx *rd = mine + 9 * lane;                             // pslot(8 lane + c) = 9 lane + c
#pragma unroll
                for (int c = 0; c < 8; c++) z[c] = rd[c];
            }
            __builtin_amdgcn_wave_barrier();
        }
        FFT4_LDK(6);
        FFT4_TL(5);                                   // (before the last three in-wave stages: layout C')
        // ---- last forward stage + publish, pointwise multiply-accumulate with GGSW_i ---------------------------------
        {
            stage8<false, 4>(z, t4, t4);
            FFT4_LDK(7);
            stage8<false, 2>(z, t2, t2);
            pipeline4<OVERLAP>([&](int k) { last_bf<false, 1>(z, k, t1a, t1b); },
                      [&](int k) { pmine[(2 * k) * 64 + lane] = z[2 * k]; pmine[(2 * k + 1) * 64 + lane] = z[2 * k + 1]; });
        }
        __syncthreads();
        FFT4_TL(6);
        __builtin_amdgcn_s_setprio(2);
        cplx og[8];                                   // the partner's 8 points, all requested before the first product
#pragma unroll
        for (int c = 0; c < 8; c++) og[c] = other[c * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int k = c % HB;
            const cplx g = og[c];
            const double fr = z[c].r, fi = z[c].i;
            double rr = fr * bo[k].x; rr = __builtin_fma(-fi, bo[k].y, rr);
            rr = __builtin_fma(g.r, bp[k].x, rr); rr = __builtin_fma(-g.i, bp[k].y, rr);
            double ii = fr * bo[k].y; ii = __builtin_fma(fi, bo[k].x, ii);
            ii = __builtin_fma(g.r, bp[k].y, ii); ii = __builtin_fma(g.i, bp[k].x, ii);
            z[c].r = rr; z[c].i = ii;
            if (c + HB < 8) { bo[k] = b_own[(c + HB) * 64]; bp[k] = b_par[(c + HB) * 64]; }
        }
#undef FFT4_LDK
        if (!WIDE) __syncthreads();                   // the other polynomial has read this wave's transform
        __builtin_amdgcn_s_setprio(0);

        FFT4_TL(7);
        // ---- inverse transform -------------------------------------------------------------------------------
        {
            stage8<true, 1>(z, t1a, t1b);
            stage8<true, 2>(z, t2, t2);
            if (REG_T2) {                             // C' -> B' inside the register file
                stage8<true, 4>(z, t4, t4);
                reg_lane_swap<1, 1>(z);
                reg_lane_swap<2, 2>(z);
                reg_lane_swap<4, 4>(z);
            } else {
                cplx *wr = mine + 9 * lane;
                pipeline4<OVERLAP>([&](int k) { last_bf<true, 4>(z, k, t4, t4); },
                          [&](int k) { wr[k] = z[k]; wr[k + 4] = z[k + 4]; });
            }
        }
        if (!REG_T2) {
            __builtin_amdgcn_wave_barrier();
            {
                const cplx *rd = mine + pslot(64 * (lane >> 3)) + (lane & 7);
#pragma unroll
                for (int r = 0; r < 8; r++) z[r] = rd[9 * r];
            }
            __builtin_amdgcn_wave_barrier();
        }
        {
            stage8<true, 1>(z, t8a, t8b);
            stage8<true, 2>(z, t16, t16);
            if (REG_T1) {                             // B' -> A' inside the register file
                stage8<true, 4>(z, t32, t32);
                reg_lane_swap<8, 1>(z);
                reg_lane_swap<16, 2>(z);
                reg_lane_swap<32, 4>(z);
            } else {
                cplx *wr = mine + pslot(64 * (lane >> 3)) + (lane & 7);
                pipeline4<OVERLAP>([&](int k) { last_bf<true, 4>(z, k, t32, t32); },
                          [&](int k) { wr[9 * k] = z[k]; wr[9 * (k + 4)] = z[k + 4]; });
            }
        }
        if (!REG_T1) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 8; r++) z[r] = mine[pslot(lane + 64 * r)];
            __builtin_amdgcn_wave_barrier();
        }
        FFT4_TL(8);                                   // (before the last three in-wave stages: layout A')
        stagesA_head<true>(z, w2, w4, w8);
        {   // last in-wave stage (t = 256) + stage t = 512 across the two halves
            pipeline4<OVERLAP>([&](int k) { bf_inv<false>(z[k], z[k + 4], w2.r, w2.i); },
                      [&](int k) { xmine[pslot(lane + 64 * k)] = z[k]; xmine[pslot(lane + 64 * (k + 4))] = z[k + 4]; });
            __syncthreads();
            FFT4_TL(9);
            cplx o[8];
#pragma unroll
            for (int r = 0; r < 8; r++) o[r] = pair[pslot(lane + 64 * r)];
            if (h == 0) {
#pragma unroll
                for (int r = 0; r < 8; r++) bf_inv<false>(z[r], o[r], w1.r, w1.i);
            } else {
#pragma unroll
                for (int r = 0; r < 8; r++) bf_inv<false>(o[r], z[r], w1.r, w1.i);
            }
            if (!WIDE) __syncthreads();               // the partner has read this half's points
        }

        FFT4_TL(10);
        // ---- back to the torus, update and restage the accumulator -------------------------------------------
#pragma unroll
        for (int r = 0; r < 8; r++) {
            acc[r] += to_torus(z[r].r);
            stage[64 + k0 + 64 * r] = acc[r];
            acc[r + 8] += to_torus(z[r].i);
            stage[64 + k0 + 64 * r + 1024] = acc[r + 8];
            if (r == 0 && h == 0) stage[64 + 2048 + lane] = acc[0];
            if (r == 7 && h == 1) stage[lane] = acc[15];
        }
        FFT4_TL(11);
    }

    uint64_t *out = P.out_ptrs ? P.out_ptrs[ct] : P.out + (size_t)ct * BIG_CT;
    if (j == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int n = k0 + 64 * (r & 7) + 1024 * (r >> 3);
            if (n == 0) out[0] = acc[r] + 1;
            else out[POLY_N - n] = (uint64_t)0 - (acc[r] + 1);
        }
    } else {
        if (k0 == 0) out[BIG_N] = acc[0] + 1;
        uint64_t *body = P.body_ptrs ? P.body_ptrs[ct] : nullptr;      // rotation sharing: the whole body polynomial
        if (body) {
#pragma unroll
            for (int r = 0; r < 16; r++) body[k0 + 64 * (r & 7) + 1024 * (r >> 3)] = acc[r] + 1;
        }
    }
}

}  // namespace

// used for batches <= 512 (at most 2 workgroups per CU): aim for 2 waves per SIMD and spend registers on latency
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void blind_rotate_fft4_kernel(BlindRotateFftParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fft4_bootstrap<false>(P, smem, blockIdx.x);
}
// batches of at most one ciphertext per CU: four LDS areas (136 KB per workgroup), 4 instead of 8 barriers per iteration
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void blind_rotate_fft4_wide_kernel(BlindRotateFftParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fft4_bootstrap<true>(P, smem, blockIdx.x);
}

// more than 64 KB of dynamic LDS is a per-DEVICE opt-in (like the exact kernel's, pbs_kernels.hip): Context::init calls
// this with the device current
hipError_t prepare_device_for_fft4() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(blind_rotate_fft4_wide_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 16 * F4_WAVE_BYTES);
}

hipError_t launch_blind_rotate_fft4(const BlindRotateFftParams &p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    if (p.B * 4 <= p.slots)        // slots = 4 per CU: at most one of these workgroups per CU
        hipLaunchKernelGGL(blind_rotate_fft4_wide_kernel, dim3(p.B), dim3(256), 16 * F4_WAVE_BYTES, s, p);
    else
        hipLaunchKernelGGL(blind_rotate_fft4_kernel, dim3(p.B), dim3(256), 4 * F4_WAVE_BYTES, s, p);
    return hipGetLastError();
}

}  // namespace fhs
