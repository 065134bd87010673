// Optional arithmetic mode: blind rotation with an f64 complex FFT (gfx950 only).
//
// This is the algorithm CLASS the reference's CPU engine uses (tfhe 0.5.2 + concrete-fft 0.4.0,
// Cargo.lock:168-179): the negacyclic product through a folded 1024-point complex transform
// (z[n] = x[n] + i x[n+1024]).  It needs ~4x fewer FP64 operations than the exact two-prime NTT of
// pbs_kernels.hip but is approximate w.r.t. exact integer arithmetic (53-bit mantissa; the error is
// far below the scheme's noise, like in the reference).  The exact NTT stays the default and the
// parity anchor; this mode is selected with fhs_set_arithmetic(ctx, FHS_ARITH_F64_FFT).
// It is deterministic, and mode 3 of the CPU oracle mirrors it lane for lane with the same
// IEEE-754 operation order, so the GPU output is still checked bit for bit.
//
// Mapping: a workgroup of 2 wavefronts works on one ciphertext at a time (persistent: it takes the next one from
// a counter until the batch is done), wavefront j owns GLWE polynomial j: 16 complex points per lane in registers.  The 10 radix-2 stages run in three register layouts, each
// with 4 (or 2) of the index bits in the register number so that every butterfly is in-lane:
//   A  lane = n mod 64,                    reg = n bits 6-9   stages t = 512..64 (lane-uniform twiddles)
//   B  lane = 4*(n div 64) + (n mod 4),    reg = n bits 2-5   stages t = 32..4
//   C  lane = n div 16,                    reg = n bits 0-3   stages t = 2, 1
// with two transposes through LDS per transform (16-byte accesses, slot n + n/16: conflict free).
// The twist of the negacyclic fold is merged into the twiddle table
//   W[m+i] = exp(i*pi/2048 * (1024/2m) * (4*bitrev(i) + 1))     (host: fft_tables.cpp)
// and W[m+i+1] = i * W[m+i] (i even) is applied as a free rotation instead of a second twiddle.
#include "fft_transform.h"

namespace fhs {

#pragma clang fp contract(off)

namespace {
using namespace fftdev;

}  // namespace

__global__ __launch_bounds__(128, 2) void blind_rotate_fft_kernel(BlindRotateFftParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int j = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // GLWE polynomial of this wave
    double *my = reinterpret_cast<double *>(smem) + j * FFT_LDS_DOUBLES;
    const double *partner = reinterpret_cast<double *>(smem) + (1 - j) * FFT_LDS_DOUBLES;
    uint64_t *my_u = reinterpret_cast<uint64_t *>(my);
    int *next_ct = reinterpret_cast<int *>(smem + 2 * FFT_LDS_DOUBLES * sizeof(double));

    // Persistent workgroups: the grid holds at most one workgroup per resident slot (4 per CU) and every workgroup
    // takes ciphertexts from a counter until none is left.  Left to the hardware dispatcher, a launch of an exact
    // multiple of the slot count ended with a few workgroups starting a whole bootstrap late (one XCD had been
    // handed a few more than its share): 20.9 ms for 2048 ciphertexts instead of 16.5 ms.
    for (;;) {
    if (threadIdx.x == 0) *next_ct = (int)atomicAdd(P.work_counter, 1u);
    __syncthreads();
    const int ct = __builtin_amdgcn_readfirstlane(*next_ct);
    __syncthreads();                                  // both waves hold ct before slot 0 can be rewritten
    if (ct >= P.B) break;

    const uint64_t *ks = P.ks + (size_t)ct * SMALL_CT;

    // acc[r] = coefficient (lane + 64 r) of polynomial j (u64 torus) MINUS ONE -- the offset makes the rotate-and-
    // subtract exact in five operations (rot_sub_hi_compl); registers r and r+16 form one complex point
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
            acc[r] = v - 1;
        }
    }

#pragma unroll
    for (int r = 0; r < 32; r++) my_u[64 + lane + 64 * r] = acc[r];
    my_u[lane] = acc[31];                             // row 31 again in front of row 0 (see the rotated read)
    my_u[64 + lane + 64 * 32] = acc[0];               // and row 0 again behind row 31: rows are read in pairs
    // the mask element of the NEXT iteration is requested one iteration ahead (ks[LWE_N], the body, is a valid address):
    // read at the top of its own iteration it cost one exposed global-memory round trip per iteration
    uint64_t ks_next = ks[0];
    for (int i = 0; i < LWE_N; i++) {
        asm volatile("" : "+v"(ks_next));             // opaque until here: nothing of the next iteration is computed early
        const uint32_t a = __builtin_amdgcn_readfirstlane(fft_mod_switch(ks_next));
        __builtin_amdgcn_sched_barrier(0);            // use the value requested an iteration ago BEFORE the next request
        ks_next = ks[i + 1];
        if (a == 0) continue;
        const uint32_t s = a & (POLY_N - 1);
        const bool neg = a >= POLY_N;
        // Wave priority follows the distance to the next workgroup barrier: the partner wave (on another SIMD,
        // beside a wave of another workgroup) is parked there until this one arrives.  Forward transform 1,
        // pointwise phase between the two barriers 2, inverse transform and update (no barrier ahead) 0:
        // measured -4..-10 % kernel time against equal priorities.
        __builtin_amdgcn_s_setprio(1);

        // keep the (loop-invariant) per-lane twiddle loads inside the iteration: 24 registers that are only
        // live while a transform runs instead of across the whole loop
        const double *lanetab = P.lanetab;
        asm volatile("" : "+s"(lanetab));

        // rotate, subtract, decompose; fold: z[r] = digit[r] + i * digit[r + 16]
        cplx z[16];
        __builtin_amdgcn_wave_barrier();
        // coefficient n = lane + 64 r comes from m = (n - s) mod 2048: a lane rotation by s mod 64 (one per-lane base
        // address) and a row rotation by s div 64 (a scalar offset per register); lanes that borrow read one row
        // lower, which for row 0 is the copy of row 31 kept in front of it.  The sign flips where the index wrapped.
        const uint32_t sl = s & 63, sh = s >> 6;
        const bool borrow = (uint32_t)lane < sl;
        const uint64_t keep_unless_wrapped = neg ? 0ull : ~0ull;
        const uint64_t *vbase = my_u + (((uint32_t)lane - sl) & 63) + (borrow ? 0 : 64);
        // the index wrapped (sign flip) where n = lane + 64 r < s, i.e. 64 r < s - lane: one vector compare per row
        // against a per-lane threshold instead of ~7 scalar operations per row building the same lane mask
        const int32_t thr = (int32_t)s - lane;
        // the reads run RW rows ahead of their use: issued one at a time right before its use, every row paid a full
        // LDS round trip (32 exposed round trips per iteration).  Rows are read in PAIRS (source rows rho, rho + 1 of
        // one address: ds_read2st64_b64, one address computation for two rows); row 32 is the copy of row 0.
        constexpr int RW = 8;
        uint64_t vq[RW];
#pragma unroll
        for (int k = 0; k < RW; k += 2) {
            const uint64_t *pb = vbase + 64 * ((k - sh) & 31);
            vq[k] = pb[0];
            vq[k + 1] = pb[64];
        }
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const uint64_t v = vq[r % RW];
            if ((r & 1) && r + RW - 1 < 32) {
                const uint64_t *pb = vbase + 64 * ((r + RW - 1 - sh) & 31);
                vq[(r - 1) % RW] = pb[0];
                vq[r % RW] = pb[64];
            }
            const uint64_t wrapmask = __builtin_amdgcn_ballot_w64(64 * r < thr);
            const uint32_t nhi = rot_sub_hi_compl(v, acc[r], wrapmask ^ keep_unless_wrapped);
            const int32_t dig = (int32_t)(0xFFu - nhi) >> 9;
            if (r < 16) z[r].r = (double)dig; else z[r - 16].i = (double)dig;
            __builtin_amdgcn_sched_barrier(0);        // keep the reads of rows r + RW - 1, r + RW behind the use of row r
        }
        __builtin_amdgcn_wave_barrier();

        // Key rows for this iteration, column j: own transform's row j, then the partner's row 1-j.  An 8-point window
        // (32 VGPRs): the own row's first 8 points are requested before the transform; see publish_mul_hook for the rest.
        const double2_t *b_own = reinterpret_cast<const double2_t *>(P.bsk_fft) + ((((size_t)i * 2 + j) * 2 + j)) * FM + lane;
        const double2_t *b_par = reinterpret_cast<const double2_t *>(P.bsk_fft) + ((((size_t)i * 2 + (1 - j)) * 2 + j)) * FM + lane;
        double2_t kb[8];
#pragma unroll
        for (int k = 0; k < 8; k++) kb[k] = b_own[k * 64];

        {
            LaneTw tw;
            load_lane_tw(tw, lanetab, lane);
            fft_forward(z, my, lane, tw);
            fft_forward_last(z, tw, publish_mul_hook{reinterpret_cast<cplx *>(my) + lane, z, kb, b_own, b_par});
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(2);
        {   // partner row: z[c] += g[c] * kb (the second half of the pointwise product, same operation order as before)
            const cplx *par = reinterpret_cast<const cplx *>(partner) + lane;
            // PW reads in flight ahead of their use (one at a time, each point waited for its own LDS round trip)
            constexpr int PW = 4;
            cplx gq[PW];
#pragma unroll
            for (int c = 0; c < PW; c++) gq[c] = par[c * 64];
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const cplx g = gq[c % PW];
                if (c + PW < 16) gq[c % PW] = par[(c + PW) * 64];
                const double2_t k = kb[c & 7];
                double rr = __builtin_fma(g.r, k.x, z[c].r); rr = __builtin_fma(-g.i, k.y, rr);
                double ii = __builtin_fma(g.r, k.y, z[c].i); ii = __builtin_fma(g.i, k.x, ii);
                z[c].r = rr; z[c].i = ii;
                if (c < 8) kb[c] = b_par[(c + 8) * 64];
            }
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(0);

        {   // reloaded rather than kept live across the pointwise phase
            const double *lanetab2 = P.lanetab;
            asm volatile("" : "+s"(lanetab2));
            LaneTw tw2;
            load_lane_tw(tw2, lanetab2, lane);
            fft_inverse(z, my, lane, tw2);
        }

        // update, and stage the new accumulator in LDS for the next iteration's rotated read (the store
        // burst overlaps the conversions instead of stalling the start of the next iteration)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            acc[r] += to_torus(z[r].r);
            my_u[64 + lane + 64 * r] = acc[r];
            acc[r + 16] += to_torus(z[r].i);
            my_u[64 + lane + 64 * (r + 16)] = acc[r + 16];
            if (r == 0) my_u[64 + lane + 64 * 32] = acc[0];
            if (r == 15) my_u[lane] = acc[31];
        }
    }

    uint64_t *out = P.out_ptrs ? P.out_ptrs[ct] : P.out + (size_t)ct * BIG_CT;
    if (j == 0) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const int n = lane + 64 * r;
            if (n == 0) out[0] = acc[r] + 1;
            else out[POLY_N - n] = (uint64_t)0 - (acc[r] + 1);
        }
    } else {
        if (lane == 0) out[BIG_N] = acc[0] + 1;
        uint64_t *body = P.body_ptrs ? P.body_ptrs[ct] : nullptr;      // rotation sharing: the whole body polynomial
        if (body) {
#pragma unroll
            for (int r = 0; r < 32; r++) body[lane + 64 * r] = acc[r] + 1;
        }
    }
    }   // persistent loop
}

// Bootstrapping key -> Fourier domain with the device's own forward transform: one wavefront per polynomial.
// in: [742*4][2048] u64 standard domain;  out: [742*4][16][64 lanes][2], pre-scaled by 2^-74 (1/1024 for the transform pair, 2^-64 for the torus conversion)
__global__ __launch_bounds__(64) void bsk_to_fft_kernel(const uint64_t *__restrict__ bsk_std, double *__restrict__ out,
                                                        const double *__restrict__ lanetab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const uint64_t *src = bsk_std + (size_t)blockIdx.x * POLY_N;
    cplx z[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        // round to the 58-bit torus grid (same rounding as the exact path), signed representative
        const uint64_t q0 = (src[lane + 64 * r] + (1ull << (BSK_QUANT_BITS - 1))) & ~((1ull << BSK_QUANT_BITS) - 1);
        const uint64_t q1 = (src[lane + 64 * r + 1024] + (1ull << (BSK_QUANT_BITS - 1))) & ~((1ull << BSK_QUANT_BITS) - 1);
        // 58 significant bits -> split so that both halves convert exactly, one rounding in the add
        z[r].r = (double)(int32_t)(q0 >> 32) * 0x1p32 + (double)(uint32_t)q0;
        z[r].i = (double)(int32_t)(q1 >> 32) * 0x1p32 + (double)(uint32_t)q1;
    }
    LaneTw tw;
    load_lane_tw(tw, lanetab, lane);
    fft_forward(z, reinterpret_cast<double *>(smem), lane, tw);
    stage_lane<false, 1>(z, tw.re[5], tw.im[5]);
    double *dst = out + (size_t)blockIdx.x * 2 * FM;
#pragma unroll
    for (int c = 0; c < 16; c++) {
        dst[(c * 64 + lane) * 2 + 0] = z[c].r * 0x1p-74;
        dst[(c * 64 + lane) * 2 + 1] = z[c].i * 0x1p-74;
    }
}

void fft_uniform_consts(double *w_re, double *w_im, double *u_re, double *u_im) {
    for (int k = 0; k < 16; k++) { w_re[k] = FW_RE[k]; w_im[k] = FW_IM[k]; }
    for (int k = 0; k < 3; k++) { u_re[k] = FU_RE[k]; u_im[k] = FU_IM[k]; }
}

hipError_t launch_blind_rotate_fft(const BlindRotateFftParams &p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    const size_t lds = (size_t)2 * FFT_LDS_DOUBLES * sizeof(double) + 16;
    hipError_t e = hipMemsetAsync(p.work_counter, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const int grid = p.B < p.slots ? p.B : p.slots;
    hipLaunchKernelGGL(blind_rotate_fft_kernel, dim3(grid), dim3(128), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_bsk_to_fft(const uint64_t *d_bsk_std, double *d_out, const double *d_lanetab, hipStream_t s, int n_polys) {
    const size_t lds = (size_t)FFT_LDS_DOUBLES * sizeof(double);
    hipLaunchKernelGGL(bsk_to_fft_kernel, dim3(n_polys), dim3(64), lds, s, d_bsk_std, d_out, d_lanetab);
    return hipGetLastError();
}

}  // namespace fhs
