// Device kernels of the batched programmable bootstrap (gfx950 only).
// Replaces tfhe::shortint's keyswitch -> modulus switch -> blind rotation -> sample extract
// (SURVEY.md 3.3 / Appendix A), reached by the reference from src/ciphertext/fheasciichar.rs:36-102.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace fhs {

constexpr int LWE_N = 742;
constexpr int POLY_N = 2048;
constexpr int BIG_N = 2048;
constexpr int BIG_CT = 2049;
constexpr int SMALL_CT = 743;
constexpr int KS_LEVEL = 5;
constexpr int KS_BASE_LOG = 3;
constexpr int DELTA_LOG = 59;
constexpr int BSK_QUANT_BITS = 6;

// NTT primes: the two largest p < 2^47 with p = 1 (mod 4096).  p0*p1 > 2^93 bounds the exact
// integer result of one external product on the 58-bit key grid (|x| <= 2^91).
constexpr uint64_t NTT_P0 = 140737488273409ull;  // 0x7ffffffec001
constexpr uint64_t NTT_P1 = 140737488252929ull;  // 0x7ffffffe7001
constexpr uint64_t NTT_PSI0 = 124135596386681ull;  // primitive 4096-th roots
constexpr uint64_t NTT_PSI1 = 116052249937262ull;

// Twiddle tables (doubles holding exact centred residues), see ntt_tables.cpp.
struct NttTables {
    const double *fwd_uni;   // [2 primes][32]      Psi[1..31]           (lane-uniform stages)
    const double *fwd_lane;  // [2][32 entries][64] per-lane stages t=32..1
    const double *inv_uni;   // [2][64]             PsiInv[1..63]        (t=32 stage + uniform stages)
    const double *inv_lane;  // [2][32][64]         per-lane stages t=1..16 (entry 0 unused)
};

struct BlindRotateParams {
    const uint64_t *ks;       // [B][743] keyswitched small LWE (u64 torus), mod-switched in-kernel
    const uint32_t *lut_idx;  // [B]
    const uint64_t *luts;     // [L][2048]
    const double *bsk_ntt;    // [742][row 2][col 2][prime 2][16][64 lanes][2], pre-scaled by N^-1
    NttTables tw;
    double crt_c;             // p0^-1 mod p1, centred
    uint64_t *out;            // [B][2049] dense output, or
    uint64_t *const *out_ptrs; // [B] per-ciphertext output blocks (used when non-null)
    uint64_t *const *body_ptrs; // [B] or null; body_ptrs[ct] != null: also store the accumulator's BODY polynomial (2048
                               // words) there -- rotation sharing: further sample extractions of the same rotation
                               // (extract_shift_kernel) need B[h], the mask polynomial is in the output LWE already
    int B;
};

// Optional f64-FFT arithmetic mode (fft_kernels.hip); same inputs/outputs as BlindRotateParams.
struct BlindRotateFftParams {
    const uint64_t *ks;
    const uint32_t *lut_idx;
    const uint64_t *luts;
    const double *bsk_fft;    // [742][row 2][col 2][16][64 lanes][2 re,im], pre-scaled by 2^-74 (1/1024 and 2^-64)
    const double *lanetab;    // [12][64] per-lane twiddle bases (fft_tables.cpp), 2-wavefront kernel
    const double *weff;       // [1024][2] effective twiddles (fft_tables.cpp), 4-wavefront kernel
    uint32_t *work_counter;   // 2-wavefront kernel: next ciphertext to take (zeroed by the launcher)
    int slots;                // 2-wavefront kernel: resident workgroup slots of the device (4 per CU)
    uint64_t *out;
    uint64_t *const *out_ptrs;
    uint64_t *const *body_ptrs;   // see BlindRotateParams
    int B;
};

// Two key bits per external product (fftmb_kernels.hip, FHS_ARITH_F64_FFT_MB2); same inputs/outputs again.
struct BlindRotateMb2Params {
    const uint64_t *ks;
    const uint32_t *lut_idx;
    const uint64_t *luts;
    const double *bsk_mb;     // [371 pairs][K1,K2,K3][row 2][col 2][16][64 lanes][2 re,im], pre-scaled by 2^-74
    const double *lanetab;    // [12][64] per-lane twiddle bases (fft_tables.cpp)
    const double *mono;       // [4096][2] exp(i*pi*k/2048): monomial evaluation table
    const double *r16;        // [16][2]   exp(i*pi*k/8)
    uint32_t *work_counter;
    int slots;
    uint64_t *out;
    uint64_t *const *out_ptrs;
    uint64_t *const *body_ptrs;   // see BlindRotateParams
    int B;
};

// Two key bits per external product in exact arithmetic (nttmb_kernels.hip, FHS_ARITH_EXACT_NTT_MB2).
struct BlindRotateNttMb2Params {
    const uint64_t *ks;
    const uint32_t *lut_idx;
    const uint64_t *luts;
    const double *bsk_ntt_mb; // [371 pairs][K1,K2,K3][row 2][col 2][prime 2][16][64 lanes][2], pre-scaled by N^-1 (57-bit grid)
    NttTables tw;
    double crt_c;
    const double *mono;       // [2 primes][4096] psi_q^k centred
    uint64_t *out;
    uint64_t *const *out_ptrs;
    uint64_t *const *body_ptrs;   // see BlindRotateParams
    int B;
};

// One lincomb output: out[dst] = sum_t coef[t] * src[t] + konst * 2^59 (body only)
struct LinDesc {
    uint32_t first_term;
    uint32_t n_terms;
    uint64_t konst_body;      // already shifted: (k mod 32) << 59
};
struct LinTerm {
    const uint64_t *src;      // device block
    int64_t coef;
};

size_t blind_rotate_lds_bytes();
// per-device function attributes (dynamic LDS above 64 KB); call with the device current
hipError_t prepare_device_for_kernels();
hipError_t read_device_ntt_consts(double *fwd_uni /*[64]*/, double *inv_uni /*[128]*/, double *crt);
hipError_t launch_blind_rotate(const BlindRotateParams &p, hipStream_t s);
hipError_t launch_blind_rotate_fft(const BlindRotateFftParams &p, hipStream_t s);    // 2 wavefronts per ciphertext
hipError_t launch_blind_rotate_fft4(const BlindRotateFftParams &p, hipStream_t s);   // 4 wavefronts per ciphertext
// standard-domain key [n_polys][2048] u64 -> Fourier-domain key, with the device's own forward transform
hipError_t launch_bsk_to_fft(const uint64_t *d_bsk_std, double *d_out, const double *d_lanetab, hipStream_t s,
                             int n_polys = LWE_N * 4);
hipError_t launch_blind_rotate_mb2(const BlindRotateMb2Params &p, hipStream_t s);    // 2 wavefronts per ciphertext
hipError_t launch_blind_rotate_ntt_mb2(const BlindRotateNttMb2Params &p, hipStream_t s);   // 4 wavefronts per ciphertext
hipError_t prepare_device_for_ntt_mb2();
hipError_t prepare_device_for_fft4();
hipError_t prepare_device_for_keyswitch();      // > 64 KB dynamic LDS opt-in of the wide keyswitch kernel, per device
// the scalar twiddle literals baked into fft_kernels.hip: W[16] (index 1 and even indices used), U[3]
void fft_uniform_consts(double *w_re, double *w_im, double *u_re, double *u_im);
// matrix-core keyswitch (ks_kernels.hip): KSK as 8 planes of balanced signed bytes in MFMA fragment order
size_t ks_planes_bytes();
size_t ks_digits_bytes(int B);
hipError_t launch_ksk_to_planes(const uint64_t *d_ksk, int8_t *d_planes, hipStream_t s);
hipError_t launch_keyswitch_mfma(const uint64_t *d_in /*[B][2049]*/, const int8_t *d_planes, int8_t *d_dig /*scratch*/,
                                 uint64_t *d_ks_out /*[B][743]*/, int B, hipStream_t s, int n_cus = 256);
hipError_t launch_modswitch(const uint64_t *d_ks, uint32_t *d_ms /*[B][743]*/, int B, hipStream_t s);
hipError_t launch_lincomb(const LinDesc *d_desc, const LinTerm *d_terms, uint64_t *d_out /*[n][2049]*/,
                          int n, hipStream_t s);
hipError_t launch_scatter_blocks(const uint64_t *d_in, uint64_t *const *d_dst, int n, hipStream_t s);
hipError_t launch_gather_rows(const uint64_t *const *d_src, uint64_t *d_out, int n, hipStream_t s);   // util_kernels.hip
// Rotation sharing: one more sample extraction of a finished blind rotation.  `lead` = the output LWE of that rotation
// (its mask IS the accumulator's mask polynomial in extract-at-0 order), `body` = the accumulator's body polynomial
// (body_ptrs of the blind-rotation kernels), K in [0, 4096) = negacyclic coefficient index to extract: out = what a
// bootstrap of (input + (K / 128) * Delta) with the same look-up table yields.
struct ExtractDesc {
    const uint64_t *lead;
    const uint64_t *body;
    uint64_t *out;
    uint32_t K, pad;
};
hipError_t launch_extract_shift(const ExtractDesc *d_desc, int n, hipStream_t s);

}  // namespace fhs
