// Host-side tables and key conversion for the optional f64-FFT arithmetic mode (fft_kernels.hip).
#pragma once
#include <cstdint>
#include <vector>

namespace fhs {

struct HostFftTables {
    std::vector<double> w_re, w_im;   // [1024] merged-twist twiddles W[m+i], index 0 unused
    std::vector<double> u_re, u_im;   // [16]   U[G+g] = exp(i*pi*bitrev(g)/G), G = 2,4,8
    std::vector<double> lanetab;      // [12][64] per-lane constants: wa, wb (fused stage), bases G=1,2,4,8 (re, im)
};
void build_fft_tables(HostFftTables &t);

// bsk_std: [742][2][2][2048] u64 (rounded to multiples of 2^6 inside).
// out: [742][row 2][col 2][16][64 lanes][2] doubles = folded forward transform, pre-scaled by 1/1024,
// computed with exactly the device kernel's operation order.
void convert_bsk_to_fft(const uint64_t *bsk_std, double *out, int nthreads);

}  // namespace fhs
