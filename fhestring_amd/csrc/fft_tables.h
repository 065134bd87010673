// Host-side twiddle tables for the optional f64-FFT arithmetic mode (fft_kernels.hip).
#pragma once
#include <cstdint>
#include <vector>

namespace fhs {

struct HostFftTables {
    std::vector<double> w_re, w_im;   // [1024] merged-twist twiddles W[m+i], index 0 unused
    std::vector<double> u_re, u_im;   // [3]    exp(i*pi/4), exp(i*pi/8), exp(3i*pi/8)
    // [12][64] per-lane bases, rows (re, im) of: layout B  W[16G + G*(lane>>2)], G = 1,2,4,8;
    //                                            layout C  W[256 + 4*lane], W[512 + 8*lane]
    std::vector<double> lanetab;
    // [1024][2] (re, im): the EFFECTIVE twiddle of every index, i.e. exactly the value the 2-wavefront kernel
    // uses for it: table value where it loads one, base * U where it derives one (same fma order), exact
    // rotation by i for odd offsets.  The 4-wavefront kernel loads its per-lane twiddles from this table, which is
    // what keeps the two kernels bit-identical.
    std::vector<double> weff;
    // two-bits-per-product kernel (fftmb_kernels.hip): mono [4096][2] = exp(i*pi*k/2048), r16 [16][2] = exp(i*pi*k/8)
    std::vector<double> mono, r16;
};
void build_fft_tables(HostFftTables &t);

}  // namespace fhs
