// Host-side exact construction of the device NTT tables and of the NTT-domain bootstrapping key.
#pragma once
#include <cstdint>
#include <vector>

namespace fhs {

struct HostNttTables {
    std::vector<double> fwd_uni;   // [2][32]
    std::vector<double> fwd_lane;  // [2][32][64]
    std::vector<double> inv_uni;   // [2][64]
    std::vector<double> inv_lane;  // [2][32][64]
    double crt_c;                  // p0^-1 mod p1, centred
    // nttmb_kernels.hip (two key bits per external product): psi_q^k centred, k < 4096 -- the value of the monomial X^e at
    // the evaluation point psi^(2 k' + 1) is mono[q][((2 k' + 1) e) mod 4096]
    std::vector<double> mono;      // [2][4096]
};
void build_ntt_tables(HostNttTables &t);
bool ntt_slot_roots_are_bitreversed();

// bsk_std: [n_ggsw][2][2][2048] u64 (rounded to multiples of 2^quant_bits inside).  out: [n_ggsw][2][2][2 primes][32][64]
// doubles = forward NTT of (signed bsk / 2^quant_bits) mod p, pre-scaled by N^-1, centred, in the device's
// contiguous-layout order.  Runs on `nthreads` host threads (key loading, not the hot path).
void convert_bsk_to_ntt(const uint64_t *bsk_std, double *out, int nthreads, int n_ggsw = 742, int quant_bits = 6);

}  // namespace fhs
